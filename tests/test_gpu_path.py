"""GPU parity of the hot path (HIP, through the C ABI) against the committed golden fixtures
(outputs of the real reference) and against the CPU oracle on fresh seeded inputs.
north_star tolerance: 1e-4 relative fp32 (max-norm), phases compared modulo 2*pi.
"""

import json
import os

import pytest
import torch

from conftest import phase_err, rel_err
from oracle import nets, optics, seeded, step

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
WL = torch.tensor([638e-9, 520e-9, 450e-9])
PITCH = 3.74e-6
PARITY = 1e-4


def _fixed(r0, c0, pad, coef):
    from learned_hologram_gan_amd.angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_single_fixed_distance as Fx

    return Fx(r0, c0, pad, coef, PITCH, WL, False, True, torch.tensor([1e-3]))


def _multi(r0, c0, stack, pad, coef):
    from learned_hologram_gan_amd.angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_multiple_distances as Mu

    return Mu(r0, c0, stack, pad, coef, PITCH, WL, False, True)


def cplx_err(a, b):
    return rel_err(torch.view_as_real(a.cpu()), torch.view_as_real(b))


# ----------------------------------------------------------------------------- A1: constants
def test_constants_match_golden_to_one_ulp(golden):
    """Host-built constants: identical op order to the reference, so they agree with the recorded ones up to the
    last-bit differences of the host's libm/MKL sqrt (<= 1 ulp of w, i.e. <= ~1e-3 rad of phase)."""
    g = golden("constants.pt")["sq48"]
    fx = _fixed(*g["args"])
    mu = _multi(g["args"][0], g["args"][1], g["distances"], g["args"][2], g["args"][3])
    assert ((fx.w_grid.cpu() - g["w"]).abs() <= 0.25).all()  # ulp(2e6) = 0.25
    assert (fx.diffraction_limited_mask.cpu() != g["mask"]).sum() <= 4
    assert (fx.H.cpu() - g["H_fixed"]).abs().max() < 2e-3 and (mu.H.cpu() - g["H_stack"]).abs().max() < 2e-3


# ----------------------------------------------------------------------------- A5 A8 A9 vs reference outputs
def test_asm_against_golden(golden):
    g = golden("asm_small.pt")
    r0, c0, pad, coef = g["args"]
    fx, mu = _fixed(r0, c0, pad, coef), _multi(r0, c0, g["stack"], pad, coef)
    fx.set_mask(g["consts"]["mask"])
    mu.set_mask(g["consts"]["mask"])
    fx.set_transfer_function(g["consts"]["H_fixed"])  # recorded with the fixture: host-CPU dependent
    mu.set_transfer_function(g["consts"]["H_stack"])
    mu.set_call_transfer_function(g["call_distances"], g["consts"]["H_call"])
    d = lambda k: g[k].to(DEV)  # noqa: E731
    assert cplx_err(fx.propagate_AP2C_backward(d("amp"), d("phs")), g["A5_field"]) < PARITY
    S = fx.propagate_POH2Freq_forward(d("poh"))
    assert cplx_err(S, g["A8_spectrum"]) < PARITY
    a, p = fx.propagate_POH2AP_forward(d("poh"))
    assert rel_err(a.cpu(), g["A8_amp"]) < PARITY
    T = mu.filter_AP2filteredFreq(d("tamp"), d("tphs"))
    assert cplx_err(T, g["A9_target_spectrum"]) < PARITY
    G = torch.cat((S, T), 0)
    a, p = mu.propagate_multiple_samples_with_random_fixed_multiple_distances_freq2amp(G, g["A9_indices"])
    assert rel_err(a.cpu(), g["A9_idx_amp"]) < PARITY
    big = g["A9_idx_amp"] > 1e-2 * g["A9_idx_amp"].max()
    assert phase_err(p.cpu()[big], g["A9_idx_phs"][big]) < 2e-3
    a, p = mu.propagate_multiple_samples_with_all_fixed_multiple_distances_freq2amp(G)
    assert rel_err(a.cpu(), g["A9_all_amp"]) < PARITY
    # fused training path == spectrum route
    ha, hp, ta, tp = mu.reconstruct_planes(fx, d("poh"), d("tamp"), d("tphs"), g["A9_indices"])
    assert rel_err(torch.cat((ha, ta)).cpu(), g["A9_idx_amp"]) < PARITY
    amp = mu(torch.ones_like(d("poh")), d("poh"), g["call_distances"])
    assert rel_err(amp.cpu(), g["call_amp"]) < PARITY


@pytest.mark.parametrize("r0,c0,pad", [(48, 48, 8), (40, 64, 4), (18, 90, 3)])
def test_asm_gradients_match_oracle(r0, c0, pad):
    """Padded extents 64x64 (radix 4), 48x72 and 24x96 (radix 4/2/3 mixed: 2^a * 3^b)."""
    coef = 0.45
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:5]
    fx, mu = _fixed(r0, c0, pad, coef), _multi(r0, c0, stack, pad, coef)
    o = optics.make_optics(r0, c0, pad, coef, PITCH, WL)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    Hs = optics.transfer_function(o.w, stack)
    g = torch.Generator().manual_seed(1)
    amp, phs = torch.rand((2, 3, r0, c0), generator=g) + 0.2, torch.rand((2, 3, r0, c0), generator=g) * 6
    poh = (torch.rand((2, 3, r0, c0), generator=g) - 0.5) * 8
    pa, pp = torch.randn((2, 3, r0, c0), generator=g), torch.randn((2, 3, r0, c0), generator=g)
    idx = torch.tensor([3, 1])

    def loss_field(f):
        return (f.real * pa.to(f.device) + f.imag * pp.to(f.device)).sum()

    ar, pr = amp.clone().requires_grad_(True), phs.clone().requires_grad_(True)
    loss_field(optics.backpropagate_to_slm(o, Hf, ar, pr)).backward()
    ag, pg = amp.to(DEV).requires_grad_(True), phs.to(DEV).requires_grad_(True)
    loss_field(fx.propagate_AP2C_backward(ag, pg)).backward()
    assert rel_err(ag.grad.cpu(), ar.grad) < PARITY and rel_err(pg.grad.cpu(), pr.grad) < PARITY

    def loss_planes(a, p):
        return (a * pa.to(a.device)).sum() + (torch.cos(p) * pp.to(a.device)).sum()

    qr = poh.clone().requires_grad_(True)
    S = optics.poh_to_filtered_spectrum(o, Hf, qr)
    T = optics.target_to_filtered_spectrum(o, amp, phs / 6)
    a, p = optics.spectrum_to_planes_indexed(o, Hs, torch.cat((S, T)), idx)
    loss_planes(a[:2], p[:2]).backward()
    qg = poh.to(DEV).requires_grad_(True)
    ha, hp, _, _ = mu.reconstruct_planes(fx, qg, amp.to(DEV), (phs / 6).to(DEV), idx)
    loss_planes(ha, hp).backward()
    assert rel_err(qg.grad.cpu(), qr.grad) < 5e-4
    # API route (full spectra) gives the same gradient
    qs = poh.to(DEV).requires_grad_(True)
    G = torch.cat((fx.propagate_POH2Freq_forward(qs), mu.filter_AP2filteredFreq(amp.to(DEV), (phs / 6).to(DEV))))
    a2, p2 = mu.propagate_multiple_samples_with_random_fixed_multiple_distances_freq2amp(G, idx)
    loss_planes(a2[:2], p2[:2]).backward()
    assert rel_err(qs.grad.cpu(), qr.grad) < 5e-4


def test_asm_full_size_1024():
    """384^2 frames, pad 320 -> 1024^2 transforms (the benchmark geometry), 2 samples."""
    r0 = c0 = 384
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1]
    fx, mu = _fixed(r0, c0, 320, 0.45), _multi(r0, c0, stack, 320, 0.45)
    o = optics.make_optics(r0, c0, 320, 0.45, PITCH, WL)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    g = torch.Generator().manual_seed(2)
    amp, phs = torch.rand((2, 3, r0, c0), generator=g), torch.rand((2, 3, r0, c0), generator=g)
    poh = (torch.rand((2, 3, r0, c0), generator=g) - 0.5) * 9
    idx = torch.tensor([17, 4])
    ref_field = optics.backpropagate_to_slm(o, Hf, amp * 1.1, phs * 6.28)
    assert cplx_err(fx.propagate_AP2C_backward((amp * 1.1).to(DEV), (phs * 6.28).to(DEV)), ref_field) < PARITY
    S = optics.poh_to_filtered_spectrum(o, Hf, poh)
    T = optics.target_to_filtered_spectrum(o, amp, phs)
    a, p = optics.spectrum_to_planes_indexed(o, optics.transfer_function(o.w, stack), torch.cat((S, T)), idx)
    ha, hp, ta, tp = mu.reconstruct_planes(fx, poh.to(DEV), amp.to(DEV), phs.to(DEV), idx)
    assert rel_err(torch.cat((ha, ta)).cpu(), a) < PARITY
    # linearity + Parseval-type property at full size: mask is a projector, |H| = 1
    S_gpu = fx.propagate_POH2Freq_forward(poh.to(DEV))
    assert cplx_err(S_gpu, S) < PARITY


# ----------------------------------------------------------------------------- A3 A4 A6 A7: generator
def _generator(rows, cols, pad, coef=0.45, consts=None):
    from learned_hologram_gan_amd.watermelon_hologram.generator import Generator

    G = Generator(rows, cols, pad, coef, 3, PITCH, WL, torch.tensor([1e-3]))
    missing = G.load_state_dict(seeded.generator_state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    if consts is not None:
        G.part2.propagator.set_mask(consts["mask"])
        G.part2.propagator.set_transfer_function(consts["H_fixed"])
    return G.to(DEV)


def test_unet_and_generator_against_golden(golden):
    g = golden("generator_small.pt")
    rows, cols, pad, coef = g["args"]
    G = _generator(rows, cols, pad, coef, g["consts"])
    assert {k: tuple(v.shape) for k, v in G.state_dict().items()} == g["key_shapes"]
    rgbd = g["rgbd"].to(DEV)
    G.eval()
    with torch.no_grad():
        assert rel_err(G.part1.part1(rgbd).cpu(), g["unet_eval"]) < PARITY
        assert phase_err(G(rgbd).cpu(), g["poh_eval"]) < 1e-3
    G.train()
    x = rgbd.clone().requires_grad_(True)
    y = G.part1.part1(x)
    (y * g["unet_proj"].to(DEV)).sum().backward()
    assert rel_err(y.detach().cpu(), g["unet_train"]) < PARITY
    # 18 train-mode BNs over as few as 8 samples (2x2 bottleneck, batch 2): ill-conditioned, fp32 summation order shows at
    # the 1e-2 level; test_unet_train_gradients_vs_fp64_truth bounds the GPU error by the CPU fp32 error instead.
    assert rel_err(x.grad.cpu(), g["unet_train_dx"]) < 5e-2
    named = dict(G.named_parameters())
    for k, ref in g["unet_train_param_grads"].items():
        if k.endswith("convolution_layer_2.bias"):
            continue  # feeds a train-mode BN: analytically zero
        assert abs(named[k].grad.norm().item() - ref["norm"]) <= 1e-3 * ref["norm"], k
        assert rel_err(named[k].grad.flatten()[:64].cpu(), ref["head"]) < 3e-2, k  # 8-sample BNs: see the fp64-truth test
    sd = G.state_dict()
    for k, ref in g["bn_after_one_train_fwd"].items():
        assert rel_err(sd[k].double().cpu(), ref.double()) < 1e-4, k
    # full generator, train mode, gradients through the ASM tail
    G = _generator(rows, cols, pad, coef, g["consts"])
    G.train()
    x = rgbd.clone().requires_grad_(True)
    poh = G(x)
    (torch.cos(poh) * g["poh_proj"].to(DEV)).sum().backward()
    assert phase_err(poh.detach().cpu(), g["poh_train"]) < 1e-3
    assert rel_err(x.grad.cpu(), g["poh_train_dx"]) < 2e-2
    named = dict(G.named_parameters())
    for k, ref in g["poh_train_param_grads"].items():
        assert rel_err(named[k].grad.cpu(), ref["full"]) < 2e-2, k


def l2_err(a, b):
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("rows,batch", [(32, 2), (64, 4)])
def test_unet_train_gradients_vs_fp64_truth(rows, batch):
    """Train-mode UNet forward + backward: the fp32 GPU result must be as close to an fp64 evaluation of the same
    maths as the fp32 CPU (reference-arithmetic) result is.

    Max-pool argmax is discontinuous: among ~5e5 2x2 windows some top-2 values differ by < 1e-6 relative, so two fp32
    implementations route a few window gradients to different pixels (measured: 1 flip at 64x64, tests/diagnostics/dbg_levels.py).
    Gradients are therefore compared in relative L2, plus an aggregate criterion over ALL parameters."""
    import statistics

    from learned_hologram_gan_amd.neural_network_components import UNet

    sd32 = {k[len("part1.part1."):]: v for k, v in seeded.generator_state_dict().items() if k.startswith("part1.part1.")}
    rgbd, _, _ = seeded.smooth_batch(batch, rows, rows, seed=21)
    proj = torch.randn((batch, 6, rows, rows), generator=torch.Generator().manual_seed(8))

    def run_oracle(dtype):
        sd = nets.as_parameters({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd32.items()})
        x = rgbd.detach().clone().to(dtype).requires_grad_(True)
        y = nets.unet(sd, "", x, True)
        (y * proj.to(dtype)).sum().backward()
        return y.detach().double(), x.grad.double(), {k: v.grad.double() for k, v in sd.items() if v.requires_grad}

    y64, dx64, gw64 = run_oracle(torch.float64)
    y32, dx32, gw32 = run_oracle(torch.float32)
    net = UNet(6, 4)
    net.load_state_dict(sd32)
    net.to(DEV).train()
    x = rgbd.detach().clone().to(DEV).requires_grad_(True)
    y = net(x)
    (y * proj.to(DEV)).sum().backward()
    assert rel_err(y.detach().cpu().double(), y64) <= max(4 * rel_err(y32, y64), 2e-6)
    assert l2_err(x.grad.cpu().double(), dx64) <= max(4 * l2_err(dx32, dx64), 2e-2)
    ratios = []
    for k, p in net.named_parameters():
        if k.endswith(("convolution_layer_1.bias", "convolution_layer_2.bias")):
            continue  # analytically zero (feeds a train-mode BN)
        e_gpu, e_cpu = l2_err(p.grad.cpu().double(), gw64[k]), l2_err(gw32[k], gw64[k])
        assert e_gpu <= max(6 * e_cpu, 2e-2), (k, e_gpu, e_cpu)
        ratios.append(e_gpu / max(e_cpu, 1e-12))
    assert statistics.median(ratios) <= 3.0  # 1.07 at 64x64 batch 4; ~2 on the 8-sample-BN 32x32 case


@pytest.mark.parametrize("rows,cols,pad,batch", [(64, 64, 32, 2), (96, 96, 16, 1)])
def test_generator_eval_vs_oracle(rows, cols, pad, batch):
    G = _generator(rows, cols, pad).eval()
    rgbd, _, _ = seeded.smooth_batch(batch, rows, cols, seed=13)
    with torch.no_grad():
        poh = G(rgbd.to(DEV)).cpu()
        amp, _ = G.part1(rgbd.to(DEV))
    o = optics.make_optics(rows, cols, pad, 0.45, PITCH, WL)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    sd = nets.as_parameters(seeded.generator_state_dict())
    with torch.no_grad():
        amp_ref, _ = nets.rgbd_to_amp_phase(sd, rgbd, False)
        poh_ref = nets.generator(sd, o, Hf, rgbd, False)
    assert rel_err(amp.cpu(), amp_ref) < PARITY
    assert phase_err(poh, poh_ref) < 1e-3
    # quality metric of the north star: reconstruction PSNR of GPU POH vs oracle POH
    a_gpu, _ = optics.poh_to_amp_phase(o, Hf, poh)
    a_ref, _ = optics.poh_to_amp_phase(o, Hf, poh_ref)
    mse = ((a_gpu - a_ref) ** 2).mean()
    assert 10 * torch.log10(a_ref.max() ** 2 / mse) > 60.0


# ----------------------------------------------------------------------------- A10 A11: critic
def test_critic_and_gradient_penalty_against_golden(golden):
    from learned_hologram_gan_amd.watermelon_hologram.discriminator import WGANGPDiscriminator192

    g = golden("critic_small.pt")
    D = WGANGPDiscriminator192(None, 32, True)
    D.load_state_dict(seeded.critic_state_dict(), strict=True)
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == g["key_shapes"]
    real, fake = g["real"].to(DEV), g["fake"].to(DEV)
    D.eval()
    with torch.no_grad():
        assert rel_err(D(real).cpu(), g["score_eval"]) < PARITY
    D.train()
    rv, fv = D(real), D(fake)
    alpha = g["alpha"].to(DEV)
    x_hat = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
    s = D(x_hat)
    (gx,) = torch.autograd.grad(s, x_hat, torch.ones_like(s), create_graph=True, retain_graph=True)
    gp = ((gx.view(2, -1).norm(2, dim=1) - 1) ** 2).mean()
    d_loss = (-rv.mean() + fv.mean()) + 10 * gp
    d_loss.backward()
    assert rel_err(rv.detach().cpu(), g["score_real_train"]) < PARITY
    assert abs(gp.item() - g["gp"]) <= 2e-4 * abs(g["gp"])
    assert abs(d_loss.item() - g["d_loss"]) <= 2e-4 * abs(g["d_loss"])
    for k, p in D.named_parameters():
        if k in ("block2.0.bias", "block3.0.bias", "block4.0.bias", "block5.0.bias", "block6.0.bias"):
            continue
        ref = g["param_grads"][k]
        assert abs(p.grad.norm().item() - ref["norm"]) <= 5e-3 * ref["norm"] + 1e-7, k
    for k, ref in g["bn_after"].items():
        assert rel_err(D.state_dict()[k].double().cpu(), ref.double()) < 1e-4, k


# ----------------------------------------------------------------------------- A12: one full training step
def test_train_step_against_reference_loop(golden):
    """Fixture produced by the reference's own watermelon.train loop (oracle/make_golden.py)."""
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    g = golden("step_small.pt")
    rows, cols, pad, coef = g["args"]
    W = watermelon(filter_radius_coefficient=coef, pad_size=pad, distance_stack=g["stack"], input_shape=(1, 4, rows, cols))
    W.generator.load_state_dict(seeded.generator_state_dict(), strict=True)
    W.discriminator.load_state_dict(seeded.critic_state_dict(), strict=True)
    W.generator.part2.propagator.set_mask(g["consts"]["mask"])
    W.propagator.set_mask(g["consts"]["mask"])
    W.generator.part2.propagator.set_transfer_function(g["consts"]["H_fixed"])
    W.propagator.set_transfer_function(g["consts"]["H_stack"])
    W.generator.to(DEV).train()
    W.discriminator.to(DEV).train()
    W.configure(phs_gradient_loss_weight=1, perceptual_loss_weight=0.0, pixel_loss_weight=1, TV_loss_weight=1e-3,
                discriminator_loss_weight=1e-1, lr_G=1e-3, lr_D=1e-3, discriminator_train_ratio=g["ratio"], discriminator_lambda=10)
    W.train_step(g["rgbd"].to(DEV), g["tamp"].to(DEV), g["tphs"].to(DEV), plane_indices=g["indices"],
                 gp_alphas=[a.to(DEV) for a in g["alphas"]])
    got = dict(zip(("focal_phase_gradient_loss", "perceptual_loss", "pixel_loss", "TV_loss", "gan_loss", "G_loss", "D_loss"),
                   W.train_losses_tensor.tolist()))
    # Quantities that pass through the UPDATED critic are compared more loosely: Adam's first steps move every
    # weight by ~lr*sign(g), the critic output moves by O(1) per step (gradient-penalty loss ~1e2), and the fp32
    # rounding differences between the GPU and CPU GEMMs (gradients agree to ~3e-6, tests/diagnostics/diag_critic.py) are
    # amplified to ~2e-3 of gan_loss after two critic updates.
    loose = {"gan_loss": 2e-2, "G_loss": 2e-2, "D_loss": 2e-2}
    for k, ref in g["losses"].items():
        assert abs(got[k] - ref) <= loose.get(k, 1e-3) * abs(ref) + 1e-6, (k, got[k], ref)
    sdG, sdD = W.generator.state_dict(), W.discriminator.state_dict()
    for sd, post in ((sdD, g["post_D_small"]), (sdG, g["post_G_small"])):
        for k, v in post.items():
            if k in ("block2.0.bias", "block3.0.bias", "block4.0.bias", "block5.0.bias", "block6.0.bias"):
                continue
            # Adam's first step moves every weight by lr*g/(|g|+eps) ~ lr*sign(g): a gradient within rounding noise of zero may come out
            # with the other sign (or a different fraction of lr) and then the weight differs by AT MOST 2*lr.  Such weights are few (a
            # handful per tensor: which ones flip changes with any change of summation order) and bounded; everything else must agree.
            d = (sd[k].cpu() - v).abs()
            bad = d > 2e-4 + 1e-3 * v.abs()
            assert bad.float().mean().item() <= max(5e-3, 6.0 / v.numel()), (k, int(bad.sum()), v.numel())
            assert not bad.any() or d[bad].max().item() <= 2.1 * 1e-3, (k, d[bad].max().item())


def test_train_step_vs_oracle_fresh_inputs():
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    rows = cols = 32
    pad, coef, ratio = 16, 0.45, 1
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
    rgbd, tamp, tphs = seeded.smooth_batch(2, rows, cols, seed=77)
    idx = torch.tensor([5, 2])
    alphas = [torch.tensor([0.3, 0.8]).view(2, 1, 1, 1)]
    st = step.make_state(rows, cols, pad, coef, stack, seeded.generator_state_dict(), seeded.critic_state_dict())
    ref = step.train_step(st, rgbd, tamp, tphs, step.LossWeights(d_ratio=ratio), idx, alphas)
    W = watermelon(filter_radius_coefficient=coef, pad_size=pad, distance_stack=stack, input_shape=(1, 4, rows, cols))
    W.generator.load_state_dict(seeded.generator_state_dict())
    W.discriminator.load_state_dict(seeded.critic_state_dict())
    W.generator.to(DEV).train()
    W.discriminator.to(DEV).train()
    W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, ratio, 10)
    out = W.train_step(rgbd.to(DEV), tamp.to(DEV), tphs.to(DEV), idx, [a.to(DEV) for a in alphas])
    # bounds the float64-truth tests support (tests/test_gpu_truth.py: two fp32 evaluations of this chain are <= ~1e-4 apart at these
    # sizes); round 4 accepted 1e-3 here.  The measured values travel with a failure.
    e_poh, e_amp = phase_err(out["POH"].cpu(), ref["POH"]), rel_err(out["hat_amps"].cpu(), ref["hat_amps"])
    e_g = abs(out["G_loss"].item() - ref["G_loss"]) / abs(ref["G_loss"])
    e_d = abs(out["D_loss"].item() - ref["D_loss"]) / abs(ref["D_loss"])
    # G_loss contains -mean D(hat) of the UPDATED critic (one Adam step of ~lr * sign(g) per weight: rounding-level differences of the critic's
    # gradients move its output, see test_train_step_against_reference_loop): measured 9.6e-4 here; D_loss is taken before the update
    assert e_poh < 2e-4 and e_amp < 2e-4 and e_g <= 3e-3 and e_d <= 2e-4, dict(poh=e_poh, hat_amps=e_amp, G_loss=e_g, D_loss=e_d)


def test_config1_192_inference_and_one_plane_propagation():
    """BASELINE configs[0]: 192x192 single-sample generatePOH forward + 1-plane propagate (pad 160 -> 512^2 FFTs)."""
    from learned_hologram_gan_amd.angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_multiple_distances as Mu

    G = _generator(192, 192, 160).eval()
    rgbd, _, _ = seeded.smooth_batch(1, 192, 192, seed=31)
    d = torch.tensor([4e-4])
    prop = Mu(192, 192, d, 160, 0.35, PITCH, WL, False, True)
    with torch.no_grad():
        poh = G(rgbd.to(DEV))
        amp = prop(torch.ones_like(poh), poh, d)
    o = optics.make_optics(192, 192, 160, 0.45, PITCH, WL)
    o35 = optics.make_optics(192, 192, 160, 0.35, PITCH, WL)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    with torch.no_grad():
        poh_ref = nets.generator(nets.as_parameters(seeded.generator_state_dict()), o, Hf, rgbd, False)
        amp_ref = optics.propagate_amplitudes(o35, torch.ones_like(poh_ref), poh_ref, d)
    e_poh, e_amp = phase_err(poh.cpu(), poh_ref), rel_err(amp.cpu(), amp_ref)
    assert e_poh < 2e-4 and e_amp < 2e-4, dict(poh=e_poh, amp=e_amp)  # (round 4: 1e-3; the eval-mode generator has no batch statistics to amplify rounding)
    assert rel_err(prop(torch.ones_like(poh), poh_ref.to(DEV), d).cpu(), amp_ref) < PARITY  # propagation alone: 1e-4


def test_config4_4k_frame_unet_locality():
    """BASELINE configs[3] geometry (3840x2160, batch 1): the eval-mode UNet is a local operator (BN is a per-channel
    affine map), so the centre of the 4K output must equal the oracle evaluated on a crop with >= 192 px of margin.
    Also exercises the > 4 GiB tensors (the pipelined kernels' 32-bit descriptors fall back to the 64-bit kernel)."""
    from learned_hologram_gan_amd.neural_network_components import UNet

    H, W = 2160, 3840
    sd = {k[len("part1.part1."):]: v for k, v in seeded.generator_state_dict().items() if k.startswith("part1.part1.")}
    net = UNet(6, 4)
    net.load_state_dict(sd)
    net.to(DEV).eval()
    g = torch.Generator().manual_seed(41)
    small = torch.rand((1, 4, H // 8, W // 8), generator=g)
    rgbd = torch.nn.functional.interpolate(small, size=(H, W), mode="bilinear", align_corners=False)
    with torch.no_grad():
        y = net(rgbd.to(DEV))
    assert y.shape == (1, 6, H, W) and torch.isfinite(y).all()
    cy, cx, half, margin = 1088, 1920, 64, 192  # multiples of 16 keep the pooling grid aligned
    crop = rgbd[:, :, cy - half - margin: cy + half + margin, cx - half - margin: cx + half + margin]
    with torch.no_grad():
        ref = nets.unet(nets.as_parameters(sd), "", crop, False)
    got = y[:, :, cy - half: cy + half, cx - half: cx + half].cpu()
    assert rel_err(got, ref[:, :, margin: margin + 2 * half, margin: margin + 2 * half]) < PARITY


def test_asm_4k_geometry_radix3():
    """BASELINE configs[4]: 2160x3840 frames, pad 72 -> 2304 x 4096 transforms (2304 = 2^8 * 3^2) on the LDS FFT."""
    r0, c0, pad = 2160, 3840, 72
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1]
    fx, mu = _fixed(r0, c0, pad, 0.45), _multi(r0, c0, stack, pad, 0.45)
    assert fx._geom.supported()
    o = optics.make_optics(r0, c0, pad, 0.45, PITCH, WL)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    g = torch.Generator().manual_seed(12)
    small = torch.rand((3, 1, 3, r0 // 8, c0 // 8), generator=g)
    amp, phs, poh = (torch.nn.functional.interpolate(t, size=(r0, c0), mode="bilinear") for t in small)
    ref_field = optics.backpropagate_to_slm(o, Hf, amp + 0.1, phs * 6.28)
    assert cplx_err(fx.propagate_AP2C_backward((amp + 0.1).to(DEV), (phs * 6.28).to(DEV)), ref_field) < PARITY
    idx = torch.tensor([13])
    S = optics.poh_to_filtered_spectrum(o, Hf, (poh - 0.5) * 9)
    T = optics.target_to_filtered_spectrum(o, amp, phs)
    a, _ = optics.spectrum_to_planes_indexed(o, optics.transfer_function(o.w, stack), torch.cat((S, T)), idx)
    ha, _, ta, _ = mu.reconstruct_planes(fx, ((poh - 0.5) * 9).to(DEV), amp.to(DEV), phs.to(DEV), idx)
    assert rel_err(torch.cat((ha, ta)).cpu(), a) < PARITY


# the last five: columns on either side of every rule of the convolution-length choice — 2049 -> 4608 = 2^9 3^2, 3073 -> 8192 (6912 does not
# fit 512 threads), 4097 -> 9216 = 2^10 3^2, 6000 -> 12288 = 2^12 3, 6200 -> 16384, 8190 -> 16384
@pytest.mark.parametrize("r0,c0,pad", [(192, 192, 320), (100, 140, 21), (77, 90, 5), (64, 2100, 4), (64, 4848, 1), (32, 2049, 0), (32, 3073, 0),
                                       (32, 4097, 0), (32, 6000, 0), (32, 6200, 0), (32, 8190, 0)], ids=lambda v: str(v))
def test_extents_outside_2a3b_run_on_the_hip_operator(r0, c0, pad):
    """192 + 2*320 = 832 = 2^6 * 13 (the reference CLI default for 192^2 frames; torch.fft takes any extent,
    angular_spectrum_method.py:382-392) and other lengths outside 2^a 3^b stay on the fused HIP operator: products of primes up to 13
    through radix-5 / 7 / 11 / 13 Stockham stages (832; 198 = 2 3^2 11; 100 = 2^2 5^2), everything else as a Bluestein convolution
    inside the same three passes (142 = 2 71 -> length 512, 87 = 3 29 -> 256, 2108 columns -> 4608 = 2^9 3^2, 4850 -> 10368 = 2^7 3^4: the
    sizes a 4K frame with that pad needs).  Forward values against the oracle, gradients against the oracle's autograd."""
    from learned_hologram_gan_amd import asm_ops

    fx = _fixed(r0, c0, pad, 0.45)
    R, C = fx.samplingRowNum, fx.samplingColNum
    def pow23(n):
        while n % 2 == 0:
            n //= 2
        while n % 3 == 0:
            n //= 3
        return n == 1

    assert fx._geom.supported() and not (pow23(R) and pow23(C))
    g = torch.Generator().manual_seed(4)
    amp, phs = torch.rand((2, 3, r0, c0), generator=g) + 0.1, torch.rand((2, 3, r0, c0), generator=g) * 6
    o = optics.make_optics(r0, c0, pad, 0.45, PITCH, WL)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    assert cplx_err(fx.propagate_AP2C_backward(amp.to(DEV), phs.to(DEV)), optics.backpropagate_to_slm(o, Hf, amp, phs)) < PARITY
    a, _ = fx.propagate_POH2AP_forward(phs.to(DEV))
    assert rel_err(a.cpu(), optics.poh_to_amp_phase(o, Hf, phs)[0]) < PARITY
    # adjoint passes: d sum(|field|^2 weighted) / d (amp, phs)
    wgt = torch.rand((2, 3, r0, c0), generator=g)
    ad, pd = amp.clone().requires_grad_(True), phs.clone().requires_grad_(True)
    f_ref = optics.backpropagate_to_slm(o, Hf, ad, pd)
    ((f_ref.real * wgt).sum() + (f_ref.imag * wgt.flip(-1)).sum()).backward()
    ah, ph = amp.to(DEV).requires_grad_(True), phs.to(DEV).requires_grad_(True)
    f_hip = fx.propagate_AP2C_backward(ah, ph)
    ((f_hip.real * wgt.to(DEV)).sum() + (f_hip.imag * wgt.flip(-1).to(DEV)).sum()).backward()
    assert rel_err(ah.grad.cpu(), ad.grad) < 5 * PARITY and rel_err(ph.grad.cpu(), pd.grad) < 5 * PARITY


def test_asm_4k_frame_with_the_cli_default_pad():
    """A 2160 x 3840 frame with the reference CLI's default pad_size 320 (generatePOH.py:96): 2800 x 4976 = (2^4 5^2 7) x (2^4 311)
    transforms: the rows direct, the columns a Bluestein convolution of length 10368 = 2^7 3^4 on the HIP operator."""
    import time

    r0, c0, pad = 2160, 3840, 320
    fx = _fixed(r0, c0, pad, 0.45)
    assert (fx.samplingRowNum, fx.samplingColNum) == (2800, 4976) and fx._geom.supported()
    o = optics.make_optics(r0, c0, pad, 0.45, PITCH, WL)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    g = torch.Generator().manual_seed(21)
    small = torch.rand((2, 1, 3, r0 // 8, c0 // 8), generator=g)
    amp, phs = (torch.nn.functional.interpolate(t, size=(r0, c0), mode="bilinear") for t in small)
    ref = optics.backpropagate_to_slm(o, Hf, amp + 0.1, phs * 6.28)
    a, p = (amp + 0.1).to(DEV), (phs * 6.28).to(DEV)
    got = fx.propagate_AP2C_backward(a, p)
    assert cplx_err(got, ref) < PARITY
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fx.propagate_AP2C_backward(a, p)
    torch.cuda.synchronize()
    print("4K pad-320 back-propagation (3 planes): %.1f ms on the HIP Bluestein path" % ((time.perf_counter() - t0) / 3 * 1e3))


def test_extent_beyond_the_lds_transforms_uses_rocfft_route():
    """Lengths the LDS kernels cannot hold (non-smooth above 8192) stay on torch.fft / rocFFT on the GPU — same maths, still no CPU
    fallback.  Checked at 8800 + 2*275 = 9350 = 2 * 5^2 * 11 * 17 columns."""
    r0, c0, pad_r = 32, 8800, 1
    fx = _fixed(r0, c0, pad_r, 0.45)
    assert not fx._geom.supported()
    g = torch.Generator().manual_seed(5)
    amp, phs = torch.rand((1, 3, r0, c0), generator=g) + 0.1, torch.rand((1, 3, r0, c0), generator=g) * 6
    o = optics.make_optics(r0, c0, pad_r, 0.45, PITCH, WL)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    assert cplx_err(fx.propagate_AP2C_backward(amp.to(DEV), phs.to(DEV)), optics.backpropagate_to_slm(o, Hf, amp, phs)) < PARITY


def test_full_size_train_step_vs_oracle(oracle_full_step):
    """BASELINE configs[1] at full size: 384x384, batch 4, pad 320 (1024^2 FFTs), 20-plane stack, one critic update with the
    gradient penalty, generator loss/backward, both Adam steps — HIP path vs the CPU oracle on identical seeded inputs."""
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    cfg, ref = oracle_full_step
    rows, cols, pad, coef, stack = cfg["rows"], cfg["cols"], cfg["pad"], cfg["coef"], cfg["stack"]
    rgbd, tamp, tphs, idx, alphas = cfg["rgbd"], cfg["tamp"], cfg["tphs"], cfg["idx"], cfg["alphas"]
    W = watermelon(filter_radius_coefficient=coef, pad_size=pad, distance_stack=stack, input_shape=(1, 4, rows, cols))
    W.generator.load_state_dict(seeded.generator_state_dict())
    W.discriminator.load_state_dict(seeded.critic_state_dict())
    W.generator.to(DEV).train()
    W.discriminator.to(DEV).train()
    W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10)
    out = W.train_step(rgbd.to(DEV), tamp.to(DEV), tphs.to(DEV), idx, [a.to(DEV) for a in alphas])
    # angle() is ill-conditioned where |field| is small: bound the bulk tightly and the worst pixel loosely
    perr = (torch.exp(1j * out["POH"].cpu()) - torch.exp(1j * ref["POH"])).abs().flatten()
    # north_star's 1e-4 asserted DIRECTLY against the CPU oracle (round 4): the default mode's distance to a float64 evaluation is
    # 2.5e-5 (hat_amps, max norm) / 3.6e-5 (hologram, 99.9 % quantile), the CPU's own 3.3e-5 / 2.3e-5 (profiles/r03_truth_tests.jsonl),
    # so the two fp32 evaluations are at most 5.9e-5 apart
    assert torch.quantile(perr[::7], 0.999) < PARITY and perr.max() < 5e-2, (torch.quantile(perr[::7], 0.999).item(), perr.max().item())
    assert rel_err(out["hat_amps"].cpu(), ref["hat_amps"]) < PARITY
    assert rel_err(out["target_amps"].cpu(), ref["target_amps"]) < PARITY
    got = dict(zip(("focal_phase_gradient_loss", "perceptual_loss", "pixel_loss", "TV_loss", "gan_loss", "G_loss", "D_loss"),
                   W.train_losses_tensor.tolist()))
    for k in ("focal_phase_gradient_loss", "pixel_loss", "TV_loss"):
        assert abs(got[k] - ref[k]) <= 1e-3 * abs(ref[k]) + 1e-7, (k, got[k], ref[k])
    for k in ("gan_loss", "G_loss", "D_loss"):  # pass through the updated critic (Adam sign amplification, see the reference-loop test)
        assert abs(got[k] - ref[k]) <= 3e-2 * abs(ref[k]) + 1e-6, (k, got[k], ref[k])
    # reconstruction quality metric of the north star: PSNR of the GPU reconstruction against the oracle reconstruction
    mse = ((out["hat_amps"].cpu() - ref["hat_amps"]) ** 2).mean()
    assert 10 * torch.log10(ref["hat_amps"].max() ** 2 / mse) > 70.0
    # post-Adam weights of both models at the bench size, bounded as test_train_step_against_reference_loop bounds
    # them: Adam's first step moves every weight by ~lr * sign(g), so a gradient within rounding noise of zero may come out with the other
    # sign and its weight then differs by at most 2 lr — such weights must be few and bounded, everything else must agree
    lr = 1e-3
    for sd, post in ((W.discriminator.state_dict(), ref["weights_D"]), (W.generator.state_dict(), ref["weights_G"])):
        for k, v in post.items():
            if k.endswith(("convolution_layer_1.bias", "convolution_layer_2.bias", "block2.0.bias", "block3.0.bias", "block4.0.bias", "block5.0.bias", "block6.0.bias")):
                continue  # analytically zero gradient (conv bias in front of a train-mode BatchNorm): Adam turns rounding noise into +-lr
            # How many weights may flip: two fp32 evaluations whose gradients are eps apart (relative L2) disagree in sign where
            # |g_i| < ~eps rms(g), i.e. for a fraction ~0.8 eps of the elements.  At this size the float64-truth test records eps up to
            # 1e-2 for BOTH evaluations on the first layer (encoder1 conv1: e_gpu 9.3e-3, e_cpu 1.08e-2, profiles/r05_truth_tests.jsonl):
            # up to ~1.5 % flips there (measured 0.4 - 0.6 %); the small-size reference-loop test keeps its 0.5 %.
            d = (sd[k].cpu() - v).abs()
            bad = d > 2e-4 + 1e-3 * v.abs()
            assert bad.float().mean().item() <= max(2e-2, 6.0 / v.numel()), (k, int(bad.sum()), v.numel())
            assert not bad.any() or d[bad].max().item() <= 2.1 * lr, (k, d[bad].max().item())


def test_perceptual_loss_vs_oracle(tmp_path):
    """VGG19 features[:32] perceptual loss on the HIP conv/pool ops vs the CPU restatement, same seeded
    weights (loaded through the weights-file path, torchvision key names)."""
    from learned_hologram_gan_amd.watermelon_hologram.perceptual import perceptualLoss
    from oracle import perceptual as P

    sd = P.vgg19_feature_state_dict()
    path = tmp_path / "vgg19_features.pth"
    torch.save({"features." + k: v for k, v in sd.items()}, path)
    mod = perceptualLoss(weights_path=str(path))
    assert mod.pretrained and all(not p.requires_grad for p in mod.parameters())
    g = torch.Generator().manual_seed(2)
    hat, tgt = torch.rand((2, 3, 64, 64), generator=g), torch.rand((2, 3, 64, 64), generator=g)
    hr = hat.clone().requires_grad_(True)
    ref = P.perceptual_loss(sd, hr, tgt)
    ref.backward()
    hg = hat.to(DEV).requires_grad_(True)
    out = mod(hg, tgt.to(DEV))
    out.backward()
    assert abs(out.item() - ref.item()) <= 1e-4 * abs(ref.item())
    assert ((hg.grad.cpu() - hr.grad).norm() / hr.grad.norm()).item() < 1e-3


def test_train_loop_validation_checkpoints_and_metrics_json(tmp_path):
    """The reference's loop contract (watermelon.py:92-416, 479-631): per-interval validation over ALL planes, the losses /
    metrics JSON schema, `<path>_epoch{n}.pth` + final checkpoints that load back into fresh modules."""
    import json

    from learned_hologram_gan_amd.watermelon_hologram.discriminator import WGANGPDiscriminator192
    from learned_hologram_gan_amd.watermelon_hologram.generator import Generator
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import LOSS_NAMES, watermelon

    rows = cols = 32
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:6]
    W = watermelon(filter_radius_coefficient=0.45, pad_size=16, distance_stack=stack, input_shape=(1, 4, rows, cols))
    W.generator.load_state_dict(seeded.generator_state_dict())
    W.discriminator.load_state_dict(seeded.critic_state_dict())
    batches = [tuple(t.to(DEV) for t in seeded.smooth_batch(2, rows, cols, seed=60 + i)) for i in range(2)]
    val = [tuple(t.to(DEV) for t in seeded.smooth_batch(1, rows, cols, seed=70))]
    pG, pD, pJ = str(tmp_path / "G.pth"), str(tmp_path / "D.pth"), str(tmp_path / "metrics.csv")
    torch.manual_seed(5)
    W.train(batches, val, phs_gradient_loss_weight=1, perceptual_loss_weight=0.0, pixel_loss_weight=1, TV_loss_weight=1e-3,
            discriminator_loss_weight=1e-1, epoch_num=1, lr_G=1e-3, lr_D=1e-3, save_path_G=pG, save_path_D=pD, info_print_interval=1,
            info_plot_interval=10**9, loss_metrics_file=pJ, save_path_img=None, checkpoint_iterval=1, discriminator_train_ratio=1,
            discriminator_lambda=10)
    rec = json.load(open(pJ))  # the reference writes JSON under a .csv name (README.md:65)
    assert rec["n_batch"] == [1, 2] and rec["n_train"] == [2, 4] and rec["epoch"] == [0, 0]
    for section in ("train_losses_tensor", "validate_losses_tensor"):
        assert set(rec[section]) == set(LOSS_NAMES) and all(len(v) == 2 for v in rec[section].values())
        assert all(torch.isfinite(torch.tensor(v)).all() for k, v in rec[section].items())
    assert set(rec["train_metrics_tensor"]) == {"PSNR", "SSIM"} and rec["validate_losses_tensor"]["D_loss"] == [0.0, 0.0]
    for f in ("G.pth", "G_epoch0.pth", "D.pth", "D_epoch0.pth"):
        assert (tmp_path / f).exists(), f
    G2 = Generator(rows, cols, 16, 0.45, 3, PITCH, WL, torch.tensor([1e-3]), pretrained_model_path=pG)
    D2 = WGANGPDiscriminator192(pretrained_model_path=pD, cuda=True)
    for k, v in W.generator.state_dict().items():
        assert torch.equal(G2.state_dict()[k].cpu(), v.cpu()), k
    assert int(D2.state_dict()["block2.1.num_batches_tracked"]) == 2 * 4  # 3 critic + 1 generator-side forwards per step
    assert W.generator.training and W.discriminator.training  # validation restores train mode


def test_hipgraph_inference_matches_eager():
    from learned_hologram_gan_amd.graph import GraphedGenerator

    G = _generator(64, 64, 32).eval()
    a, _, _ = seeded.smooth_batch(1, 64, 64, seed=81)
    b, _, _ = seeded.smooth_batch(1, 64, 64, seed=82)
    with torch.no_grad():
        ea, eb = G(a.to(DEV)).clone(), G(b.to(DEV)).clone()
    graphed = GraphedGenerator(G, a.to(DEV))
    assert torch.equal(graphed(b.to(DEV)), eb) and torch.equal(graphed(a.to(DEV)), ea)  # bit-identical replays, new inputs honoured
    with pytest.raises(ValueError):
        graphed(torch.zeros(2, 4, 64, 64, device=DEV))


def test_product_has_no_cpu_fallback():
    from learned_hologram_gan_amd import hip_ops, native

    with pytest.raises(native.NativeLibraryError):
        hip_ops.ToNHWC.apply(torch.rand(1, 3, 4, 4), 32)


# ----------------------------------------------------------------------------- A14 / N2: all-planes validation
def test_validate_generator_against_reference(golden):
    """Fixture produced by the reference's own ``_validate_generator``; SSIM against the oracle's float64 restatement."""
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    g = golden("validate_small.pt")
    rows, cols, pad, coef = g["args"]
    W = watermelon(filter_radius_coefficient=coef, pad_size=pad, distance_stack=g["stack"], input_shape=(1, 4, rows, cols))
    W.generator.load_state_dict(seeded.generator_state_dict(), strict=True)
    W.discriminator.load_state_dict(seeded.critic_state_dict(), strict=True)
    for pr, H in ((W.generator.part2.propagator, g["consts"]["H_fixed"]), (W.propagator, g["consts"]["H_stack"])):
        pr.set_mask(g["consts"]["mask"])
        pr.set_transfer_function(H)
    W.generator.to(DEV)
    W.discriminator.to(DEV)
    W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10)
    v_losses, v_metrics = W._validate_generator([tuple(t.to(DEV) for t in b) for b in g["batches"]])
    assert W.generator.training and W.discriminator.training  # the reference switches back to train mode
    got = dict(zip(("focal_phase_gradient_loss", "perceptual_loss", "pixel_loss", "TV_loss", "gan_loss", "G_loss", "D_loss"),
                   v_losses.tolist()))
    for k, ref in g["losses"].items():
        assert abs(got[k] - ref) <= 2e-4 * abs(ref) + 1e-7, (k, got[k], ref)
    assert abs(v_metrics[0].item() - g["psnr"]) < 1e-3
    st = step.make_state(rows, cols, pad, coef, g["stack"], seeded.generator_state_dict(), seeded.critic_state_dict(), consts=g["consts"])
    ref = step.validate(st, g["batches"], step.LossWeights(d_ratio=1))
    assert abs(v_metrics[1].item() - ref["SSIM"]) < 1e-4


# ----------------------------------------------------------------------------- N4: stand-alone pre-training loops
def test_pretraining_loops_against_reference(golden):
    """Fixture produced by the reference's own RGBD2AP.train_model / AP2POH.train_model (two epochs, Adam, plateau schedule)."""
    from learned_hologram_gan_amd.watermelon_hologram.AP2POH import AP2POH
    from learned_hologram_gan_amd.watermelon_hologram.RGBD2AP import RGBD2AP

    g = golden("pretrain_small.pt")
    rows, cols, pad, coef = g["args"]
    sdG = seeded.generator_state_dict()
    dev = lambda batches: [tuple(t.to(DEV) for t in b) for b in batches]  # noqa: E731

    m1 = RGBD2AP(input_shape=(1, 4, rows, cols))
    m1.load_state_dict({k[len("part1."):]: v for k, v in sdG.items() if k.startswith("part1.")}, strict=True)
    m1.to(DEV)
    m1.train_model(dev(g["train"]), dev(g["val"]), epochs=2, lr=1e-3, alpha=1e-3, hyperparameter_gamma=0.1, save_path=None)
    ref = g["rgbd2ap"]
    # epoch 1 already contains one Adam update (second batch): two fp32 runs separate through the lr*sign(g) first steps
    # (the reference vs its own CPU restatement: 1e-4 after two epochs, tests/test_oracle_golden.py)
    assert abs(m1.train_loss[0] - ref["train_loss"][0]) < 2e-4 * ref["train_loss"][0]
    assert abs(m1.train_loss[1] - ref["train_loss"][1]) < 2e-3 * ref["train_loss"][1]
    assert all(abs(a - b) < 2e-3 * b for a, b in zip(m1.test_loss, ref["test_loss"]))

    m2 = AP2POH(input_shape=(1, 6, rows, cols), pad_size=pad, filter_radius_coefficient=coef, pixel_pitch=PITCH, wave_length=WL,
                distance=torch.tensor([1e-3]), kernel_size=3)
    m2.load_state_dict({k[len("part2."):]: v for k, v in sdG.items() if k.startswith("part2.")}, strict=True)
    ref = g["ap2poh"]
    m2.propagator.set_mask(ref["consts"]["mask"])
    m2.propagator.set_transfer_function(ref["consts"]["H_fixed"])
    m2.to(DEV)
    fa, fp = m2.dataloader_filter(g["train"][0][1].to(DEV), (g["train"][0][2] * 6.0).to(DEV), coef)
    assert rel_err(fa.cpu(), ref["filtered_amp"]) < PARITY
    ok = ref["filtered_amp"] > 0.05 * ref["filtered_amp"].max()  # angle() is ill-conditioned where the field vanishes
    assert phase_err(fp.cpu()[ok], ref["filtered_phs"][ok]) < 1e-3
    ap = lambda batches: [(b[1].to(DEV), (b[2] * 6.0).to(DEV)) for b in batches]  # noqa: E731
    m2.train_model(ap(g["train"]), ap(g["val"]), filter_radius_coefficient=coef, epochs=2, lr=1e-3, alpha=1e-3, beta=1e-5,
                   hyperparameter_gamma=0.1, save_path=None)
    assert all(abs(a - b) < 2e-4 * b for a, b in zip(m2.train_loss + m2.test_loss, ref["train_loss"] + ref["test_loss"]))
    for k, v in ref["post"].items():
        assert rel_err(m2.state_dict()[k].cpu(), v) < 2e-3, k


# ----------------------------------------------------------------------------- N3: input pipeline
def test_prefetch_loader_delivers_dataset_batches_on_device(tmp_path):
    """Pinned staging + async copies on a side stream give the same batches as indexing the dataset (the reference's path)."""
    import numpy as np

    from learned_hologram_gan_amd.watermelon_hologram.data_loader import PrefetchLoader, dataloaderImgDepthAmpPhs

    N, C, H, W = 37, 3, 48, 64
    rng = np.random.default_rng(1)
    paths = {}
    for k in ("img", "depth", "amp", "phs"):
        paths[k] = str(tmp_path / f"{k}.bin")
        rng.random((N, C, H, W), dtype=np.float32).tofile(paths[k])
    ds = dataloaderImgDepthAmpPhs(paths["img"], paths["depth"], paths["amp"], paths["phs"], N, C, H, W, cuda=True)
    loader = PrefetchLoader(ds, batch_size=4, shuffle=False, drop_last=False, depth=2)
    seen = 0
    for epoch in range(2):
        i = 0
        for rgbd, amp, phs in loader:
            assert rgbd.is_cuda and rgbd.shape[1] == 4
            want = [torch.stack(t) for t in zip(*(ds[j] for j in range(i, min(i + 4, N))))]
            assert torch.equal(rgbd, want[0]) and torch.equal(amp, want[1]) and torch.equal(phs, want[2])
            i += rgbd.shape[0]
            seen += rgbd.shape[0]
    assert seen == 2 * N


# ----------------------------------------------------------------------------- bf16 operand mode (BASELINE configs[2], [4])
def test_bf16_mode_train_step_quality():
    """bf16-operand conv GEMMs (fp32 accumulation, tensors, FFTs): the reconstruction of one training step stays within PSNR bounds of the
    fp32 CPU oracle on identical inputs (BASELINE.json: 'recon PSNR vs ref'), and the mode switches back cleanly."""
    from learned_hologram_gan_amd import hip_ops
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    rows = cols = 64
    pad, coef, ratio = 32, 0.45, 1
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
    rgbd, tamp, tphs = seeded.smooth_batch(2, rows, cols, seed=78)
    idx = torch.tensor([5, 2])
    alphas = [torch.tensor([0.3, 0.8]).view(2, 1, 1, 1)]
    st = step.make_state(rows, cols, pad, coef, stack, seeded.generator_state_dict(), seeded.critic_state_dict())
    ref = step.train_step(st, rgbd, tamp, tphs, step.LossWeights(d_ratio=ratio), idx, alphas)

    def psnr(a, b):
        return (10 * torch.log10((b.max() - b.min()) ** 2 / torch.mean((a - b) ** 2))).item()

    hip_ops.set_conv_precision("bf16")
    try:
        W = watermelon(filter_radius_coefficient=coef, pad_size=pad, distance_stack=stack, input_shape=(1, 4, rows, cols))
        W.generator.load_state_dict(seeded.generator_state_dict())
        W.discriminator.load_state_dict(seeded.critic_state_dict())
        W.generator.to(DEV).train()
        W.discriminator.to(DEV).train()
        W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, ratio, 10)
        out = W.train_step(rgbd.to(DEV), tamp.to(DEV), tphs.to(DEV), idx, [a.to(DEV) for a in alphas])
    finally:
        hip_ops.set_conv_precision("default")
    p_amp = psnr(out["hat_amps"].cpu(), ref["hat_amps"])
    p_poh = psnr(torch.cos(out["POH"].cpu()), torch.cos(ref["POH"]))
    assert p_amp > 35.0 and p_poh > 30.0, (p_amp, p_poh)
    assert rel_err(out["target_amps"].cpu(), ref["target_amps"]) < PARITY  # no conv GEMM on the target path
    assert abs(out["G_loss"].item() - ref["G_loss"]) <= 5e-2 * abs(ref["G_loss"])
    assert abs(out["D_loss"].item() - ref["D_loss"]) <= 1e-1 * abs(ref["D_loss"])
    assert hip_ops.conv_precision() == hip_ops.default_precision()


# ----------------------------------------------------------------------------- the two entry points, end to end through files
def _write_bins(tmp_path, prefix, N, H, W, seed):
    import numpy as np

    rng = np.random.default_rng(seed)
    paths = {}
    for k in ("img", "depth", "amp", "phs"):
        paths[k] = str(tmp_path / f"{prefix}_{k}.bin")
        rng.random((N, 3, H, W), dtype=np.float32).tofile(paths[k])
    return paths


def test_training_and_generation_clis_end_to_end(tmp_path):
    """trainingModel.py (reference flags, PrefetchLoader input pipeline, checkpoints + metrics JSON) and generatePOH.py --propagate
    (checkpoint -> POH file -> PNG planes) on tiny synthetic .bin data sets."""
    import json
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import generatePOH
    import trainingModel

    H = W = 32
    tr, va = _write_bins(tmp_path, "train", 8, H, W, 1), _write_bins(tmp_path, "val", 100, H, W, 2)
    out = tmp_path / "out"
    argv = []
    for prefix, p in (("train", tr), ("validate", va)):
        for k in ("img", "depth", "amp", "phs"):
            argv += [f"--{prefix}_{k}_path", p[k]]
    argv += ["--samplesNum", "8", "--channlesNum", "3", "--height", str(H), "--width", str(W), "--batch_size", "4", "--epoch_num", "1",
             "--save_path_G", str(out / "G.pth"), "--save_path_D", str(out / "D.pth"), "--loss_metrics_file", str(out / "metrics.json"),
             "--save_path_img", str(out / "img")]
    trainingModel.train_gan(trainingModel.build_parser().parse_args(argv))
    sd = torch.load(out / "G.pth", map_location="cpu")
    assert len(sd) == len(seeded.generator_state_dict()) and all(torch.isfinite(v.float()).all() for v in sd.values())
    assert (out / "G_epoch0.pth").exists() and isinstance(json.load(open(out / "metrics.json")), dict)

    gen_args = generatePOH.build_parser().parse_args([
        "--img_path", tr["img"], "--depth_path", tr["depth"], "--index", "3", "--model_path", str(out / "G.pth"),
        "--poh_output_path", str(out / "poh.pt"), "--samplesNum", "8", "--sample_row_num", str(H), "--sample_col_num", str(W),
        "--pad_size", "16", "--propagate", "--num_intervals", "3", "--output_image_dir", str(out / "planes")])
    generatePOH.main(gen_args)
    poh = torch.load(out / "poh.pt", map_location="cpu")
    assert poh.shape == (3, H, W) and torch.isfinite(poh).all() and poh.abs().max() <= 1.5 * torch.pi + 1e-4
    assert sorted(os.listdir(out / "planes")) == ["0.png", "1.png", "2.png"]


# ----------------------------------------------------------------------------- data-parallel bench path, two ranks on one GPU
def test_two_rank_bench_rehearsal():
    """`python bench.py --gpus 2` launched PLAINLY, as the driver launches `--gpus 1` (no torchrun, WORLD_SIZE unset): the parent starts
    the two ranks itself before touching the GPU and relays rank 0's line.  Rehearsed on ONE GPU with the gloo backend (RCCL needs one
    GPU per rank): every rank must issue the same collectives — a rank-conditional train step deadlocks here."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LHG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "64",
           "--cols", "64", "--pad", "32", "--cpu-baseline", "0", "--secondary", "0", "--other-modes", "0"]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["value"] > 0 and out["config"]["global_batch"] == 8
    assert out["rccl_ranks"] is None  # gloo rehearsal: the line must not claim RCCL ranks it did not have
    assert set(out["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


def _run_dist_worker(mode, nproc, env_extra=None, timeout=280):
    import json
    import os
    import socket
    import subprocess
    import sys

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(env_extra or {}))
    worker = os.path.join(root, "tests", "_dist_gpu_worker.py")
    if nproc == 1:
        cmd = [sys.executable, worker, mode]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(port), worker, mode]
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-2500:])
    out, dec = [], json.JSONDecoder()
    for ln in res.stdout.splitlines():   # the ranks share the launcher's pipe: tolerate two records on one line
        pos = 0
        while ln.startswith("{", pos):
            rec, pos = dec.raw_decode(ln, pos)
            out.append(rec)
    return out


def test_rccl_backend_initialises_and_reduces_world1():
    """backend="nccl" is RCCL on ROCm: a world of one on this box's GPU must initialise, all-reduce (async, as GradSynchronizer
    issues it), broadcast and barrier.  (More than one rank needs one GPU per rank: the driver's 8-GPU run.)"""
    out = _run_dist_worker("rccl", 1)
    assert out and out[-1]["rccl"] is True and out[-1]["backend"] == "nccl"


def test_gradient_buckets_are_reduced_inside_backward_two_ranks():
    """Two ranks share this GPU over gloo.  Conv weight gradients never pass through autograd (accumulated into the flat buffer on the
    weight-gradient stream), yet every bucket except the last must have its all-reduce ENQUEUED from inside backward — the first one
    before most of the pass's weight-gradient contributions exist — and the reduced gradients equal the mean of the local ones."""
    # "reduced == mean of the local gradients" compares three passes over the same data, so it presumes passes that repeat.  Rounds 1-2
    # saw about one run in twenty with a pass that differed in a few bits; round 3 found the cause (packed-fp32 VALU
    # instructions of the angular-spectrum kernels next to MFMA workgroups of another queue) and the library is built without those
    # instructions, so a pass that does not repeat is a FAILURE here — no retry.  The worker's full record (forward checksums per pass,
    # which pixels / parameters differed, a 200-call determinism probe of the operator under the same contention) is appended to
    # gpurun_out/<tag>_two_rank_overlap.jsonl either way (stamped with the kernel-source hash).
    from conftest import record_path, record_stamp

    out = sorted(_run_dist_worker("overlap", 2), key=lambda r: r["rank"])
    with open(record_path("two_rank_overlap.jsonl"), "a") as f:
        for r in out:
            f.write(json.dumps({**record_stamp(), **r}) + "\n")
    assert [r["rank"] for r in out] == [0, 1]
    for r in out:
        assert r["forward_repeats"] and r["local_repeatable"], ("a pass did not repeat", r["forward"], r["recompute_notes"], r["diff_pass0_vs_pass2"], r["asm_probe"])
        assert not any(r["asm_probe"]["mismatches"].values()), r["asm_probe"]
        assert r["ranks_agree"], "reduced gradients differ between the ranks"
        assert r["err"] < 1e-5, f"reduced != mean of the local gradients (err {r['err']})"
        in_backward = [(b, c) for b, c, from_finish in r["launch_log"] if not from_finish]
        assert len(in_backward) >= r["buckets"] - 1, "buckets left to finish()"          # at most the first-layer bucket is left to finish()
        assert in_backward[0][1] < 0.5 * r["contributions"], "bucket 0 launched late"    # before half of the contributions were enqueued
        assert [b for b, _, _ in r["launch_log"]] == sorted(b for b, _, _ in r["launch_log"]), "bucket order"  # same order on every rank
    assert out[0]["launch_log"] == out[1]["launch_log"], "launch logs differ between the ranks"


def test_critic_step_with_gradient_penalty_two_ranks():
    """The full step (critic update with the gradient penalty, then the generator update) on two ranks sharing this GPU over gloo:
    reduced gradients = mean of the local ones for BOTH models (the critic's BatchNorm gammas collect slot contributions from three
    forwards and from the penalty's double backward: none may land after its bucket's all-reduce went out), and only the small first
    bucket (the critic's head conv and last BatchNorm are visited by the penalty's inner autograd.grad only) is left to finish()."""
    out = sorted(_run_dist_worker("critic", 2), key=lambda r: r["rank"])
    assert [r["rank"] for r in out] == [0, 1]
    for r in out:
        for name in ("D", "G"):
            m = r[name]
            assert m["ranks_agree"], (name, m)
            assert m["local_repeatable"], (name, m)
            assert m["err"] < 1e-5, (name, m)
            from_finish = [b for b, ff in m["launch_log"] if ff]
            assert len(m["launch_log"]) == m["buckets"] and set(from_finish) <= {0, m["buckets"] - 1}, (name, m)  # first (small) / first-layer bucket at most
            assert m["bucket_elems"][0] <= (1 << 18) or m["buckets"] == 1, (name, m)
    assert out[0]["D"]["launch_log"] == out[1]["D"]["launch_log"] and out[0]["G"]["launch_log"] == out[1]["G"]["launch_log"]


def test_synchronised_operators_two_ranks_equal_one_process():
    """The pieces of configure(sync_batch_stats=True) where the comparison is well conditioned: train-mode BatchNorm forward, backward
    and double backward, and the reconstruction losses — two ranks x half the batch against one process on the whole batch, to fp32
    rounding."""
    out = _run_dist_worker("syncops", 2)
    assert len(out) == 2
    for r in out:
        for tag in ("bn", "bn_leaky"):
            for k, v in r[tag].items():
                assert v < (2e-5 if k in ("dx", "dgamma", "gx") else 5e-6), (tag, k, r)
        for k, v in r["recon"].items():
            assert v < 1e-5, (k, r)


def test_sync_batch_stats_two_ranks_equal_one_process_at_twice_the_batch():
    """configure(sync_batch_stats=True): 2 ranks x 2 samples (sharing this GPU over gloo) against one process at 4 samples — the full
    step with the critic's gradient penalty.  Global BatchNorm statistics in forward, backward and double backward, the focal loss's
    global max normalisers and the global TV means make the averaged gradients of both models, the reconstructions, the losses and the
    running statistics those of the single process (ref: the single-device step, watermelon.py:207-277; loss_func.py:94-98, 152-157)."""
    # LHG_AUTOTUNE=0: both ranks also run the SINGLE-PROCESS step and must agree on it bit for bit — since ABI 10 the batch statistics are
    # folded from the conv epilogues' partial rows, whose grouping follows the GEMM's tiling, and two processes that each time the tilings
    # themselves may pick differently (1-ulp statistics).  The fixed heuristic (or a shared LHG_TUNE_CACHE) makes the choice the same.
    out = {r["rank"]: r for r in _run_dist_worker("syncbn", 2, env_extra={"LHG_AUTOTUNE": "0"})}
    r0 = out[0]
    from conftest import record_path, record_stamp

    with open(record_path("syncbn.jsonl"), "a") as f:
        f.write(json.dumps({**record_stamp(), **r0}) + "\n")
    agree = r0["single_process_step_agrees_between_ranks"]
    # the two ranks' SINGLE-PROCESS reference steps are identical work without collectives: they must agree bit for bit (until round 3 the
    # non-repeat of DESIGN.md §5 could hit them; its cause is removed from the build, so a disagreement fails the test)
    assert all(v for k, v in agree.items() if not k.endswith("_damage")), agree
    assert r0["hat_err"] < 2e-5 and r0["bn_err"] < 1e-5, r0
    # gradients behind 18 / 5 train-mode BatchNorm backwards at 64x64 (4x4 pixels in the deepest layers) are ill conditioned: two fp32
    # evaluations in different summation orders differ by ~1e-2 in L2 (noise-like: every parameter's norm ratio is 1.000; the operators
    # themselves agree to 1e-5, test_synchronised_operators_two_ranks_equal_one_process); WITHOUT the synchronisation the error is O(1)
    assert r0["g_err"] < 3e-2 and r0["d_err"] < 3e-2, r0
    assert all(abs(ratio - 1) < 2e-2 for m in ("G", "D") for _, _, ratio in r0["worst"][m]), r0
    # ... and that floor is MEASURED in the same worker: the single-process step again with the samples in reverse order —
    # the same mathematics in another summation order.  The synchronised run may be off by a small multiple of it, no more: a dropped
    # or mis-scaled term of the synchronised backward would be O(1 / W) of a gradient, far above the floor.
    assert r0["g_err"] <= 4.0 * r0["floor_g"] + 1e-4 and r0["d_err"] <= 4.0 * r0["floor_d"] + 1e-4, (r0["g_err"], r0["floor_g"], r0["d_err"], r0["floor_d"])
    for got, want in zip(r0["losses"], r0["ref_losses"]):
        assert abs(got - want) <= 2e-4 * abs(want) + 1e-7, r0


def test_train_step_repeats_bit_for_bit():
    """One process, one GPU: the same step on the same weights and data gives bit-identical losses, holograms, reconstructions and
    gradients three times in a row — with the weight-gradient GEMMs on the second stream, and with the caching allocator's free blocks
    overwritten by NaNs in between (no kernel reads memory that was not written for it; tools/poison_probe.py)."""
    from learned_hologram_gan_amd import hip_ops
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    rows = cols = 64
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
    W = watermelon(filter_radius_coefficient=0.45, pad_size=32, distance_stack=stack, input_shape=(1, 4, rows, cols))
    W.generator.load_state_dict(seeded.generator_state_dict())
    W.discriminator.load_state_dict(seeded.critic_state_dict())
    W.generator.to(DEV).train()
    W.discriminator.to(DEV).train()
    W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10)
    rgbd, tamp, tphs = seeded.smooth_batch(2, rows, cols, seed=91)
    x = (rgbd.to(DEV), tamp.to(DEV), tphs.to(DEV), torch.tensor([5, 2]), [torch.tensor([0.3, 0.8]).view(2, 1, 1, 1).to(DEV)])
    grabbed = []
    for opt in (W._opt_D, W._opt_G):
        def step(grad_scale=1.0, opt=opt):  # capture instead of Adam: every pass sees the same weights
            hip_ops.join_side_stream()
            torch.cuda.synchronize()
            grabbed.append(opt.flat.grad.detach().clone())
        opt.step = step
    runs = []
    for k in range(4):
        if k == 2:  # poison the allocator's cache: stale NaNs everywhere a kernel might look without having been written for
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            junk = torch.full((1 << 28,), float("nan"), device=DEV)
            torch.cuda.synchronize()
            del junk
        out = W.train_step(*x)
        runs.append([out["POH"].clone(), out["hat_amps"].clone(), out["G_loss"].clone(), out["D_loss"].clone(), grabbed[-2], grabbed[-1]])
    for k in (2, 3):  # (run 0 is the first sight of every geometry)
        for a, b in zip(runs[1], runs[k]):
            assert torch.equal(a, b)


def test_bench_size_training_repeats_bit_for_bit():
    """The headline workload (384 x 384, batch 4, 1024-point transforms): ten optimiser steps from the same state, twice — every step's seven
    losses and the final weights of both models are bit-identical.  This is the configuration in which the angular-spectrum kernels run
    beside MFMA weight-gradient workgroups of the second stream in ONE process, the condition of DESIGN.md §5's non-repeat."""
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    def run():
        torch.manual_seed(11)
        W = watermelon(filter_radius_coefficient=0.45, pad_size=320, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, 384, 384))
        W.generator.to(DEV).train()
        W.discriminator.to(DEV).train()
        W.configure(1, 0.0, 1, 1e-3, 1e-1, 1e-3, 1e-3, 1, 10)
        g = torch.Generator().manual_seed(5)
        rgbd, tamp, tphs = (torch.rand((4, c, 384, 384), generator=g).to(DEV) for c in (4, 3, 3))
        alphas = [torch.rand((4, 1, 1, 1), generator=g).to(DEV)]
        losses = []
        for k in range(10):
            W.train_step(rgbd, tamp, tphs, plane_indices=torch.tensor([(3 * k) % 20, (7 * k + 1) % 20, 5, 11]), gp_alphas=alphas)
            losses.append(W.train_losses_tensor.clone())
        torch.cuda.synchronize()
        flat = [torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone() for m in (W.generator, W.discriminator)]
        del W
        return torch.stack(losses), flat

    la, fa = run()
    lb, fb = run()
    assert torch.isfinite(la).all()
    assert torch.equal(la, lb), (la - lb).abs().max(dim=1).values
    assert torch.equal(fa[0], fb[0]) and torch.equal(fa[1], fb[1])


# ----------------------------------------------------------------------------- second stream for the weight gradients
def test_side_stream_weight_gradients_match_single_stream():
    """Gradients of a training step with the weight-gradient GEMMs on the second stream (accumulated straight into the flat gradient
    slots) against the same step on one stream through autograd's own accumulation, captured right before each optimiser step:
    the generator's are identical (one contribution per weight), the critic's agree up to the summation order of its three
    contributions.  Run twice to give a stream race a chance to show."""
    from learned_hologram_gan_amd import hip_ops
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    rows = cols = 64
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
    rgbd, tamp, tphs = seeded.smooth_batch(2, rows, cols, seed=90)
    idx = torch.tensor([5, 2])
    alphas = [torch.tensor([0.3, 0.8]).view(2, 1, 1, 1).to(DEV)]

    def run(side):
        hip_ops.SIDE_WGRAD = hip_ops.SLOT_ACCUMULATE = side
        W = watermelon(filter_radius_coefficient=0.45, pad_size=32, distance_stack=stack, input_shape=(1, 4, rows, cols))
        W.generator.load_state_dict(seeded.generator_state_dict())
        W.discriminator.load_state_dict(seeded.critic_state_dict())
        W.generator.to(DEV).train()
        W.discriminator.to(DEV).train()
        W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10)
        grads = {}
        for name, opt in (("D", W._opt_D), ("G", W._opt_G)):
            real_step = opt.step

            def step(grad_scale=1.0, name=name, opt=opt, real_step=real_step):
                hip_ops.join_side_stream()
                torch.cuda.synchronize()
                grads[name] = opt.flat.grad.detach().cpu().clone()
                real_step(grad_scale=grad_scale)

            opt.step = step
        W.train_step(rgbd.to(DEV), tamp.to(DEV), tphs.to(DEV), idx, alphas)
        torch.cuda.synchronize()
        return grads

    keep = hip_ops.SIDE_WGRAD
    try:
        for _ in range(2):
            g1, g0 = run(True), run(False)
            assert torch.equal(g1["G"], g0["G"])
            assert (g1["D"] - g0["D"]).norm() <= 1e-5 * g0["D"].norm()
            assert g0["G"].abs().max() > 0 and g0["D"].abs().max() > 0
    finally:
        hip_ops.SIDE_WGRAD, hip_ops.SLOT_ACCUMULATE = keep, True


# ----------------------------------------------------------------------------- the C ABI without Python
def test_standalone_cpp_host_of_the_c_abi():
    """examples/abi_demo (built by __graft_entry__.build()): a C++ program that links liblhg_hip.so, owns its buffers and stream, and checks
    a convolution against a scalar loop and the angular-spectrum operator against the identity."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "abi_demo")
    assert os.path.exists(exe), "run python __graft_entry__.py first"
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "abi_demo ok" in res.stdout, res.stdout + res.stderr


# ----------------------------------------------------------------------------- does it learn
def test_training_reduces_the_loss_and_raises_psnr():
    """60 steps of the reconstruction-only trainer (watermelon_without_GAN, as shipped in trainingModel.py) on four smooth synthetic
    batches: the generator loss falls and the reconstruction PSNR rises (measured: 0.30 -> 0.11, 14.9 -> 18.8 dB)."""
    from learned_hologram_gan_amd.poh_ops import psnr_ssim
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon_without_GAN

    R = 64
    torch.manual_seed(0)
    W = watermelon_without_GAN(filter_radius_coefficient=0.45, pad_size=R // 2, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1],
                               input_shape=(1, 4, R, R))
    W.generator.to(DEV).train()
    W.configure(1, 0.0, 1, 1e-3, 1e-1, 1e-3, 1e-3, 0, 10)
    batches = [tuple(t.to(DEV) for t in seeded.smooth_batch(4, R, R, seed=200 + i)) for i in range(4)]
    first = last = None
    for it in range(60):
        out = W.train_step(*batches[it % 4])
        if it < 4 or it >= 56:
            rec = (out["G_loss"].item(), psnr_ssim(out["hat_amps"], out["target_amps"])[0].item())
            first = rec if it == 0 else first
            last = rec
    assert all(map(lambda v: v == v, first + last))  # no NaN
    assert last[0] < 0.6 * first[0] and last[1] > first[1] + 2.0, (first, last)
