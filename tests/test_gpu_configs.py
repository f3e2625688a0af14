"""GPU parity on the workloads BASELINE.json names (configs[1]..[4]) at their full sizes, and the reference's only known-answer
artefact (`poh.pt` -> ten PNGs, README.md:123-156) through the HIP path.  Everything goes through the C ABI; the CPU oracle
(`oracle/`) is the checker.  fp32: north_star's 1e-4 where the quantity is well conditioned, otherwise the bound the fp64-truth
tests support; bf16 mode: reconstruction PSNR against the fp32 oracle (BASELINE.json "recon PSNR vs ref").
"""

import contextlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, phase_err, rel_err
from oracle import losses, nets, optics, seeded

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
WL = torch.tensor([638e-9, 520e-9, 450e-9])
PITCH = 3.74e-6
PARITY = 1e-4


def _multi(r0, c0, stack, pad, coef):
    from learned_hologram_gan_amd.angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_multiple_distances as Mu

    return Mu(r0, c0, stack, pad, coef, PITCH, WL, False, True)


def _psnr(a, b):
    return (10 * torch.log10((b.max() - b.min()) ** 2 / torch.mean((a - b) ** 2))).item()


# ----------------------------------------------------------------------------- the reference's known-answer artefact
def test_known_answer_terminal_test_pngs_on_the_hip_path():
    """generatePOH.py --propagate --num_intervals 10 on the committed hologram (ref: generatePOH.py:51-78): the multi-distance
    propagator's ``__call__`` at the CLI defaults (pad 320 -> 1024^2 transforms on the LDS FFT, filter 0.35, distances
    linspace(4e-4, 10e-4, 10)), ``tensor_normalizor_2D``, x255 truncated to 8 bit -> the reference's ten PNGs, never more than one grey
    level apart.  The transfer functions are built by the product on THIS host (no injected constants)."""
    PIL = pytest.importorskip("PIL.Image")
    from learned_hologram_gan_amd import utilities

    kat = os.path.join(GOLDEN, "kat_terminalTest")
    poh = torch.load(os.path.join(kat, "poh.pt"), map_location="cpu", weights_only=False)
    assert poh.shape == (3, 384, 384)
    poh = poh.unsqueeze(0).float().to(DEV)
    d = torch.linspace(4e-4, 10e-4, 10)
    prop = _multi(384, 384, d, 320, 0.35)
    assert prop._geom.supported()
    with torch.no_grad():
        amp = prop(torch.ones_like(poh), poh, d)
    assert amp.shape == (10, 3, 384, 384)
    img = (utilities.tensor_normalizor_2D(amp) * 255.0).permute(0, 2, 3, 1).cpu().numpy()
    for k in range(10):
        png = np.asarray(PIL.open(os.path.join(kat, f"{k}.png")).convert("RGB"), dtype=np.float64)
        quant = np.floor(img[k].astype(np.float64))  # the plotter truncates x*255 to 8 bit
        assert np.abs(quant - png).max() <= 1.0, k
        assert (quant != png).mean() < 0.02, k  # > 98 % of the pixels bit-identical (the fixture was made on another device and host)


# ----------------------------------------------------------------------------- configs[1] / configs[2]: 384^2, batch 4 per rank
@contextlib.contextmanager
def _bf16(kind):
    """"operands": fp32 tensors, conv-GEMM operands rounded to bf16.  "storage": NHWC activations and their gradients bf16 in HBM as well
    (BatchNorm / loss / FFT / Adam arithmetic and every parameter stay fp32)."""
    from learned_hologram_gan_amd import hip_ops

    if kind == "storage":
        hip_ops.set_activation_storage("bf16")
    elif kind == "fp32":
        hip_ops.set_conv_precision("default")  # "fp32" = fp32 tensors in the SHIPPED default GEMM mode (fp32_split_f16), not the exact kernels
    else:
        hip_ops.set_conv_precision("bf16")
    try:
        yield
    finally:
        if kind == "storage":
            hip_ops.set_activation_storage("fp32")
        hip_ops.set_conv_precision("default")


@pytest.mark.parametrize("kind", ["operands", "storage"])
def test_config2_per_rank_bf16_train_step_384_bs4(oracle_full_step, kind):
    """BASELINE configs[2], one rank's share: the 384x384 batch-4 GAN train step in the bf16 modes (bf16 conv-GEMM operands, and bf16
    activation storage; fp32 accumulation, fp32 BatchNorm / losses / FFT / Adam) against the fp32 CPU oracle on identical inputs:
    reconstruction PSNR."""
    from learned_hologram_gan_amd import hip_ops
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    cfg, ref = oracle_full_step
    B = cfg["rgbd"].shape[0]
    with _bf16(kind):
        W = watermelon(filter_radius_coefficient=cfg["coef"], pad_size=cfg["pad"], distance_stack=cfg["stack"],
                       input_shape=(1, 4, cfg["rows"], cfg["cols"]))
        W.generator.load_state_dict(seeded.generator_state_dict())
        W.discriminator.load_state_dict(seeded.critic_state_dict())
        W.generator.to(DEV).train()
        W.discriminator.to(DEV).train()
        W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10)
        out = W.train_step(cfg["rgbd"].to(DEV), cfg["tamp"].to(DEV), cfg["tphs"].to(DEV), cfg["idx"], [a.to(DEV) for a in cfg["alphas"]])
        got = dict(zip(("focal_phase_gradient_loss", "perceptual_loss", "pixel_loss", "TV_loss", "gan_loss", "G_loss", "D_loss"),
                       W.train_losses_tensor.tolist()))
    assert out["hat_amps"].shape == (B, 3, cfg["rows"], cfg["cols"])
    p_amp = _psnr(out["hat_amps"].cpu(), ref["hat_amps"])
    p_poh = _psnr(torch.cos(out["POH"].cpu()), torch.cos(ref["POH"]))
    assert (p_amp > 35.0 and p_poh > 28.0) if kind == "operands" else (p_amp > 30.0 and p_poh > 24.0), (p_amp, p_poh)
    assert rel_err(out["target_amps"].cpu(), ref["target_amps"]) < PARITY  # no conv GEMM on the target path
    tol = 5e-2 if kind == "operands" else 1e-1
    for k in ("focal_phase_gradient_loss", "pixel_loss", "TV_loss"):
        assert abs(got[k] - ref[k]) <= tol * abs(ref[k]) + 1e-6, (k, got[k], ref[k])
    assert abs(got["D_loss"] - ref["D_loss"]) <= 2 * tol * abs(ref["D_loss"]) + 1e-6
    assert hip_ops.conv_precision() == hip_ops.default_precision() and hip_ops.activation_storage() == "fp32"


# ----------------------------------------------------------------------------- configs[4]: batch 1 per rank, 3-plane reconstruction loss
@pytest.mark.parametrize("precision", ["fp32", "bf16", "bf16_storage"])
def test_config4_per_rank_bs1_three_plane_reconstruction_loss_384(precision):
    """BASELINE configs[4], one rank's share: batch 1 at 384x384 (pad 320 -> 1024^2 fp32 FFTs), the hologram and the target propagated
    to ALL planes of a 3-plane stack (``..._all_fixed_multiple_distances_freq2amp``, ref: angular_spectrum_method.py:524-531) and the
    reconstruction loss (focal phase-gradient + MSE + 1e-3 TV, ref: watermelon.py:418-445) back-propagated into the generator."""
    from learned_hologram_gan_amd import hip_ops
    from learned_hologram_gan_amd.poh_ops import ReconLossFn
    from learned_hologram_gan_amd.watermelon_hologram.generator import Generator

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    rows = cols = 384
    pad, coef = 320, 0.45
    stack = torch.linspace(-4e-4, 0.0, 4)[:-1]  # three planes
    rgbd, tamp, tphs = seeded.smooth_batch(1, rows, cols, seed=61)

    # oracle (fp32, CPU)
    o = optics.make_optics(rows, cols, pad, coef, PITCH, WL)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    Hs = optics.transfer_function(o.w, stack)
    sd = nets.as_parameters(seeded.generator_state_dict())
    poh_ref = nets.generator(sd, o, Hf, rgbd, True)
    G_ref = torch.cat((optics.poh_to_filtered_spectrum(o, Hf, poh_ref), optics.target_to_filtered_spectrum(o, tamp, tphs)), 0)
    a_ref, p_ref = optics.spectrum_to_planes_all(o, Hs, G_ref)
    n = 3
    terms_ref = (losses.focal_sincos_phase_gradient_loss(p_ref[:n], p_ref[n:]), losses.pixel_loss(a_ref[:n], a_ref[n:]),
                 losses.total_variation_loss(a_ref[:n], a_ref[n:]))
    (terms_ref[0] + terms_ref[1] + 1e-3 * terms_ref[2]).backward()

    with _bf16({"fp32": "fp32", "bf16": "operands", "bf16_storage": "storage"}[precision]):
        G = Generator(rows, cols, pad, coef, 3, PITCH, WL, torch.tensor([1e-3]))
        G.load_state_dict(seeded.generator_state_dict())
        G.to(DEV).train()
        mu = _multi(rows, cols, stack, pad, coef)
        poh = G(rgbd.to(DEV))
        spectra = torch.cat((G.part2.propagator.propagate_POH2Freq_forward(poh), mu.filter_AP2filteredFreq(tamp.to(DEV), tphs.to(DEV))), 0)
        amps, phss = mu.propagate_multiple_samples_with_all_fixed_multiple_distances_freq2amp(spectra)
        assert amps.shape == (2 * n, 3, rows, cols)
        focal, mse, tv = ReconLossFn.apply(amps[:n].contiguous(), amps[n:].contiguous(), phss[:n].contiguous(), phss[n:].contiguous()).unbind(0)
        (focal + mse + 1e-3 * tv).backward()
        torch.cuda.synchronize()
    named = dict(G.named_parameters())
    probe = ("part1.part1.decoder4.0.convolution_layer_2.weight", "part1.part1.bottleneck.1.0.convolution_layer_1.weight",
             "part1.part1.encoder1.0.0.convolution_layer_1.weight", "part2.part1.conv_g.params")
    assert rel_err(amps[n:].cpu(), a_ref[n:]) < PARITY  # the target planes never see a conv GEMM
    if precision == "fp32":
        # north_star's 1e-4 against the CPU oracle (round 4; was 1e-3): in L2 with a wide margin; in the max norm this batch-1 geometry
        # (23 train-mode BatchNorms over ONE sample) puts two fp32 evaluations 1.18e-4 apart at its worst pixel (measured, GPUTEST r04),
        # so the max norm is bounded at 2e-4 — the distance to a float64 evaluation is what tests/test_gpu_truth.py bounds
        got_a, ref_a = amps[:n].detach().cpu().double(), a_ref[:n].detach().double()
        e_amp, e_l2 = rel_err(got_a, ref_a), ((got_a - ref_a).norm() / ref_a.norm()).item()
        assert e_l2 < 2e-5 and e_amp < 2e-4, (e_l2, e_amp)
        for got, ref in zip((focal, mse, tv), terms_ref):
            assert abs(got.item() - ref.item()) <= 1e-3 * abs(ref.item()) + 1e-7
        for k in probe:
            g, r = named[k].grad.cpu().double(), sd[k].grad.double()
            assert ((g - r).norm() / r.norm()).item() < 2e-2, k
    else:
        storage = precision == "bf16_storage"
        assert _psnr(amps[:n].detach().cpu(), a_ref[:n].detach()) > (30.0 if storage else 35.0)
        for got, ref in zip((focal, mse, tv), terms_ref):
            assert abs(got.item() - ref.item()) <= (1e-1 if storage else 5e-2) * abs(ref.item()) + 1e-6
        for k in probe:
            g, r = named[k].grad.cpu().double(), sd[k].grad.double()
            assert (torch.dot(g.flatten(), r.flatten()) / (g.norm() * r.norm())).item() > (0.8 if storage else 0.9), k  # direction of the update (bf16 operand rounding through up to 27 conv layers and batch-1 BatchNorm)


# ----------------------------------------------------------------------------- configs[3]: one 4K frame end to end
def test_multi_distance_call_shares_the_first_pass_between_the_distances():
    """``__call__`` (angular_spectrum_method.py:503-522: every field to D planes) runs the first pass — polar -> complex, row transforms —
    once per FIELD and lets the D filtered column / inverse passes read it by index (lhg_asm_propagate_shared, round 5).  Same kernels on
    the same values: bit-identical to the form that materialises the (B D, 3, h, w) copies, which a call that records gradients still takes."""
    for (h, w, pad, B, D) in ((64, 96, 32, 2, 5), (192, 192, 160, 1, 3), (48, 80, 8, 3, 1)):
        d = torch.linspace(4e-4, 10e-4, D)
        prop = _multi(h, w, d, pad, 0.35)
        g = torch.Generator().manual_seed(h + D)
        amp, phs = torch.rand((B, 3, h, w), generator=g).to(DEV), (torch.rand((B, 3, h, w), generator=g) * 6.28).to(DEV)
        with torch.no_grad():
            shared = prop(amp, phs, d)
        with torch.enable_grad():
            copies = prop(amp.clone().requires_grad_(True), phs, d)
        assert shared.shape == copies.shape == (B * D, 3, h, w)
        assert torch.equal(shared, copies.detach()), (h, w, pad, B, D, (shared - copies.detach()).abs().max().item())


def test_config3_4k_generator_tail_and_eight_plane_propagation():
    """BASELINE configs[3]: one 3840x2160 frame through the eval-mode Generator (pad 72 -> 2304x4096 transforms) and the hologram
    propagated to 8 planes by ``__call__`` (generatePOH.py --propagate).  The UNet is checked on a crop by
    test_config4_4k_frame_unet_locality (it is a local operator); here the GPU UNet's own (amp, phase) output is handed to the CPU
    oracle, which must reproduce the GPU's hologram (back-propagation + symmetric stencil + double-phase encode at full 4K extent) and
    the 8 reconstructed planes."""
    from learned_hologram_gan_amd.watermelon_hologram.generator import Generator

    torch.set_num_threads(min(16, os.cpu_count() or 1))
    H, W, pad = 2160, 3840, 72
    G = Generator(H, W, pad, 0.45, 3, PITCH, WL, torch.tensor([1e-3]))
    G.load_state_dict(seeded.generator_state_dict())
    G.to(DEV).eval()
    g = torch.Generator().manual_seed(43)
    rgbd = torch.nn.functional.interpolate(torch.rand((1, 4, H // 8, W // 8), generator=g), size=(H, W), mode="bilinear", align_corners=False)
    d = torch.linspace(4e-4, 10e-4, 8)
    prop = _multi(H, W, d, pad, 0.35)
    assert prop._geom.supported() and (prop.samplingRowNum, prop.samplingColNum) == (2304, 4096)
    with torch.no_grad():
        amp, phs = G.part1(rgbd.to(DEV))
        poh = G(rgbd.to(DEV))
        planes = prop(torch.ones_like(poh), poh, d)
    assert poh.shape == (1, 3, H, W) and planes.shape == (8, 3, H, W) and torch.isfinite(planes).all()
    o = optics.make_optics(H, W, pad, 0.45, PITCH, WL)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    sd = nets.as_parameters(seeded.generator_state_dict())
    with torch.no_grad():
        poh_ref = nets.amp_phase_to_poh(sd, o, Hf, amp.cpu(), phs.cpu())
    # angle() is ill-conditioned where the field is small: the bulk tightly, the worst pixel loosely (as the 384^2 step test)
    perr = (torch.exp(1j * poh.cpu()) - torch.exp(1j * poh_ref)).abs().flatten()
    assert torch.quantile(perr[::53], 0.999) < 2e-4 and perr.max() < 5e-2, (torch.quantile(perr[::53], 0.999).item(), perr.max().item())  # (round 4: 1e-3)
    del o, Hf
    o35 = optics.make_optics(H, W, pad, 0.35, PITCH, WL)
    with torch.no_grad():
        for k in (0, 7):  # first and last plane (each is 3 colour transforms of 2304x4096 on the CPU)
            ref = optics.propagate_amplitudes(o35, torch.ones_like(poh_ref), poh.cpu(), d[k:k + 1])
            assert rel_err(planes[k:k + 1].cpu(), ref) < PARITY, k
