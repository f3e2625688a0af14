"""Oracle: the assembled GAN training step.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pure torch, CPU, fp32.

``watermelon.__init__`` cannot be constructed offline (VGG19 download, torchmetrics,
hard-coded cuda), so the step is assembled from the parts in the order of
ref: watermelon_hologram/watermelon.py:207-277, with the perceptual (VGG19) term left
out.  The two random draws of the reference step — the plane
permutation (angular_spectrum_method.py:536) and the interpolation factors of the
gradient penalty (watermelon.py:459) — are arguments so that the implementation under
test can be driven with the same values.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import torch

from . import losses, nets, optics


@dataclass
class LossWeights:
    """Defaults of the shipped CLI. ref: trainingModel.py:77-95."""
    phs_gradient: float = 1.0
    perceptual: float = 0.0  # reference CLI: 0.1 (VGG19, SURVEY §8f N1 — not built yet)
    pixel: float = 1.0
    tv: float = 1e-3
    discriminator: float = 0.1
    gp_lambda: float = 10.0
    d_ratio: int = 5


@dataclass
class TrainState:
    o: optics.Optics
    H_fixed: torch.Tensor  # (3,R,C)
    H_stack: torch.Tensor  # (D,3,R,C)
    sd_G: dict
    sd_D: dict
    lr_G: float = 1e-3
    lr_D: float = 1e-3
    opt_G: torch.optim.Optimizer = field(default=None)
    opt_D: torch.optim.Optimizer = field(default=None)

    def __post_init__(self):
        # ref: watermelon.py:137-138 — Adam with default betas/eps on all parameters.
        self.opt_G = torch.optim.Adam(nets.parameters_of(self.sd_G), lr=self.lr_G)
        self.opt_D = torch.optim.Adam(nets.parameters_of(self.sd_D), lr=self.lr_D)


def make_state(rows0, cols0, pad, coefficient, distance_stack, sd_G, sd_D,
               z_fixed=1e-3, pitch=optics.DEFAULT_PITCH, wave_length=None, lr_G=1e-3, lr_D=1e-3, consts=None):
    """Constants as ``watermelon.__init__`` builds them. ref: watermelon.py:46-82,
    trainingModel.py:59-67 (filter 0.45, pad 320, linspace(-4e-4,0,21)[:-1])."""
    o = optics.make_optics(rows0, cols0, pad, coefficient, pitch, wave_length)
    H_fixed = optics.transfer_function(o.w, torch.tensor([z_fixed]))[0]
    H_stack = optics.transfer_function(o.w, distance_stack)
    if consts is not None:  # transfer functions recorded with a fixture (they are host-CPU dependent)
        H_fixed, H_stack = consts["H_fixed"], consts["H_stack"]
    return TrainState(o, H_fixed, H_stack, nets.as_parameters(sd_G), nets.as_parameters(sd_D), lr_G, lr_D)


# --------------------------------------------------------------------------- A11
def gradient_penalty(sd_D, real, fake, alpha):
    """mean_b (||d sum(D(x^)) / d x^||_2 - 1)^2 at x^ = alpha*real + (1-alpha)*fake,
    differentiable w.r.t. the critic parameters (create_graph=True).
    ref: watermelon.py:458-477."""
    x_hat = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
    score = nets.critic(sd_D, x_hat, True)
    (g,) = torch.autograd.grad(score, x_hat, torch.ones_like(score), create_graph=True, retain_graph=True)
    g = g.view(g.size(0), -1)
    return ((g.norm(2, dim=1) - 1) ** 2).mean()


def generator_loss(hat_amps, target_amps, hat_phs, target_phs, adversarial, w: LossWeights):
    """ref: watermelon.py:418-445 (G_loss) minus the perceptual term."""
    terms = {
        "focal_phase_gradient_loss": losses.focal_sincos_phase_gradient_loss(hat_phs, target_phs) * w.phs_gradient,
        "pixel_loss": losses.pixel_loss(hat_amps, target_amps) * w.pixel,
        "TV_loss": losses.total_variation_loss(hat_amps, target_amps) * w.tv,
        "gan_loss": adversarial * w.discriminator,
    }
    terms["G_loss"] = sum(terms.values())
    return terms


def reconstruct(st: TrainState, rgbd, target_amp, target_phs, plane_indices):
    """G forward + amplitude/phase of hat and target at one plane per sample.
    ref: watermelon.py:216-241."""
    B = rgbd.size(0)
    poh = nets.generator(st.sd_G, st.o, st.H_fixed, rgbd, True)
    hat_freq = optics.poh_to_filtered_spectrum(st.o, st.H_fixed, poh)
    tgt_freq = optics.target_to_filtered_spectrum(st.o, target_amp, target_phs)
    amps, phss = optics.spectrum_to_planes_indexed(st.o, st.H_stack, torch.cat((hat_freq, tgt_freq), 0), plane_indices)
    return poh, amps[:B], amps[B:], phss[:B], phss[B:]


# --------------------------------------------------------------------------- A12
def _grads_of(sd):
    return {k: v.grad.detach().clone() for k, v in sd.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}


def _weights_of(sd):
    return {k: v.detach().clone() for k, v in sd.items() if isinstance(v, torch.Tensor) and v.requires_grad}


def train_step(st: TrainState, rgbd, target_amp, target_phs, w: LossWeights, plane_indices, gp_alphas, capture: bool = False):
    """One batch of ``watermelon.train``. ref: watermelon.py:207-277.

    Returns a dict of python floats (losses) plus the tensors needed for parity checks.  ``capture``: also the per-parameter gradients
    of both models as the optimisers see them — ``grads_D`` (one dict per critic update, taken between ``d_loss.backward()`` and
    ``optimizer_D.step()``, ref: watermelon.py:252-256) and ``grads_G`` (between ``G_loss.backward()`` and ``optimizer_G.step()``,
    ref: watermelon.py:275-277) — and the post-Adam weights ``weights_D`` / ``weights_G``.
    """
    poh, hat_amps, target_amps, hat_phs, target_phs_z = reconstruct(st, rgbd, target_amp, target_phs, plane_indices)

    d_losses, grads_D = [], []
    for it in range(w.d_ratio):
        real_v = nets.critic(st.sd_D, target_amps, True)
        fake_v = nets.critic(st.sd_D, hat_amps.detach(), True)
        gp = gradient_penalty(st.sd_D, target_amps, hat_amps.detach(), gp_alphas[it])
        d_loss = (-real_v.mean() + fake_v.mean()) + w.gp_lambda * gp
        st.opt_D.zero_grad()
        d_loss.backward(retain_graph=True)
        if capture:
            grads_D.append(_grads_of(st.sd_D))
        st.opt_D.step()
        d_losses.append((d_loss.item(), gp.item()))

    adversarial = -nets.critic(st.sd_D, hat_amps, True).mean()
    terms = generator_loss(hat_amps, target_amps, hat_phs, target_phs_z, adversarial, w)
    st.opt_G.zero_grad()
    terms["G_loss"].backward()
    grads_G = _grads_of(st.sd_G) if capture else None
    st.opt_G.step()

    out = {k: v.item() for k, v in terms.items()}
    if capture:
        out.update(grads_D=grads_D, grads_G=grads_G, weights_D=_weights_of(st.sd_D), weights_G=_weights_of(st.sd_G))
    out["D_loss"] = sum(d for d, _ in d_losses) / max(w.d_ratio, 1)
    out["d_iters"] = d_losses
    out["POH"] = poh.detach()
    out["hat_amps"] = hat_amps.detach()
    out["target_amps"] = target_amps.detach()
    out["PSNR"] = float(losses.psnr(hat_amps.detach(), target_amps.detach()))
    return out


# --------------------------------------------------------------------------- A14
def validate(st: TrainState, batches, w: LossWeights):
    """``_validate_generator``: eval-mode G and D (running BN statistics), every sample propagated to every plane of the
    stack, loss terms and PSNR / SSIM averaged over the batches.  ref: watermelon.py:479-552."""
    sums = {k: 0.0 for k in ("focal_phase_gradient_loss", "pixel_loss", "TV_loss", "gan_loss", "G_loss", "PSNR", "SSIM")}
    with torch.no_grad():
        for rgbd, target_amp, target_phs in batches:
            poh = nets.generator(st.sd_G, st.o, st.H_fixed, rgbd, False)
            G = torch.cat((optics.poh_to_filtered_spectrum(st.o, st.H_fixed, poh),
                           optics.target_to_filtered_spectrum(st.o, target_amp, target_phs)), 0)
            amps, phss = optics.spectrum_to_planes_all(st.o, st.H_stack, G)
            n = rgbd.size(0) * st.H_stack.size(0)
            adversarial = -nets.critic(st.sd_D, amps[:n], False).mean()
            terms = generator_loss(amps[:n], amps[n:], phss[:n], phss[n:], adversarial, w)
            for k, v in terms.items():
                sums[k] += v.item()
            sums["PSNR"] += losses.psnr(amps[:n], amps[n:]).item()
            sums["SSIM"] += losses.ssim(amps[:n], amps[n:]).item()
    return {k: v / max(len(batches), 1) for k, v in sums.items()}
