"""Reconstruction losses of the GAN step. ref: learnedMethodForHologram/watermelon_hologram/loss_func.py:66-208.
Pointwise / reduction work on (B,3|6,H,W) tensors (a few MB): expressed with device tensor ops."""

from __future__ import annotations

import torch
from torch.nn import functional as F


def _diffs(t):
    return t[:, :, :, 1:] - t[:, :, :, :-1], t[:, :, 1:, :] - t[:, :, :-1, :]


def total_variation(tensor):
    dw, dh = _diffs(tensor)
    return torch.mean(torch.abs(dw)) + torch.mean(torch.abs(dh))


def total_variation_for_POH(tensor):
    d1 = tensor[:, :, :, 2:] - tensor[:, :, :, :-2]
    d2 = tensor[:, :, 2:, :] - tensor[:, :, :-2, :]
    return torch.mean(torch.abs(d1)) + torch.mean(torch.abs(d2))


def total_variation_loss(y_hat, y):
    return torch.abs(total_variation(y_hat) - total_variation(y))


def amp_loss(amp_hat, amp, alpha=1.0):
    return F.mse_loss(amp_hat, amp) + alpha * total_variation_loss(amp_hat, amp)


def _sincos(p):
    return torch.cat((torch.sin(p), torch.cos(p)), dim=1)


def amp_phs_loss(amp_hat, phs_hat, amp, phs, alpha=1.0):
    a = torch.cat((amp_hat, torch.sin(phs_hat), torch.cos(phs_hat)), dim=1)
    b = torch.cat((amp, torch.sin(phs), torch.cos(phs)), dim=1)
    return F.mse_loss(a, b) + alpha * total_variation_loss(a, b)


def _focal_mean(d):
    with torch.no_grad():
        w = d / torch.max(d)
    return torch.mean(d * w)


def focal_freq_loss(fake_freq, real_freq):
    d = torch.abs(fake_freq - real_freq)
    with torch.no_grad():
        w = d / torch.max(d)
    return torch.mean((d * w) ** 2)


def focal_sincos_phase_gradient_loss(fake_phase, real_phase):
    """ref: loss_func.py:135-163 — weights d/max(d) are detached, max over the whole batch tensor."""
    (fw, fh), (rw, rh) = _diffs(_sincos(fake_phase)), _diffs(_sincos(real_phase))
    return _focal_mean(torch.abs(fw - rw)) + _focal_mean(torch.abs(fh - rh))


def phase_sincos_gradient_loss(fake_phase, real_phase):
    (fw, fh), (rw, rh) = _diffs(_sincos(fake_phase)), _diffs(_sincos(real_phase))
    return torch.mean(torch.abs(fw - rw)) + torch.mean(torch.abs(fh - rh))


def focal_sincos_phase_loss(fake_phase, real_phase):
    return _focal_mean(torch.abs(_sincos(fake_phase) - _sincos(real_phase)))


def plain_phase_loss(fake_phase, real_phase):
    return torch.mean(torch.abs(fake_phase - real_phase))


def perceptualLoss(feature_map_layers=(3, 8, 13, 22, 31), cuda=True, weights_path=None):
    """VGG19 perceptual loss (ref: loss_func.py:12-51) — implemented in perceptual.py on the HIP ops."""
    from .perceptual import perceptualLoss as _impl

    return _impl(feature_map_layers, cuda, weights_path)


class fakePerceptualLoss(torch.nn.Module):
    """Zero perceptual term (the VGG19 loss of loss_func.py:12-51 is SURVEY §8f N1)."""

    def __init__(self, feature_map_layers=(3, 8, 13, 22, 31), cuda=True):
        super().__init__()

    def forward(self, hat, target):
        return torch.zeros((), device=hat.device)
