"""Oracle: reconstruction losses and image metrics.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pure torch, CPU, fp32.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F


def _forward_differences(t):
    """(d/dW, d/dH) first differences. ref: loss_func.py:71-72, 144-148."""
    return t[..., :, 1:] - t[..., :, :-1], t[..., 1:, :] - t[..., :-1, :]


def total_variation(t):
    """mean|dW| + mean|dH|. ref: loss_func.py:66-78."""
    dw, dh = _forward_differences(t)
    return dw.abs().mean() + dh.abs().mean()


def total_variation_loss(hat, target):
    """|TV(hat) - TV(target)|. ref: loss_func.py:94-98."""
    return torch.abs(total_variation(hat) - total_variation(target))


def focal_sincos_phase_gradient_loss(fake_phase, real_phase):
    """cat(sin, cos) -> first differences of hat and target -> d = |dhat - dtgt|,
    focal weight d / max(d) (detached, max over the WHOLE batch tensor), mean(d*w)
    summed over the two directions.  NaN when hat == target (max = 0).
    ref: loss_func.py:135-163."""
    sc_f = torch.cat((torch.sin(fake_phase), torch.cos(fake_phase)), dim=1)
    sc_r = torch.cat((torch.sin(real_phase), torch.cos(real_phase)), dim=1)
    total = 0.0
    for df, dr in zip(_forward_differences(sc_f), _forward_differences(sc_r)):
        d = torch.abs(df - dr)
        with torch.no_grad():
            weight = d / torch.max(d)
        total = total + torch.mean(d * weight)
    return total


def pixel_loss(hat, target):
    """ref: watermelon.py:433 (F.mse_loss)."""
    return F.mse_loss(hat, target)


def psnr(hat, target):
    """torchmetrics PeakSignalNoiseRatio() default: data_range = max(target) - min(target),
    10*log10(range^2 / mse) over the whole batch. ref: watermelon.py:134, 447-456
    (torchmetrics is not installed here: parity of this metric is unpinned)."""
    rng = target.max() - target.min()
    return 10.0 * torch.log10(rng**2 / F.mse_loss(hat, target))
