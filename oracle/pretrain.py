"""N4: the stand-alone pre-training loops of the two generator halves (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

ref: learnedMethodForHologram/watermelon_hologram/RGBD2AP.py:52-153 (UNet on amplitude + sin/cos phase),
     learnedMethodForHologram/watermelon_hologram/AP2POH.py:75-84, 118-230 (symmetric stencils through the propagator with the
     differentiable sigmoid mask and the spectrum loss of angular_spectrum_method.py:394-436).
Pinned by tests/golden/pretrain_small.pt, which the reference's own ``train_model`` methods produced (oracle/make_golden.py).
"""

from __future__ import annotations

import torch

from . import losses, nets, optics


class Plateau:
    """``ReduceLROnPlateau(opt, "min", factor, patience=4, threshold=1e-3, threshold_mode="rel", min_lr=1e-6)`` restated:
    an epoch is bad unless loss < best*(1-threshold); after more than `patience` bad epochs in a row lr <- max(lr*factor, min_lr)
    (only if that changes lr by more than 1e-8) and the count restarts.  ref: RGBD2AP.py:86-95, AP2POH.py:154-163."""

    def __init__(self, lr, factor, patience=4, threshold=1e-3, min_lr=1e-6, eps=1e-8):
        self.lr, self.factor, self.patience, self.threshold, self.min_lr, self.eps = lr, factor, patience, threshold, min_lr, eps
        self.best, self.bad = float("inf"), 0

    def step(self, loss):
        if loss < self.best * (1.0 - self.threshold):
            self.best, self.bad = loss, 0
        else:
            self.bad += 1
        if self.bad > self.patience:
            new = max(self.lr * self.factor, self.min_lr)
            if self.lr - new > self.eps:
                self.lr = new
            self.bad = 0
        return self.lr


def amp_loss(amp_hat, amp, alpha):
    """ref: loss_func.py:101-104."""
    return losses.pixel_loss(amp_hat, amp) + alpha * losses.total_variation_loss(amp_hat, amp)


def amp_phs_loss(amp_hat, phs_hat, amp, phs, alpha):
    """ref: loss_func.py:107-120."""
    a = torch.cat((amp_hat, torch.sin(phs_hat), torch.cos(phs_hat)), 1)
    b = torch.cat((amp, torch.sin(phs), torch.cos(phs)), 1)
    return losses.pixel_loss(a, b) + alpha * losses.total_variation_loss(a, b)


def _set_lr(opt, lr):
    for g in opt.param_groups:
        g["lr"] = lr


def train_rgbd2ap(sd, train_batches, val_batches, epochs=30, lr=1e-3, alpha=1e-3, gamma=0.1):
    """sd: generator state_dict as parameters (``nets.as_parameters``); only ``part1.*`` is trained.  The reference scales the
    target phase by 2*pi in ``train_model`` AND again in ``loss`` (RGBD2AP.py:104, 139-153): kept.  Loss sums are divided by the
    number of SAMPLES although each term is a batch mean (RGBD2AP.py:110-127): kept."""
    params = [v for k, v in sd.items() if k.startswith("part1.") and v.requires_grad]
    opt = torch.optim.Adam(params, lr=lr)
    sched = Plateau(lr, gamma)
    two_pi = 2 * torch.pi
    train_loss, test_loss = [], []
    for _ in range(epochs):
        tot, n = 0.0, 0
        for rgbd, amp, phs in train_batches:
            amp_hat, phs_hat = nets.rgbd_to_amp_phase(sd, rgbd, True)
            loss = amp_phs_loss(amp_hat, phs_hat, amp, two_pi * (two_pi * phs), alpha)
            opt.zero_grad()
            loss.backward()
            opt.step()
            tot, n = tot + loss.item(), n + rgbd.size(0)
        train_loss.append(tot / n)
        tot, n = 0.0, 0
        for rgbd, amp, phs in val_batches:
            with torch.no_grad():
                amp_hat, phs_hat = nets.rgbd_to_amp_phase(sd, rgbd, False)
                tot += amp_phs_loss(amp_hat, phs_hat, amp, two_pi * (two_pi * phs), alpha).item()
            n += rgbd.size(0)
        test_loss.append(tot / n)
        _set_lr(opt, sched.step(test_loss[-1]))
    return train_loss, test_loss


def sigmoid_mask(o: optics.Optics, coefficient):
    """sigmoid(radius - D), D = sqrt(u^2+v^2)*min(R,C) on the fftfreq grid, radius = min(R,C)*coefficient.
    ref: utilities.py:276-296, angular_spectrum_method.py:426-436."""
    short = min(o.rows, o.cols)
    u = torch.fft.fftfreq(o.rows).unsqueeze(-1)
    v = torch.fft.fftfreq(o.cols).unsqueeze(0)
    return torch.sigmoid(1.0 * (short * coefficient - torch.sqrt(u**2 + v**2) * short))


def filter_targets(o: optics.Optics, amp, phs, coefficient):
    """``dataloader_filter``: crop(ifft2(fft2(pad(a e^{i phs})) * sigmoid_mask)) -> (abs, angle). ref: AP2POH.py:75-84."""
    g = optics.crop_field(o, torch.fft.ifft2(torch.fft.fft2(optics.pad_field(o, optics.polar(amp, phs))) * sigmoid_mask(o, coefficient)))
    return torch.abs(g), torch.angle(g)


def poh_to_amp_phase_with_spectrum_loss(o: optics.Optics, H_fixed, poh, coefficient):
    """ref: angular_spectrum_method.py:394-412."""
    G0 = torch.fft.fft2(optics.pad_field(o, torch.exp(1j * poh)))
    Gz = G0 * H_fixed * sigmoid_mask(o, coefficient)
    g = optics.crop_field(o, torch.fft.ifft2(Gz))
    return torch.abs(g), torch.angle(g), torch.mean(torch.abs(G0) - torch.abs(Gz))


def train_ap2poh(sd, o: optics.Optics, H_fixed, train_batches, val_batches, coefficient=0.45, epochs=30, lr=1e-3, alpha=1e-3,
                 beta=1e-5, gamma=0.1):
    """Only ``part2.*`` (three stencil taps + bias per colour) is trained.  ref: AP2POH.py:118-230."""
    params = [v for k, v in sd.items() if k.startswith("part2.") and v.requires_grad]
    opt = torch.optim.Adam(params, lr=lr)
    sched = Plateau(lr, gamma)

    def batch_loss(amp, phs):
        amp, phs = filter_targets(o, amp, phs, coefficient)
        poh = nets.amp_phase_to_poh(sd, o, H_fixed, amp, phs)
        amp_hat, _, spectrum = poh_to_amp_phase_with_spectrum_loss(o, H_fixed, poh, coefficient)
        return amp_loss(amp_hat, amp, alpha) + beta * spectrum, poh.size(0)

    train_loss, test_loss = [], []
    for _ in range(epochs):
        tot, n = 0.0, 0
        for amp, phs in train_batches:
            loss, b = batch_loss(amp, phs)
            opt.zero_grad()
            loss.backward()
            opt.step()
            tot, n = tot + loss.item(), n + b
        train_loss.append(tot / n)
        tot, n = 0.0, 0
        for amp, phs in val_batches:
            with torch.no_grad():
                loss, b = batch_loss(amp, phs)
            tot, n = tot + loss.item(), n + b
        test_loss.append(tot / n)
        _set_lr(opt, sched.step(test_loss[-1]))
    return train_loss, test_loss
