"""(amplitude, phase) at z -> phase-only hologram at the SLM plane.
ref: learnedMethodForHologram/watermelon_hologram/AP2POH.py:16-230 (incl. the stand-alone pre-training loop)."""

from __future__ import annotations

import torch
from torch import nn

from ..poh_ops import PohEncodeFn
from ..angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_single_fixed_distance as fixed_distance_propogator
from ..neural_network_components import ChannelWiseSymmetricConv
from ..asm_ops import F_MUL, Factor, IN_POLAR, OUT_ABS_ANGLE
from ..optim import run_pretraining
from ..utilities import amplitude_normalizor, generate_checkerboard_mask, try_gpu
from .loss_func import amp_loss
from .RGBD2AP import initialize_like_reference


class AP2POH(nn.Module):
    def __init__(self, input_shape=(1, 6, 192, 192), pretrained_model_path=None, freeze=False, cuda=True, pad_size=192,
                 filter_radius_coefficient=0.5, pixel_pitch=3.74e-6, wave_length=torch.tensor([638e-9, 520e-9, 450e-9]),
                 distance=torch.tensor([1e-3]), kernel_size=3):
        super().__init__()
        self.input_shape, self.pretrained_model_path, self.freeze = input_shape, pretrained_model_path, freeze
        self.device = try_gpu() if cuda else torch.device("cpu")
        rows, cols = input_shape[-2], input_shape[-1]
        self.checkerboard_mask_1 = generate_checkerboard_mask(rows, cols, 1, True).to(self.device)
        self.checkerboard_mask_2 = generate_checkerboard_mask(rows, cols, 1, False).to(self.device)
        self.propagator = fixed_distance_propogator(sample_row_num=rows, sample_col_num=cols, pad_size=pad_size,
                                                    filter_radius_coefficient=filter_radius_coefficient, pixel_pitch=pixel_pitch,
                                                    wave_length=wave_length, band_limit=False, cuda=cuda, distance=distance)
        self.part1 = ChannelWiseSymmetricConv(kernel_size=kernel_size, padding=(kernel_size - 1) // 2).to(self.device)
        initialize_like_reference(self)
        if pretrained_model_path is not None:
            self.load_state_dict(torch.load(pretrained_model_path, map_location="cpu"))
            if freeze:
                self.eval()
                self.requires_grad_(False)

    def dataloader_filter(self, amp, phs, filter_radius_coefficient):
        """Low-pass the targets with the differentiable sigmoid mask: (abs, angle) of crop(ifft2(fft2(pad(a e^{i phs})) * mask)).
        ref: AP2POH.py:75-84.  One fused angular-spectrum launch with the (real) mask as its only spectral factor."""
        pr = self.propagator
        mask = pr.generate_circular_frequency_mask_differentiable(filter_radius_coefficient)
        slab = torch.complex(mask, torch.zeros_like(mask)).unsqueeze(0).contiguous()
        index = torch.zeros(amp.shape[0] * 3, dtype=torch.int32, device=amp.device)
        a, p, _ = pr._run(amp, phs, IN_POLAR, OUT_ABS_ANGLE, [Factor(slab, F_MUL, index)])
        return a, p

    def train_model(self, train_loader, val_loader, filter_radius_coefficient=0.45, epochs=30, lr=1e-3, alpha=1e-3, beta=1e-5,
                    hyperparameter_gamma=0.1, save_path=None, checkpoint_iterval=10):
        """Pre-train the symmetric stencils on (amplitude, phase) batches through the propagator.  ref: AP2POH.py:118-218."""

        def batch_loss(batch):
            amp, phs = self.dataloader_filter(batch[0], batch[1], filter_radius_coefficient)
            poh = self(amp, phs)
            amp_hat, _, spectrum_loss = self.propagator.propagate_POH2AP_forward_with_spectrum_loss(poh, filter_radius_coefficient)
            return self.loss(amp_hat, amp, alpha) + beta * spectrum_loss, poh.size(0)

        run_pretraining(self, batch_loss, train_loader, val_loader, epochs, lr, hyperparameter_gamma, save_path, checkpoint_iterval)

    def loss(self, amp_hat, amp, alpha):
        """ref: AP2POH.py:220-230."""
        return amp_loss(amp_hat, amp, alpha)

    def double_phase_method(self, amp, phs):
        """POH = m1*(phs + acos a) + m2*(phs - acos a). ref: AP2POH.py:86-96."""
        ac = torch.acos(amp)
        return self.checkerboard_mask_1 * (phs + ac) + self.checkerboard_mask_2 * (phs - ac)

    def phs_sincos(self, phs):
        return torch.cat((torch.sin(phs), torch.cos(phs)), dim=-3)

    def _encode_fused(self, field):
        """Stencil + per-plane max, then normalise / angle / acos / checkerboard encode: two HIP kernels forward,
        three backward (poh_ops.PohEncodeFn)."""
        taps, bias = self.part1.taps_and_bias()  # torch.stack / cat of the six parameters: differentiable
        return PohEncodeFn.apply(field, taps, bias)

    def forward(self, amp_z, phs_z):
        field = self.propagator.propagate_AP2C_backward(amp_z, phs_z)
        if isinstance(self.part1, ChannelWiseSymmetricConv) and field.shape[1] == 3 and field.is_cuda:
            return self._encode_fused(field)
        mod = torch.complex(self.part1(torch.real(field)), self.part1(torch.imag(field)))
        return self.double_phase_method(amplitude_normalizor(torch.abs(mod)), torch.angle(mod))
