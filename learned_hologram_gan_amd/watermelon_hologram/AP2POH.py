"""(amplitude, phase) at z -> phase-only hologram at the SLM plane.
ref: learnedMethodForHologram/watermelon_hologram/AP2POH.py:16-116.
(The stand-alone pre-training loop ``train_model`` of the reference is SURVEY §8f N4.)"""

from __future__ import annotations

import torch
from torch import nn

from ..poh_ops import PohEncodeFn
from ..angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_single_fixed_distance as fixed_distance_propogator
from ..neural_network_components import ChannelWiseSymmetricConv
from ..utilities import amplitude_normalizor, generate_checkerboard_mask, try_gpu
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
