"""Generator = RGBD2AP (UNet) -> AP2POH. ref: learnedMethodForHologram/watermelon_hologram/generator.py:15-59."""

from __future__ import annotations

import torch
from torch import nn

from .AP2POH import AP2POH
from .RGBD2AP import RGBD2AP


class Generator(nn.Module):
    def __init__(self, sample_row_num=192, sample_col_num=192, pad_size=160, filter_radius_coefficient=0.5, kernel_size=3,
                 pixel_pitch=3.74e-6, wave_length=torch.tensor([638e-9, 520e-9, 450e-9]), distance=torch.tensor([1e-3]),
                 pretrained_model_path=None, pretrained_model_path_RGBD2AP=None, pretrained_model_path_AP2POH=None):
        super().__init__()
        self.part1 = RGBD2AP(input_shape=(1, 4, sample_row_num, sample_col_num), pretrained_model_path=pretrained_model_path_RGBD2AP,
                             freeze=False, cuda=True, amplitude_scaler=1.1)
        self.part2 = AP2POH(input_shape=(1, 6, sample_row_num, sample_col_num), pretrained_model_path=pretrained_model_path_AP2POH,
                            freeze=False, cuda=True, filter_radius_coefficient=filter_radius_coefficient, pad_size=pad_size,
                            pixel_pitch=pixel_pitch, wave_length=wave_length, distance=distance, kernel_size=kernel_size)
        if pretrained_model_path is not None:
            # checkpoints are saved from whatever device the module lived on (generator.py:53-54)
            self.load_state_dict(torch.load(pretrained_model_path, map_location="cpu"))

    def forward(self, RGBD):
        amp_hat, phs_hat = self.part1(RGBD)
        return self.part2(amp_hat, phs_hat)
