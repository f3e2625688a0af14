"""RGBD -> (amplitude, phase) stage of the generator.
ref: learnedMethodForHologram/watermelon_hologram/RGBD2AP.py:15-176 (incl. the stand-alone pre-training loop)."""

from __future__ import annotations

import torch
from torch import nn

from ..neural_network_components import UNet
from ..optim import run_pretraining
from ..utilities import try_gpu
from .loss_func import amp_phs_loss


def initialize_like_reference(module: nn.Module) -> None:
    """Xavier-normal convs, Kaiming-normal(fan_out) transposed convs, BN (1, 0), zero biases.
    ref: RGBD2AP.py:160-176 / AP2POH.py:238-253 (same routine in both classes)."""
    for m in module.modules():
        if isinstance(m, nn.ConvTranspose2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.xavier_normal_(m.weight)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
        else:
            continue
        if getattr(m, "bias", None) is not None:
            nn.init.constant_(m.bias, 0)


class RGBD2AP(nn.Module):
    def __init__(self, input_shape=(1, 4, 192, 192), pretrained_model_path=None, freeze=False, cuda=True, amplitude_scaler=1.1):
        super().__init__()
        self.input_shape, self.pretrained_model_path, self.freeze = input_shape, pretrained_model_path, freeze
        self.device = try_gpu() if cuda else torch.device("cpu")
        self.amplitude_scaler = amplitude_scaler
        self.part1 = UNet(output_channels=6, in_channels=input_shape[1]).to(self.device)
        initialize_like_reference(self)
        if pretrained_model_path is not None:
            self.load_state_dict(torch.load(pretrained_model_path, map_location="cpu"))
            if freeze:
                self.eval()
                self.requires_grad_(False)

    def forward(self, RGBD):
        y = self.part1(RGBD)
        return self.amplitude_scaler * y[:, :3, :, :], 2 * torch.pi * y[:, 3:, :, :]

    def train_model(self, train_loader, val_loader, epochs=30, lr=1e-3, alpha=1e-3, hyperparameter_gamma=0.1, save_path=None,
                    checkpoint_iterval=10):
        """Pre-train the UNet on (RGBD, amplitude, phase in [0,1)) batches.  ref: RGBD2AP.py:52-137."""

        def batch_loss(batch):
            img_depth, amp, phs = batch
            amp_hat, phs_hat = self(img_depth)
            return self.loss(amp_hat, phs_hat, amp, 2 * torch.pi * phs, alpha), img_depth.size(0)

        run_pretraining(self, batch_loss, train_loader, val_loader, epochs, lr, hyperparameter_gamma, save_path, checkpoint_iterval)

    def loss(self, amp_hat, phs_hat, amp, phs, alpha):
        """ref: RGBD2AP.py:139-153 — the target phase is scaled by 2*pi here as well as by the caller in ``train_model``."""
        return amp_phs_loss(amp_hat, phs_hat, amp, 2 * torch.pi * phs, alpha)
