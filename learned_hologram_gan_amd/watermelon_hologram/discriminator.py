"""WGAN-GP critic. ref: learnedMethodForHologram/watermelon_hologram/discriminator.py:5-67."""

from __future__ import annotations

import torch
from torch import nn

from .. import hip_ops as ops
from ..hip_ops import ACT_LEAKY
from ..neural_network_components import _count_batch, flush_batch_counters


class WGANGPDiscriminator192(nn.Module):
    """conv3x3(3->32)+LReLU; [conv3x3 + BN + LReLU(0.2)] x5 with (C, stride) = (64,2) (128,1) (256,2) (512,1) (1024,2);
    conv3x3(1024->1); flatten -> (B, H*W/64)."""

    def __init__(self, pretrained_model_path=None, feature_d=32, cuda=True):
        super().__init__()
        self.device = torch.device("cuda") if cuda and torch.cuda.is_available() else torch.device("cpu")
        f = feature_d
        self.block1 = nn.Sequential(nn.Conv2d(3, f, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(0.2, inplace=True))
        self.block2 = self._make_layer(f, f * 2, stride=2)
        self.block3 = self._make_layer(f * 2, f * 4, stride=1)
        self.block4 = self._make_layer(f * 4, f * 8, stride=2)
        self.block5 = self._make_layer(f * 8, f * 16, stride=1)
        self.block6 = self._make_layer(f * 16, f * 32, stride=2)
        self.conv = nn.Conv2d(f * 32, 1, kernel_size=3, stride=1, padding=1)
        self.flatten = nn.Flatten()
        self.to(self.device)
        if pretrained_model_path is not None:
            self.load_state_dict(torch.load(pretrained_model_path, map_location="cpu"))

    def _make_layer(self, in_channels, out_channels, stride):
        return nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1),
                             nn.BatchNorm2d(out_channels), nn.LeakyReLU(0.2, inplace=True))

    def forward(self, x):
        h = ops.ToNHWC.apply(x, 32)
        c1 = self.block1[0]
        if self.training:
            # thin first layer (3 -> 32) and thin head (1024 -> 1): direct kernels (csrc/thin_conv.hip), LeakyReLU fused
            h = ops.ConvBiasActFn.apply(h, c1.weight, c1.bias, 1, ACT_LEAKY, 0.2)
            for blk in (self.block2, self.block3, self.block4, self.block5, self.block6):
                conv, bn = blk[0], blk[1]
                y = ops.Conv2dFn.apply(h, conv.weight, conv.bias, conv.stride[0], "feeds_bn")
                h = ops.BatchNormTrainFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, None, ACT_LEAKY, 0.2, None)
                _count_batch(bn)
            flush_batch_counters()
            s = ops.Conv2dFn.apply(h, self.conv.weight, self.conv.bias, 1, None)
        else:
            with torch.no_grad():
                h = ops.conv2d_forward_raw(h, c1.weight, c1.bias, 1, act=ACT_LEAKY, slope=0.2, measure_out=True)
                for blk in (self.block2, self.block3, self.block4, self.block5, self.block6):
                    conv, bn = blk[0], blk[1]
                    sc, sh = ops.batch_norm_eval_affine(bn.weight, bn.bias, bn.running_mean, bn.running_var)
                    h = ops.conv2d_forward_raw(h, conv.weight, conv.bias, conv.stride[0], act=ACT_LEAKY, slope=0.2, scale=sc, shift=sh,
                                               measure_out=True)
                s = ops.conv2d_forward_raw(h, self.conv.weight, self.conv.bias, 1)
        return s.reshape(s.shape[0], -1).float()  # (N, H/8, W/8, 1) -> (N, H*W/64), same order as Flatten on NCHW; scores are fp32


    def forward_pair(self, a, b):
        """(D(a), D(b)) — the two critic passes of a WGAN-GP update (ref: watermelon.py:243-244: ``self.discriminator(target_amps)``,
        ``self.discriminator(fake)``) as ONE pass over the two batches stacked on dim 0: every convolution is one GEMM over 2B samples
        (twice the pixels: better chip fill on the deep layers, one weight-gradient GEMM instead of two), every BatchNorm normalises the
        halves separately and in order (BatchNormPairTrainFn: a's statistics and running-statistics update, then b's) — the values two
        calls compute.  Falls back to two calls outside training, with synchronised statistics or for unequal batches."""
        if not self.training or ops.sync_world() > 1 or a.shape != b.shape or not a.is_cuda:
            return self.forward(a), self.forward(b)
        B = a.shape[0]
        h = ops.ToNHWC.apply(torch.cat((a, b), 0), 32)
        c1 = self.block1[0]
        h = ops.ConvBiasActFn.apply(h, c1.weight, c1.bias, 1, ACT_LEAKY, 0.2)
        for blk in (self.block2, self.block3, self.block4, self.block5, self.block6):
            conv, bn = blk[0], blk[1]
            y = ops.Conv2dFn.apply(h, conv.weight, conv.bias, conv.stride[0], "feeds_bn_pair")
            h = ops.BatchNormPairTrainFn.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, ACT_LEAKY, 0.2)
            _count_batch(bn, bn)
        flush_batch_counters()
        s = ops.Conv2dFn.apply(h, self.conv.weight, self.conv.bias, 1, None)
        s = s.reshape(s.shape[0], -1).float()
        return s[:B], s[B:]


class fakeDiscriminator(nn.Module):
    """Returns 0 (training without the critic). ref: discriminator.py:54-67."""

    def __init__(self, pretrained_model_path=None, feature_d=32, cuda=True):
        super().__init__()
        self.a = nn.parameter.Parameter(torch.tensor([1.0]))
        self.device = torch.device("cuda") if cuda and torch.cuda.is_available() else torch.device("cpu")
        self._requires_grad = False

    def forward(self, _):
        return torch.zeros(1, device=self.device)
