"""VGG19 perceptual loss on the HIP conv / pool ops.
ref: learnedMethodForHologram/watermelon_hologram/loss_func.py:12-51 — vgg19.features[:32], ImageNet normalisation,
MSE between the hat and target feature maps after layers 3, 8, 13, 22, 31, averaged over the five taps.

The reference downloads ``VGG19_Weights.DEFAULT``; there is no network here, so weights come from a local file
(``weights_path`` or $LHG_VGG19_WEIGHTS: a torchvision ``vgg19`` or ``vgg19().features`` state_dict).  Without one the
layers keep a seeded random initialisation — fine for throughput work, meaningless as a perceptual metric (a warning
says so).  ``self.net`` has torchvision's ``features`` indices, so ``net.<i>.weight`` keys line up with the reference.
"""

from __future__ import annotations

import os
import warnings

import torch
from torch import nn

from .. import hip_ops as ops
from ..hip_ops import ACT_RELU
from ..utilities import try_gpu

# torchvision vgg19 "E" configuration up to features[31]
_VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512)
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def build_vgg19_features(upto: int = 31) -> nn.Sequential:
    layers, cin = [], 3
    for v in _VGG19_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers[: upto + 1])


class perceptualLoss(nn.Module):
    def __init__(self, feature_map_layers=(3, 8, 13, 22, 31), cuda=True, weights_path=None):
        super().__init__()
        self.device = try_gpu() if cuda else torch.device("cpu")
        self.feature_map_layers = list(feature_map_layers)
        self.feature_map_layers_num = len(self.feature_map_layers)
        gen_state = torch.random.get_rng_state()
        torch.manual_seed(19)
        self.net = build_vgg19_features(max(self.feature_map_layers))
        torch.random.set_rng_state(gen_state)
        weights_path = weights_path or os.environ.get("LHG_VGG19_WEIGHTS")
        self.pretrained = False
        if weights_path:
            sd = torch.load(weights_path, map_location="cpu")
            sd = {k[len("features."):] if k.startswith("features.") else k: v for k, v in sd.items()}
            own = self.net.state_dict()
            self.net.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=True)
            self.pretrained = True
        else:
            warnings.warn("perceptualLoss: no VGG19 weights file given (weights_path / $LHG_VGG19_WEIGHTS); using a seeded random "
                          "initialisation — throughput only, not a perceptual metric", stacklevel=2)
        self.net.to(self.device)
        for p in self.net.parameters():
            p.requires_grad = False
        self.register_buffer("_mean", torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1).to(self.device), persistent=False)
        self.register_buffer("_std", torch.tensor(IMAGENET_STD).view(1, 3, 1, 1).to(self.device), persistent=False)

    def features(self, x):
        """NCHW (B,3,H,W) -> list of NHWC feature maps after the tapped ReLU layers."""
        h = ops.ToNHWC.apply((x - self._mean) / self._std, 32)
        taps = []
        for name, layer in self.net._modules.items():
            idx = int(name)
            if isinstance(layer, nn.Conv2d):
                # conv + the following ReLU, fused (the 3 -> 64 first layer takes the thin-convolution kernels inside the op)
                h = ops.ConvBiasActFn.apply(h, layer.weight, layer.bias, 1, ACT_RELU, 0.0)
            elif isinstance(layer, nn.MaxPool2d):
                h = ops.MaxPool2x2Fn.apply(h)
            # nn.ReLU entries are fused into the conv epilogue; a tap index names the ReLU's output
            if idx in self.feature_map_layers and isinstance(layer, nn.ReLU):
                taps.append(h)
        return taps

    def forward(self, hat, target):
        with torch.no_grad():
            ft = self.features(target)  # no gradient flows to the target: skip its backward entirely
        fh = self.features(hat)
        loss = torch.zeros((), device=hat.device)
        for a, b in zip(fh, ft):
            loss = loss + torch.nn.functional.mse_loss(a.float(), b.float())  # (feature maps are bf16 in the bf16 storage mode)
        return loss / self.feature_map_layers_num
