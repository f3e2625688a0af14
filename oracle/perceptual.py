"""Oracle: VGG19 perceptual loss.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

ref: loss_func.py:12-51.  The reference class cannot be constructed offline (it downloads VGG19_Weights.DEFAULT), so
this restatement is checked against the product with seeded random weights only — parity of N1 is UNPINNED by
reference fixtures (the algorithm is restated from the cited lines: features[:32], ImageNet normalisation of
cat(hat, target), MSE after layers 3/8/13/22/31, mean over the five taps)."""

import torch
import torch.nn.functional as F

CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512)
MEAN, STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def vgg19_feature_state_dict(seed=19):
    g = torch.Generator().manual_seed(seed)
    sd, cin, idx = {}, 3, 0
    for v in CFG:
        if v == "M":
            idx += 1
            continue
        sd[f"{idx}.weight"] = torch.randn((v, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
        sd[f"{idx}.bias"] = torch.randn((v,), generator=g) * 0.05
        cin, idx = v, idx + 2
    return sd


def perceptual_loss(sd, hat, target, taps=(3, 8, 13, 22, 31)):
    x = torch.cat((hat, target), 0)
    x = (x - torch.tensor(MEAN).view(1, 3, 1, 1)) / torch.tensor(STD).view(1, 3, 1, 1)
    B, idx, loss = hat.size(0), 0, 0.0
    for v in CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            idx += 1
            continue
        x = F.relu(F.conv2d(x, sd[f"{idx}.weight"], sd[f"{idx}.bias"], padding=1))
        if idx + 1 in taps:
            loss = loss + F.mse_loss(x[:B], x[B:])
        idx += 2
        if idx > max(taps):
            break
    return loss / len(taps)
