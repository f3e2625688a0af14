"""Oracle: deterministic per-key weights with the reference's state_dict schema.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's checkpoints are 130 MB (generator) + 25 MB (critic) and the pretrained
ones are a Google-Drive download, so parity runs fill every key from a seeded CPU
generator instead (SURVEY §8d "Synthetic inputs").  Key names / shapes / dtypes follow
SURVEY Appendix A; ``oracle/make_golden.py`` loads these dicts into the real reference
modules with ``strict=True``, which is what pins the schema.

Statistics follow the reference initialisers (ref: RGBD2AP.py:155-176 — Xavier-normal
convs, Kaiming-normal(fan_out) transposed convs, BN weight 1 / bias 0; sym-conv params
``abs(randn)``, neural_network_components.py:44-45) but biases, BN affine parameters and
running statistics are perturbed away from their trivial values so that every term of
every kernel is exercised.
"""

from __future__ import annotations

import math
from collections import OrderedDict

import torch

from .nets import CRITIC_HEAD, CRITIC_LAYERS, UNET_BLOCKS, UNET_HEAD, UNET_UPCONVS


def _normal(g, shape, std):
    return torch.randn(shape, generator=g, dtype=torch.float32) * std


def _uniform(g, shape, lo, hi):
    return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo


def _conv(sd, g, name, cout, cin, k):
    fan_in, fan_out = cin * k * k, cout * k * k
    sd[name + ".weight"] = _normal(g, (cout, cin, k, k), math.sqrt(2.0 / (fan_in + fan_out)))
    sd[name + ".bias"] = _uniform(g, (cout,), -0.05, 0.05)


def _bn(sd, g, name, c):
    sd[name + ".weight"] = 1.0 + _normal(g, (c,), 0.1)
    sd[name + ".bias"] = _normal(g, (c,), 0.1)
    sd[name + ".running_mean"] = _normal(g, (c,), 0.1)
    sd[name + ".running_var"] = _uniform(g, (c,), 0.5, 1.5)
    sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)


def generator_state_dict(seed: int = 122731) -> "OrderedDict[str, torch.Tensor]":
    """160 keys, 32,440,274 parameters."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    pre = "part1.part1."
    ups = {u[0].split(".")[0]: u for u in UNET_UPCONVS}
    for blk, cin, cout in UNET_BLOCKS:
        p = pre + blk
        _conv(sd, g, p + ".convolution_layer_1", cout, cin, 3)
        _conv(sd, g, p + ".convolution_layer_2", cout, cout, 3)
        _conv(sd, g, p + ".convolution_layer_3", cout, cin, 1)
        _bn(sd, g, p + ".batch_norm_layer_1", cout)
        _bn(sd, g, p + ".batch_norm_layer_2", cout)
        stage = blk.split(".")[0]
        if stage in ups:  # the transposed conv follows its block inside the same nn.Sequential
            name, ci, co = ups[stage]
            # ConvTranspose2d weight layout is (Cin, Cout, kH, kW); kaiming fan_out = Cin*k*k
            sd[pre + name + ".weight"] = _normal(g, (ci, co, 2, 2), math.sqrt(2.0 / (ci * 4)))
            sd[pre + name + ".bias"] = _uniform(g, (co,), -0.05, 0.05)
    name, cin, cout = UNET_HEAD
    _conv(sd, g, pre + name, cout, cin, 1)
    for colour in ("conv_r", "conv_g", "conv_b"):
        sd[f"part2.part1.{colour}.params"] = _normal(g, (3,), 1.0).abs()
        sd[f"part2.part1.{colour}.bias"] = _normal(g, (1,), 0.05)
    return sd


def critic_state_dict(seed: int = 122732) -> "OrderedDict[str, torch.Tensor]":
    """39 keys, 6,301,377 parameters."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for conv, bn, cin, cout, _s in CRITIC_LAYERS:
        _conv(sd, g, conv, cout, cin, 3)
        if bn is not None:
            _bn(sd, g, bn, cout)
    name, cin, cout = CRITIC_HEAD
    _conv(sd, g, name, cout, cin, 3)
    return sd


def synthetic_batch(batch, rows, cols, seed=122731):
    """RGBD, target amplitude and target phase in [0,1) (the reference's own
    seed is trainingModel.py:18)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    rgbd = torch.rand((batch, 4, rows, cols), generator=g)
    amp = torch.rand((batch, 3, rows, cols), generator=g)
    phs = torch.rand((batch, 3, rows, cols), generator=g)
    return rgbd, amp, phs


def smooth_batch(batch, rows, cols, seed=7):
    """Low-frequency synthetic frames (sums of a few sinusoids) — closer to natural
    RGBD than white noise; used where conditioning matters (angle() near |z|=0)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    yy = torch.linspace(0, 1, rows).view(1, 1, -1, 1)
    xx = torch.linspace(0, 1, cols).view(1, 1, 1, -1)

    def field(ch):
        acc = torch.zeros(batch, ch, rows, cols)
        for _ in range(4):
            fy = torch.rand((batch, ch, 1, 1), generator=g) * 6
            fx = torch.rand((batch, ch, 1, 1), generator=g) * 6
            ph = torch.rand((batch, ch, 1, 1), generator=g) * 6.28
            acc += torch.sin(fy * yy * 6.28 + fx * xx * 6.28 + ph)
        return (acc / 8 + 0.5).clamp(0.02, 0.98)

    return field(4), field(3), field(3)
