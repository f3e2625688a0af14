"""Oracle: networks of the hot path as pure functions of a ``state_dict``
(SURVEY §8a rows A3, A4, A6, A7, A10).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pure torch, CPU, fp32.

The reference builds these as ``nn.Module`` trees with lazy layers
(ref: learnedMethodForHologram/neural_network_components.py:6-32, 35-95, 241-315;
watermelon_hologram/RGBD2AP.py:43-50; AP2POH.py:86-116; generator.py:56-59;
discriminator.py:16-51).  Here every network is a function ``f(sd, x, training)``
that reads tensors from a flat dict keyed by the reference's ``state_dict`` names,
so the same dict can be loaded into the reference modules (golden generation),
into this oracle, and into the HIP implementation under test.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

from . import optics

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# (block prefix, Cin, Cout) in execution order. ref: neural_network_components.py:247-300
UNET_BLOCKS = (
    ("encoder1.0.0", 4, 64),
    ("encoder2.1.0", 64, 128),
    ("encoder3.1.0", 128, 256),
    ("encoder4.1.0", 256, 512),
    ("bottleneck.1.0", 512, 1024),
    ("decoder1.0.0", 1024, 512),
    ("decoder2.0.0", 512, 256),
    ("decoder3.0.0", 256, 128),
    ("decoder4.0", 128, 64),
)
# (prefix, Cin, Cout) of the 2x2 stride-2 transposed convolutions
UNET_UPCONVS = (
    ("bottleneck.2", 1024, 512),
    ("decoder1.1", 512, 256),
    ("decoder2.1", 256, 128),
    ("decoder3.1", 128, 64),
)
UNET_HEAD = ("final_layer.0", 64, 6)

# (conv prefix, bn prefix or None, Cin, Cout, stride). ref: discriminator.py:16-41
CRITIC_LAYERS = (
    ("block1.0", None, 3, 32, 1),
    ("block2.0", "block2.1", 32, 64, 2),
    ("block3.0", "block3.1", 64, 128, 1),
    ("block4.0", "block4.1", 128, 256, 2),
    ("block5.0", "block5.1", 256, 512, 1),
    ("block6.0", "block6.1", 512, 1024, 2),
)
CRITIC_HEAD = ("conv", 1024, 1)


def is_buffer_key(key: str) -> bool:
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))


def as_parameters(sd: dict) -> dict:
    """Clone a state_dict and mark the learnable tensors as requiring grad."""
    out = {}
    for k, v in sd.items():
        t = v.detach().clone()
        if not is_buffer_key(k) and t.is_floating_point():
            t.requires_grad_(True)
        out[k] = t
    return out


def parameters_of(sd: dict, prefix: str = ""):
    return [v for k, v in sd.items() if k.startswith(prefix) and not is_buffer_key(k)]


# --------------------------------------------------------------------------- A3
def _batch_norm(sd, p, x, training):
    """PyTorch BatchNorm2d semantics: biased batch variance normalises in training mode,
    the unbiased one updates running_var (momentum 0.1, eps 1e-5);
    num_batches_tracked += 1 per training forward."""
    if training:
        sd[p + ".num_batches_tracked"] += 1
    return F.batch_norm(
        x,
        sd[p + ".running_mean"],
        sd[p + ".running_var"],
        sd[p + ".weight"],
        sd[p + ".bias"],
        training,
        BN_MOMENTUM,
        BN_EPS,
    )


def residual_block(sd, p, x, training):
    """relu(BN2(conv3x3(relu(BN1(conv3x3(x))))) + conv1x1(x)).

    ref: neural_network_components.py:26-32 — every block of the UNet is built with
    ``use_1x1conv=True`` (:297-300), so the 1x1 projection is always present.
    """
    y = F.conv2d(x, sd[p + ".convolution_layer_1.weight"], sd[p + ".convolution_layer_1.bias"], padding=1)
    y = F.relu(_batch_norm(sd, p + ".batch_norm_layer_1", y, training))
    y = F.conv2d(y, sd[p + ".convolution_layer_2.weight"], sd[p + ".convolution_layer_2.bias"], padding=1)
    y = _batch_norm(sd, p + ".batch_norm_layer_2", y, training)
    skip = F.conv2d(x, sd[p + ".convolution_layer_3.weight"], sd[p + ".convolution_layer_3.bias"])
    return F.relu(y + skip)


# --------------------------------------------------------------------------- A4
def unet(sd, prefix, x, training):
    """Four-level residual UNet 64..1024, sigmoid head with 6 channels.

    ref: neural_network_components.py:302-315 (forward), :241-300 (layers).
    The skip tensor comes FIRST in every channel concatenation (:310-313).
    """
    blk = lambda name, t: residual_block(sd, prefix + name, t, training)  # noqa: E731
    up = lambda name, t: F.conv_transpose2d(  # noqa: E731
        t, sd[prefix + name + ".weight"], sd[prefix + name + ".bias"], stride=2
    )
    pool = lambda t: F.max_pool2d(t, 2, 2)  # noqa: E731

    e1 = blk("encoder1.0.0", x)
    e2 = blk("encoder2.1.0", pool(e1))
    e3 = blk("encoder3.1.0", pool(e2))
    e4 = blk("encoder4.1.0", pool(e3))
    b = up("bottleneck.2", blk("bottleneck.1.0", pool(e4)))
    d1 = up("decoder1.1", blk("decoder1.0.0", torch.cat((e4, b), 1)))
    d2 = up("decoder2.1", blk("decoder2.0.0", torch.cat((e3, d1), 1)))
    d3 = up("decoder3.1", blk("decoder3.0.0", torch.cat((e2, d2), 1)))
    d4 = blk("decoder4.0", torch.cat((e1, d3), 1))
    head = F.conv2d(d4, sd[prefix + "final_layer.0.weight"], sd[prefix + "final_layer.0.bias"])
    return torch.sigmoid(head)


def rgbd_to_amp_phase(sd, rgbd, training, amplitude_scaler=1.1):
    """amp = 1.1*y[:, :3], phs = 2*pi*y[:, 3:]. ref: RGBD2AP.py:43-50."""
    y = unet(sd, "part1.part1.", rgbd, training)
    return amplitude_scaler * y[:, :3], 2 * torch.pi * y[:, 3:]


# --------------------------------------------------------------------------- A6
_TAP_CLASS = torch.tensor([[2, 1, 2], [1, 0, 1], [2, 1, 2]])


def symmetric_conv3x3(sd, p, x):
    """1-channel 3x3 conv whose 9 taps share 3 parameters by squared distance from
    the centre (index 0 centre, 1 edge, 2 corner) plus a scalar bias.
    ref: neural_network_components.py:42-75."""
    w = sd[p + ".params"][_TAP_CLASS].view(1, 1, 3, 3)
    return F.conv2d(x, w, sd[p + ".bias"], padding=1)


def channelwise_symmetric_conv(sd, prefix, x):
    """One symmetric conv per colour. ref: neural_network_components.py:85-95."""
    return torch.cat(
        [symmetric_conv3x3(sd, prefix + name, x[:, c : c + 1]) for c, name in enumerate(("conv_r", "conv_g", "conv_b"))],
        dim=1,
    )


# --------------------------------------------------------------------------- A7
def checkerboards(rows: int, cols: int):
    """m2 = (x+y)%2, m1 = 1-m2 with unit cells.
    ref: utilities.py:354-382 (generate_checkerboard_mask), AP2POH.py:37-49."""
    yy = torch.arange(rows).view(-1, 1)
    xx = torch.arange(cols).view(1, -1)
    m2 = ((xx + yy) % 2).to(torch.float32)
    return 1 - m2, m2


def normalize_amplitude(amp):
    """amp / (1.01 * max over (H,W)) per (b,c). ref: utilities.py:53-66."""
    peak = amp.amax(dim=-1, keepdim=True).amax(dim=-2, keepdim=True)
    return amp / (peak * 1.01)


def double_phase_encode(amp01, phs):
    """POH = m1*(phs+acos a) + m2*(phs-acos a). ref: AP2POH.py:86-96."""
    m1, m2 = checkerboards(amp01.shape[-2], amp01.shape[-1])
    ac = torch.acos(amp01)
    return m1 * (phs + ac) + m2 * (phs - ac)


def amp_phase_to_poh(sd, o: optics.Optics, H_fixed, amp_z, phs_z):
    """ref: AP2POH.py:105-116 (forward)."""
    field = optics.backpropagate_to_slm(o, H_fixed, amp_z, phs_z)
    mod = torch.complex(
        channelwise_symmetric_conv(sd, "part2.part1.", torch.real(field)),
        channelwise_symmetric_conv(sd, "part2.part1.", torch.imag(field)),
    )
    return double_phase_encode(normalize_amplitude(torch.abs(mod)), torch.angle(mod))


def generator(sd, o: optics.Optics, H_fixed, rgbd, training):
    """RGBD (B,4,H,W) -> POH (B,3,H,W). ref: generator.py:56-59."""
    amp, phs = rgbd_to_amp_phase(sd, rgbd, training)
    return amp_phase_to_poh(sd, o, H_fixed, amp, phs)


# --------------------------------------------------------------------------- A10
def critic(sd, x, training):
    """WGAN-GP critic: 48x48 patch scores flattened to (B, H*W/64).
    ref: discriminator.py:43-51 (forward), :16-41 (layers)."""
    for conv, bn, _cin, _cout, stride in CRITIC_LAYERS:
        x = F.conv2d(x, sd[conv + ".weight"], sd[conv + ".bias"], stride=stride, padding=1)
        if bn is not None:
            x = _batch_norm(sd, bn, x, training)
        x = F.leaky_relu(x, 0.2)
    x = F.conv2d(x, sd["conv.weight"], sd["conv.bias"], padding=1)
    return x.flatten(1)


# ------------------------------------------------------------------ work counts
def conv_macs_unet(h: int, w: int) -> int:
    """Multiply-accumulates of one UNet forward on an h x w frame (SURVEY §8d:
    114.586 GMAC at 384^2)."""
    total, level_hw = 0, [h * w, h * w // 4, h * w // 16, h * w // 64, h * w // 256]
    lvl = {"encoder1": 0, "encoder2": 1, "encoder3": 2, "encoder4": 3, "bottleneck": 4,
           "decoder1": 3, "decoder2": 2, "decoder3": 1, "decoder4": 0}
    for name, cin, cout in UNET_BLOCKS:
        px = level_hw[lvl[name.split(".")[0]]]
        total += px * (9 * cin * cout + 9 * cout * cout + cin * cout)
    for name, cin, cout in UNET_UPCONVS:
        px = level_hw[lvl[name.split(".")[0]]]  # input pixels; 4 outputs each
        total += px * 4 * cin * cout
    total += h * w * UNET_HEAD[1] * UNET_HEAD[2]
    return total


def conv_macs_critic(h: int, w: int) -> int:
    """28.007 GMAC per 384^2 sample."""
    total, hh, ww = 0, h, w
    for _c, _b, cin, cout, stride in CRITIC_LAYERS:
        hh, ww = (hh + stride - 1) // stride, (ww + stride - 1) // stride
        total += hh * ww * 9 * cin * cout
    total += hh * ww * 9 * CRITIC_HEAD[1] * CRITIC_HEAD[2]
    return total
