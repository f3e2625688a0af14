"""Split-K gather-GEMM (GGParams::ks, gg_splitk_finish_kernel): the launches that cannot fill the chip — the UNet's 24^2 x 1024-channel
bottleneck (ref: neural_network_components.py:246-250, 2304 pixels at batch 4) — cut their K axis into ranges of whole 32-channel chunks.
Run on the MI355X box:  python -m pytest tests -m gpu -q
"""

import hashlib
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# N, H, W, Ci, Co, k, stride: every one has <= 160 tiles of 128 x 128 and >= 64 K steps (the rule of launch_gg_split) — ragged extents,
# a Co that is no multiple of 64, 3 and 4 ranges, a stride-2 layer (no strip kernels) and a 1x1 layer with a long K axis
CASES = [
    (4, 24, 24, 1024, 1024, 3, 1),
    (2, 24, 24, 512, 1024, 3, 1),
    (1, 17, 23, 256, 96, 3, 1),
    (2, 20, 12, 320, 512, 3, 2),
    (3, 8, 8, 2080, 64, 1, 1),
]


def _tensors(case):
    N, H, W, Ci, Co, k, stride = case
    g = torch.Generator().manual_seed(Ci + Co + H)
    x = torch.randn((N, Ci, H, W), generator=g) * torch.logspace(-1, 1, Ci).view(1, Ci, 1, 1)
    w = torch.randn((Co, Ci, k, k), generator=g) * (Ci * k * k) ** -0.5
    b = torch.randn((Co,), generator=g)
    return x, w, b


def _run_case(ops, case, with_stats=False):
    """(y, gx) of the HIP conv and its input gradient on the GPU (NHWC), plus the statistics tag when asked."""
    N, H, W, Ci, Co, k, stride = case
    x, w, b = _tensors(case)
    xg = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wg, bg = w.to(DEV), b.to(DEV)
    y = ops.conv2d_forward_raw(xg, wg, bg, stride, bn_stats=with_stats)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
    gx = ops.Conv2dInputGradFn.apply(gy, wg, stride, H, W, Ci)
    return y, gx, gy


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_splitk_conv_matches_float64(case):
    """Forward and input gradient of the split launches against a float64 convolution, at the bar of the unsplit kernels (3e-6 of the
    largest element).  (Tiling variants and the unsplit launch: test_splitk_is_a_function_of_the_geometry_not_of_the_tiling.)"""
    from learned_hologram_gan_amd import hip_ops as ops

    N, H, W, Ci, Co, k, stride = case
    x, w, b = _tensors(case)
    y, gx, gy = _run_case(ops, case)
    y64 = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=k // 2)
    e = (y.permute(0, 3, 1, 2).cpu().double() - y64).abs().max().item() / y64.abs().max().item()
    assert e < 3e-6, ("forward", case, e)
    x64 = x.double().requires_grad_(True)
    (F.conv2d(x64, w.double(), None, stride=stride, padding=k // 2) * gy.permute(0, 3, 1, 2).cpu().double()).sum().backward()
    e = (gx[..., :Ci].permute(0, 3, 1, 2).cpu().double() - x64.grad).abs().max().item() / x64.grad.abs().max().item()
    assert e < 3e-6, ("input gradient", case, e)


def test_splitk_leaves_the_batch_norm_statistics_rows():
    """A split launch asked for the BatchNorm statistics (lhg_conv2d_forward_stats): the finish kernel writes the partial rows; finished,
    they agree with float64 statistics of the stored tensor."""
    from learned_hologram_gan_amd import hip_ops as ops
    from learned_hologram_gan_amd.native import call, ptr, stream_ptr

    for case in CASES[:3]:
        N, H, W, Ci, Co, k, stride = case
        y, _, _ = _run_case(ops, case, with_stats=True)
        tag = y.__dict__.get("_lhg_bn_partial")
        assert tag is not None and tag[2] > 0, case
        pixels = y.shape[0] * y.shape[1] * y.shape[2]
        stats = torch.empty(2 * Co, device=DEV)
        call("lhg_bn_stats_finish", ptr(tag[1]), tag[2], ptr(tag[3]), pixels, Co, ptr(stats), None, None, 0.1, 1e-5, stream_ptr())
        yd = y.double().reshape(pixels, Co)
        assert (stats[:Co].double() - yd.mean(0)).abs().max().item() <= 2e-6 * yd.abs().max().item(), case
        assert (stats[Co:].double() * torch.sqrt(yd.var(0, unbiased=False) + 1e-5) - 1).abs().max().item() <= 5e-6, case


_CHILD = r"""
import hashlib, sys, torch
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import test_gpu_splitk as T
from learned_hologram_gan_amd import hip_ops as ops
h = hashlib.sha256()
for case in T.CASES:
    y, gx, _ = T._run_case(ops, case)
    h.update(y.cpu().numpy().tobytes()); h.update(gx.cpu().numpy().tobytes())
print("HASH", h.hexdigest())
"""


def test_splitk_is_a_function_of_the_geometry_not_of_the_tiling():
    """The K ranges follow from the geometry alone and are added in ascending order by one kernel: every tiling variant of a split launch —
    gg3s tiles of 128 x 128, 128 x 64, 64 x 64, eight consumer waves, 64 x 128, the strip kernels (padded pixel rows) incl. the twelve-wave
    form — gives the same bits (one child process per forced variant: LHG_GGS_VARIANT is read once).  With LHG_SPLITK=0 the results
    differ in rounding only (3e-6)."""
    seen = {}
    for gv in (0, 1, 2, 3, 4, 5, 7, 8, 9, 10):
        env = dict(os.environ, LHG_AUTOTUNE="0", LHG_GGS_VARIANT=str(gv), PYTHONPATH=ROOT)
        out = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, os.path.join(ROOT, "tests"))], cwd=ROOT, env=env, capture_output=True, text=True)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("HASH")]
        assert out.returncode == 0 and lines, out.stdout[-800:] + out.stderr[-1500:]
        seen[gv] = lines[-1]
    assert len(set(seen.values())) == 1, seen
    env = dict(os.environ, LHG_AUTOTUNE="0", LHG_SPLITK="0", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", _CHILD % (ROOT, os.path.join(ROOT, "tests"))], cwd=ROOT, env=env, capture_output=True, text=True)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("HASH")]
    assert out.returncode == 0 and lines and lines[-1] not in set(seen.values()), "LHG_SPLITK=0 gave the split launch's bits: the switch does nothing?"
    _ = hashlib
