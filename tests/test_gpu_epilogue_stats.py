"""BatchNorm batch statistics from the conv GEMM's epilogue (ABI 10: lhg_conv2d_forward_stats + lhg_bn_stats_finish) against the
statistics pass over the tensor (lhg_bn_stats) — ref: conv -> bn of neural_network_components.py:27-30, discriminator.py:34-39.
Run on the MI355X box:  python -m pytest tests -m gpu -q
"""

import ctypes
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# N, H, W, Ci, Co, k, stride — ragged extents (rows past the end, padding columns of the strip kernels), a Co that is not a multiple of
# the 64 / 128-wide N tile, a stride-2 and a 1x1 layer, a layer with more than 2048 partial rows (the wide finish kernel)
CASES = [
    (2, 48, 48, 64, 64, 3, 1),
    (4, 24, 40, 128, 256, 3, 1),
    (2, 50, 34, 32, 64, 3, 2),
    (1, 17, 23, 64, 96, 3, 1),
    (2, 32, 32, 64, 128, 1, 1),
    (3, 8, 8, 256, 512, 3, 1),
    (2, 192, 192, 64, 64, 3, 1),
]


def _check(ops, mode, case):
    from learned_hologram_gan_amd.native import call, ptr, stream_ptr

    N, H, W, Ci, Co, k, stride = case
    g = torch.Generator().manual_seed(N * H + Co)
    x = (torch.randn((N, H, W, Ci), generator=g) * torch.logspace(-1, 1, Ci)).to(DEV)
    w = (torch.randn((Co, Ci, k, k), generator=g) * (Ci * k * k) ** -0.5).to(DEV)
    bias = (torch.randn((Co,), generator=g) * 3.0).to(DEV)  # a mean far from zero: the sums are shifted by it
    with ops.precision(mode):
        y_ref = ops.conv2d_forward_raw(x, w, bias, stride)
        outs = []
        for rep in range(3):
            y = ops.conv2d_forward_raw(x, w, bias, stride, bn_stats=True)
            tag = y.__dict__.get("_lhg_bn_partial")
            assert tag is not None and tag[2] > 0, (mode, case, "no partial rows")
            pixels = y.shape[0] * y.shape[1] * y.shape[2]
            stats, rm, rv = torch.empty(2 * Co, device=DEV), torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
            call("lhg_bn_stats_finish", ptr(tag[1]), tag[2], ptr(tag[3]), pixels, Co, ptr(stats), ptr(rm), ptr(rv), 0.1, 1e-5, stream_ptr())
            outs.append((y, stats, rm, rv))
        assert torch.equal(outs[0][0], y_ref), (mode, case, "the stored output changed")
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(outs[0], o)), (mode, case, "repeat")
        y, stats, rm, rv = outs[0]
        stats0, rm0, rv0 = torch.empty(2 * Co, device=DEV), torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        ws = torch.empty((8192 * Co,), device=DEV)
        call("lhg_bn_stats", ptr(y), pixels, Co, Co, ptr(stats0), ptr(rm0), ptr(rv0), 0.1, 1e-5, ptr(ws), stream_ptr())
        # float64 statistics of the stored tensor: both routes are measured against them (bf16 storage: the epilogue sums the fp32
        # accumulators, the pass sums the rounded tensor — they differ by the rounding of y, not by a summation error)
        yd = y.double().reshape(pixels, Co)
        mean64, var64 = yd.mean(0), yd.var(0, unbiased=False)
        inv64 = 1.0 / torch.sqrt(var64 + 1e-5)
        stored_bf16 = y.dtype == torch.bfloat16
        tol_mean = (4e-3 if stored_bf16 else 2e-6) * (yd.abs().max().item() + 1e-30)
        tol_inv = 4e-3 if stored_bf16 else 5e-6
        e_mean, e_mean0 = (stats[:Co].double() - mean64).abs().max().item(), (stats0[:Co].double() - mean64).abs().max().item()
        e_inv, e_inv0 = (stats[Co:].double() / inv64 - 1).abs().max().item(), (stats0[Co:].double() / inv64 - 1).abs().max().item()
        assert e_mean <= tol_mean and e_inv <= tol_inv, (mode, case, e_mean, e_mean0, e_inv, e_inv0)
        assert (rm - rm0).abs().max().item() <= 0.1 * tol_mean * 2 and ((rv - rv0).abs() / rv0.abs()).max().item() <= (1e-2 if stored_bf16 else 2e-5), (mode, case)
    return e_mean, e_inv


@pytest.mark.parametrize("mode", ["default", "fp32", "fp32_split", "bf16"])
def test_epilogue_statistics_inprocess(mode):
    """Every arithmetic mode, autotuned tilings: the stored output is the plain call's bit for bit, the statistics finished from the
    partial rows agree with float64 statistics of the stored tensor as well as the pass does, three repeats give the same bits."""
    from learned_hologram_gan_amd import hip_ops as ops

    for case in CASES:
        _check(ops, mode, case)


def test_epilogue_statistics_bf16_storage():
    from learned_hologram_gan_amd import hip_ops as ops

    with ops.precision("bf16", storage="bf16"):
        for case in CASES[:4]:
            N, H, W, Ci, Co, k, stride = case
            _check_bf16_storage(ops, case)


def _check_bf16_storage(ops, case):
    from learned_hologram_gan_amd.native import call, ptr, stream_ptr

    N, H, W, Ci, Co, k, stride = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn((N, H, W, Ci), generator=g).to(DEV).to(torch.bfloat16)
    w = (torch.randn((Co, Ci, k, k), generator=g) * (Ci * k * k) ** -0.5).to(DEV)
    bias = torch.randn((Co,), generator=g).to(DEV)
    y_ref = ops.conv2d_forward_raw(x, w, bias, stride)
    y = ops.conv2d_forward_raw(x, w, bias, stride, bn_stats=True)
    assert y.dtype == torch.bfloat16 and torch.equal(y, y_ref)
    tag = y.__dict__["_lhg_bn_partial"]
    pixels = y.shape[0] * y.shape[1] * y.shape[2]
    stats = torch.empty(2 * Co, device=DEV)
    call("lhg_bn_stats_finish", ptr(tag[1]), tag[2], ptr(tag[3]), pixels, Co, ptr(stats), None, None, 0.1, 1e-5, stream_ptr())
    yd = y.double().reshape(pixels, Co)
    assert (stats[:Co].double() - yd.mean(0)).abs().max().item() <= 4e-3 * yd.abs().max().item()
    assert (stats[Co:].double() * torch.sqrt(yd.var(0, unbiased=False) + 1e-5) - 1).abs().max().item() <= 4e-3


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], ids=lambda v: f"ggs_variant{v}")
def test_epilogue_statistics_with_every_forced_tiling(variant):
    """The row count handed to the finish kernel is a table over the tiling variants (pixels per M tile, consumer-wave rows per tile,
    padded strip coordinates): force each variant of the default mode (LHG_GGS_VARIANT is read once per process, hence the child) and
    run the in-process check — a wrong entry folds uninitialised rows or drops real ones."""
    env = dict(os.environ, LHG_AUTOTUNE="0", LHG_GGS_VARIANT=str(variant))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                          "-k", "inprocess and default"], cwd=ROOT, env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-500:]


def test_batch_norm_after_conv_uses_the_partial_rows_and_matches_the_pass():
    """Op layer: Conv2dFn(..., "feeds_bn") -> BatchNormTrainFn takes the statistics from the conv's partial rows; with
    LHG_EPILOGUE_BN_STATS=0 semantics (tag removed) the same chain runs the pass.  Outputs, running statistics and gradients agree to
    the rounding of the statistics."""
    from learned_hologram_gan_amd import hip_ops as ops
    from learned_hologram_gan_amd.hip_ops import ACT_RELU

    torch.manual_seed(3)
    N, H, W, Ci, Co = 2, 40, 36, 64, 128
    x0 = torch.randn((N, H, W, Ci), device=DEV)
    w0 = torch.randn((Co, Ci, 3, 3), device=DEV) * (Ci * 9) ** -0.5
    b0 = torch.randn((Co,), device=DEV)
    gamma0, beta0 = torch.rand(Co, device=DEV) + 0.5, torch.randn(Co, device=DEV) * 0.1
    proj = torch.randn((N, H, W, Co), device=DEV)

    def run(use_rows):
        x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        gamma, beta = gamma0.clone().requires_grad_(True), beta0.clone().requires_grad_(True)
        rm, rv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        y = ops.Conv2dFn.apply(x, w, b, 1, "feeds_bn")
        assert "_lhg_bn_partial" in y.__dict__
        if not use_rows:
            del y.__dict__["_lhg_bn_partial"]
        z = ops.BatchNormTrainFn.apply(y, gamma, beta, rm, rv, None, ACT_RELU, 0.0, None)
        (z * proj).sum().backward()
        ops.join_side_stream(x.device)
        torch.cuda.synchronize()
        return z.detach(), rm, rv, x.grad, w.grad, gamma.grad, beta.grad

    a, b = run(True), run(False)
    for u, v, name in zip(a, b, ("z", "running_mean", "running_var", "gx", "gw", "ggamma", "gbeta")):
        scale = v.abs().max().item() + 1e-30
        assert (u.double() - v.double()).abs().max().item() / scale <= 2e-5, name
    _ = ctypes  # (kept for interactive use of the C entry points)
