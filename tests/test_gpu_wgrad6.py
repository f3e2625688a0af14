"""The tap-fused weight-gradient GEMM (csrc/wg6_kernel.inc, ABI 6: lhg_conv2d_backward_weight_into / lhg_conv_transpose2x2_backward_weight_into)
through the C ABI: every tile variant, split count and reduction form against a float64 reference, bit-identity across variants, the
in-launch reduction under repeated use of one ticket buffer, geometries with ragged rows / channel padding / more splits than steps."""

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def to_nhwc(x, ld=None):
    N, C, H, W = x.shape
    ld = ld or C
    out = torch.zeros(N, H, W, ld)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return out.to(DEV)


@pytest.fixture(scope="module")
def lib():
    from learned_hologram_gan_amd import hip_ops, native

    hip_ops.set_conv_precision("fp32_split_f16")
    yield native.load()
    native.load().lhg_wg6_force(-1, -1, -1)
    hip_ops.set_conv_precision("default")


def _variants(lib):
    return [(v, lib.lhg_wg6_variant_name(v).decode()) for v in range(lib.lhg_wg6_variants())]


def _fits(name, Cm, Cn, nt, stride):
    tile, _, t, ny, s = name.split()
    bm, bn = map(int, tile.split("x"))
    pad = lambda c: (c + 63) // 64 * 64  # noqa: E731
    return int(t[2:]) == nt and int(s[1:]) == stride and pad(Cm) % bm == 0 and pad(Cn) % bn == 0, int(ny[2:])


def _conv_wgrad(ops, x, gy, wshape, stride):
    with torch.no_grad():
        return ops.conv2d_weight_grad_raw(x, gy, wshape, stride)


# N, Ci, Co, H, W, k, stride
CONV = [
    (2, 64, 64, 24, 20, 3, 1),     # ragged rows (22-wide padded rows: two wraps inside one 32-position step)
    (1, 128, 256, 17, 23, 3, 1),   # odd extents
    (2, 96, 160, 9, 41, 3, 1),     # channels below the 64-padding (Cm = 96 -> 128, Cn = 160 -> 192)
    (2, 32, 64, 18, 22, 3, 2),     # stride 2, Cm = 32 on a 64-row tile
    (1, 128, 128, 31, 29, 3, 2),   # stride 2, odd extents
    (3, 256, 128, 8, 8, 1, 1),     # 1x1
    (1, 64, 128, 40, 36, 1, 1),
    (2, 256, 256, 6, 6, 3, 1),     # fewer steps than splits asked for
]


@pytest.mark.parametrize("case", CONV, ids=lambda c: "x".join(map(str, c)))
def test_conv_weight_gradient_every_variant_split_and_reduction_form(lib, case):
    """Every variant that fits the geometry x split counts {1, 3, 7} x {separate reduce launch, in-launch last arriver}: within 1e-5 of
    the float64 gradient, and — for one split count — bit-identical across variants and reduction forms (the same 16-position groups
    accumulate in the same order whatever the tile)."""
    from learned_hologram_gan_amd import hip_ops as ops

    N, Ci, Co, H, W, k, stride = case
    x, gy = rnd(N, Ci, H, W, seed=21), None
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    gy = rnd(N, Co, Ho, Wo, seed=22)
    wd = torch.zeros(Co, Ci, k, k, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wd, None, stride=stride, padding=k // 2).backward(gy.double())
    truth = wd.grad
    xh, gh = to_nhwc(x, ops.pad_to(Ci, 32)), to_nhwc(gy, ops.pad_to(Co, 4))
    ran = 0
    for S in (1, 3, 7):
        ref_bits = None
        for v, name in _variants(lib):
            ok, ny = _fits(name, Ci, Co, k, stride)
            if not ok or k % ny:
                continue
            for fused in (0, 1):
                lib.lhg_wg6_force(v, S, fused)
                gw = _conv_wgrad(ops, xh, gh, (Co, Ci, k, k), stride)
                torch.cuda.synchronize()
                e = rel_err(gw.cpu().double(), truth)
                assert e < 1e-5, (name, S, fused, e)
                if ref_bits is None:
                    ref_bits = gw.clone()
                else:
                    assert torch.equal(gw, ref_bits), (name, S, fused, (gw - ref_bits).abs().max().item())
                # accumulate into a slot: grad += dW
                slot = torch.full((Co, Ci, k, k), 0.25, device=DEV)
                with torch.no_grad():
                    ops.conv2d_weight_grad_raw(xh, gh, (Co, Ci, k, k), stride, slot)
                assert torch.equal(slot, gw + 0.25), (name, S, fused, "accumulate")
                ran += 1
    lib.lhg_wg6_force(-1, -1, -1)
    assert ran >= 6, ran
    # the library's own plan
    gw = _conv_wgrad(ops, xh, gh, (Co, Ci, k, k), stride)
    assert rel_err(gw.cpu().double(), truth) < 1e-5


@pytest.mark.parametrize("case", [(2, 64, 32, 8, 10), (1, 128, 64, 13, 11), (2, 256, 128, 5, 7)], ids=str)
def test_conv_transpose_weight_gradient_every_variant(lib, case):
    """ConvTranspose2d(2, stride 2): the strip operand is gy (twice the extent, two taps per kernel row)."""
    from learned_hologram_gan_amd import hip_ops as ops

    N, Ci, Co, H, W = case
    x, w = rnd(N, Ci, H, W, seed=1), torch.zeros(Ci, Co, 2, 2, dtype=torch.float64, requires_grad=True)
    gy = rnd(N, Co, 2 * H, 2 * W, seed=4)
    F.conv_transpose2d(x.double(), w, None, stride=2).backward(gy.double())
    truth = w.grad
    for S in (1, 4):
        ref_bits = None
        for v, name in _variants(lib):
            ok, _ = _fits(name, Co, Ci, 2, 2)  # strip channels = Co, point channels = Ci
            if not ok or "nt2" not in name:
                continue
            for fused in (0, 1):
                lib.lhg_wg6_force(v, S, fused)
                xg = to_nhwc(x).requires_grad_(True)
                wg = torch.zeros(Ci, Co, 2, 2, device=DEV, requires_grad=True)
                yg = ops.ConvTranspose2x2Fn.apply(xg, wg, None, None)
                yg.backward(to_nhwc(gy))
                torch.cuda.synchronize()
                e = rel_err(wg.grad.cpu().double(), truth)
                assert e < 1e-5, (name, S, fused, e)
                if ref_bits is None:
                    ref_bits = wg.grad.clone()
                else:
                    assert torch.equal(wg.grad, ref_bits), (name, S, fused)
    lib.lhg_wg6_force(-1, -1, -1)


def test_many_splits_take_the_split_parallel_reduce(lib):
    """S >= 32 with few outputs (the 64-channel layers at 384^2 run S = 128 .. 256): lhg_wgrad's reduce splits the S axis over the four waves
    of a workgroup and adds the quarters in order — against float64, equal to itself on repetition, and within rounding of the sequential
    order (S = 31 runs the one-thread-per-output kernel)."""
    from learned_hologram_gan_amd import hip_ops as ops

    N, Ci, Co, H, W = 4, 64, 64, 96, 96
    x, gy = rnd(N, Ci, H, W, seed=31), rnd(N, Co, H, W, seed=32)
    wd = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wd, None, padding=1).backward(gy.double())
    xh, gh = to_nhwc(x), to_nhwc(gy)
    got = {}
    for S in (31, 32, 33, 64, 147):
        lib.lhg_wg6_force(-1, S, 0)
        a = _conv_wgrad(ops, xh, gh, (Co, Ci, 3, 3), 1)
        b = _conv_wgrad(ops, xh, gh, (Co, Ci, 3, 3), 1)
        assert torch.equal(a, b), S
        assert rel_err(a.cpu().double(), wd.grad) < 1e-5, (S, rel_err(a.cpu().double(), wd.grad))
        slot = torch.full((Co, Ci, 3, 3), 0.5, device=DEV)
        with torch.no_grad():
            ops.conv2d_weight_grad_raw(xh, gh, (Co, Ci, 3, 3), 1, slot)
        assert torch.equal(slot, a + 0.5), (S, "accumulate")
        got[S] = a
    lib.lhg_wg6_force(-1, -1, -1)
    assert rel_err(got[32].cpu().double(), got[31].cpu().double()) < 1e-6


def test_in_launch_reduction_is_repeatable_under_load(lib):
    """The last-arriver reduction (agent-scope release / ticket / acquire) run 200 times back to back on a geometry with many splits,
    alternating with a second stream that keeps the chip busy: every result identical to the separate-launch reduction's bits."""
    from learned_hologram_gan_amd import hip_ops as ops

    N, Ci, Co, H, W = 4, 128, 128, 48, 48
    xh, gh = to_nhwc(rnd(N, Ci, H, W, seed=5)), to_nhwc(rnd(N, Co, H, W, seed=6))
    lib.lhg_wg6_force(-1, 9, 0)
    want = _conv_wgrad(ops, xh, gh, (Co, Ci, 3, 3), 1)
    lib.lhg_wg6_force(-1, 9, 1)
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=DEV)
    for it in range(200):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                junk = (junk @ junk).clamp_(-1, 1)
        got = _conv_wgrad(ops, xh, gh, (Co, Ci, 3, 3), 1)
        assert torch.equal(got, want), (it, (got - want).abs().max().item())
    torch.cuda.synchronize()
    lib.lhg_wg6_force(-1, -1, -1)


def test_structured_dynamic_range_per_channel(lib):
    """One strip channel and one point channel 2^-20 below the rest: per-channel scales keep their slices at full accuracy."""
    from learned_hologram_gan_amd import hip_ops as ops

    N, Ci, Co, H, W = 2, 64, 128, 24, 24
    x, gy = rnd(N, Ci, H, W, seed=7), rnd(N, Co, H, W, seed=8)
    x[:, 5] *= 2.0 ** -20
    gy[:, 9] *= 2.0 ** -20
    wd = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wd, None, padding=1).backward(gy.double())
    gw = _conv_wgrad(ops, to_nhwc(x), to_nhwc(gy), (Co, Ci, 3, 3), 1).cpu().double()
    for sl in ((slice(None), 5), (9, slice(None))):
        assert rel_err(gw[sl], wd.grad[sl]) < 1e-5, sl
