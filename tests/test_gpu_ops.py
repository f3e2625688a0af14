"""GPU parity of the individual HIP ops (through the C ABI) against plain PyTorch fp32 on the CPU.
Run on the MI355X box:  python -m pytest tests -m gpu -q
"""

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 2e-5  # fp32 GEMMs with different summation order


@pytest.fixture(scope="module")
def ops():
    from learned_hologram_gan_amd import hip_ops

    return hip_ops


def to_nhwc(x, ld=None):
    """CPU NCHW -> GPU NHWC (channels zero-padded to ld)."""
    N, C, H, W = x.shape
    ld = ld or C
    out = torch.zeros(N, H, W, ld)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return out.to(DEV)


def to_nchw(y, C=None):
    y = y.detach().cpu()
    C = C or y.shape[-1]
    return y[..., :C].permute(0, 3, 1, 2).contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


CONV_CASES = [
    # N, Ci, Co, H, W, k, stride
    (2, 64, 64, 24, 20, 3, 1),     # 256x64 or 64x64 tile, ragged M
    (1, 32, 128, 40, 40, 3, 1),    # 128-wide N tile
    (2, 128, 256, 16, 16, 3, 1),
    (1, 64, 6, 16, 24, 1, 1),      # narrow head
    (2, 4, 64, 16, 16, 3, 1),      # Ci padded 4 -> 32
    (2, 3, 32, 16, 16, 3, 1),      # critic first layer
    (2, 32, 64, 18, 22, 3, 2),     # stride 2
    (1, 1024, 1, 6, 6, 3, 1),      # critic head
    (3, 256, 128, 8, 8, 1, 1),     # 1x1
    (4, 64, 64, 96, 96, 3, 1),     # enough blocks for the 256x64 tile path
    (2, 128, 128, 64, 64, 3, 1),   # 128x128 tile path
    # thin convolutions (csrc/thin_conv.hip): ragged widths, every (taps, thin, wide) kernel instance
    (2, 4, 64, 15, 37, 1, 1),      # thin input 1x1
    (2, 3, 32, 17, 23, 3, 1),      # thin input, 3 real channels of a 32-padded tensor
    (2, 64, 6, 13, 29, 1, 1),      # thin output, 6 channels
    (1, 1024, 1, 7, 5, 3, 1),      # thin output, 4 chunks per lane
    (2, 64, 1, 11, 9, 3, 1),       # one output channel, 16 lanes per pixel
    (2, 1, 64, 9, 14, 3, 1),       # one input channel
    (1, 2, 8, 6, 70, 3, 1),        # two lanes per pixel, 32 pixels per wave
    (3, 256, 4, 5, 6, 3, 1),       # 64 lanes per pixel
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv2d_forward_and_gradients(ops, case):
    N, Ci, Co, H, W, k, stride = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5)
    b = rnd(Co, seed=3, scale=0.1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=stride, padding=k // 2)
    proj = rnd(*yr.shape, seed=4)
    (yr * proj).sum().backward()

    ld = ops.pad_to(Ci, 32)
    xg = to_nhwc(x, ld).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    yg = ops.Conv2dFn.apply(xg, wg, bg, stride, None)
    assert rel_err(to_nchw(yg), yr.detach()) < TOL
    (yg * to_nhwc(proj)).sum().backward()
    assert rel_err(to_nchw(xg.grad, Ci), xr.grad) < TOL
    if ld > Ci:
        assert xg.grad[..., Ci:].abs().max().item() == 0
    assert rel_err(wg.grad.cpu(), wr.grad) < 5e-5
    assert rel_err(bg.grad.cpu(), br.grad) < 5e-5


@pytest.mark.parametrize("case", [(2, 64, 64, 48, 48, 3, 1), (2, 128, 256, 24, 24, 3, 1), (2, 512, 512, 12, 12, 3, 1), (2, 64, 128, 24, 24, 1, 1),
                                  (2, 64, 128, 32, 32, 3, 2), (4, 1024, 1024, 8, 8, 3, 1), (1, 96, 160, 17, 23, 3, 1)], ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("mode,spread", [("fp32_split", 0), ("fp32_split_f16", 0), ("fp32_split_f16", 6)])
def test_split_gemm_is_fp32_faithful(ops, case, mode, spread):
    """The split GEMM modes — "fp32_split": every operand an exact sum of three bf16 terms, six MFMA products; "fp32_split_f16": tensor-
    scaled operands as sums of two fp16 terms, three products; fp32 accumulation in both — against a FLOAT64 convolution, next to the
    exact fp32 MFMA kernels ("fp32") on the same data: forward, input gradient and weight gradient must be as close to the truth as the
    exact kernels are (same order of rounding error: the sum is accumulated in fp32 either way) and far inside the 2e-5 the per-op
    parity tests use.  spread > 0: element magnitudes spread log-uniformly over 10^spread (the fp16 scheme's tensor scale keeps full
    relative accuracy over a 2^18 window and an absolute error below 2^-39 of the tensor maximum under it) at a gradient-like level."""
    N, Ci, Co, H, W, k, stride = case
    x = rnd(N, Ci, H, W, seed=11)
    w = rnd(Co, Ci, k, k, seed=12, scale=(Ci * k * k) ** -0.5)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    gy = rnd(N, Co, Ho, Wo, seed=13)
    if spread:
        g = torch.Generator().manual_seed(14)
        x = x * torch.pow(10.0, -spread * torch.rand(x.shape, generator=g)) * 1e3
        gy = gy * torch.pow(10.0, -spread * torch.rand(gy.shape, generator=g)) * 1e-7
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, stride=stride, padding=k // 2)
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), gy.double())
    truth = (yd.detach(), gxd, gwd)
    Cip = ops.pad_to(Ci, 32)
    xh, gh, wg = to_nhwc(x, Cip), to_nhwc(gy, ops.pad_to(Co, 4)), w.to(DEV)

    def errors(mode):
        ops.set_conv_precision(mode)
        try:
            with torch.no_grad():
                y = to_nchw(ops.conv2d_forward_raw(xh, wg, None, stride))
                gx = to_nchw(ops.Conv2dInputGradFn.apply(gh, wg, stride, H, W, Cip))[:, :Ci]
                gw = ops.Conv2dWeightGradFn.apply(xh, gh, w.shape, stride).cpu()
        finally:
            ops.set_conv_precision("default")
        return [rel_err(a.double(), b) for a, b in zip((y, gx, gw), truth)]

    e_split, e_exact = errors(mode), errors("fp32")
    for es, ee in zip(e_split, e_exact):
        assert es < 1e-5 and es <= 3.0 * ee + 5e-7, (e_split, e_exact)


@pytest.mark.parametrize("case", [(2, 64, 64, 48, 48, 3, 1), (2, 128, 256, 24, 24, 3, 1), (2, 64, 128, 32, 32, 3, 2), (2, 64, 128, 24, 24, 1, 1)],
                         ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("shift", [10, 20, 30])
def test_split_f16_gemm_structured_dynamic_range(ops, case, shift):
    """The default GEMM mode where a per-TENSOR scale is weakest: ONE input channel, ONE gout channel and ONE
    spatial quadrant of x and gy are scaled by 2^-shift, so a whole slice of every result sits far below the tensor's maximum, and each
    result is scored PER SLICE — max error over the slice relative to the slice's own maximum — against a float64 convolution, next to
    the exact fp32 MFMA kernels on the same data:

      forward        the quadrant's outputs            (every gathered activation there is 2^-shift below max|x|)
      input gradient the quadrant's gx                 (gathered gy 2^-shift below max|gy|)
      weight grad.   dW[:, c_small] and dW[n_small, :] (one operand's channel 2^-shift below the rest of its tensor)

    The fp16-split kernels must be as accurate on the small slices as the exact kernels (boosted residual plane for gathered
    activations, per-channel scales in the weight-gradient GEMMs); with the round-2 per-tensor scale the 2^-30 slices were off by 1e-3."""
    N, Ci, Co, H, W, k, stride = case
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    f = 2.0 ** -shift
    x = rnd(N, Ci, H, W, seed=21)
    gy = rnd(N, Co, Ho, Wo, seed=23)
    w = rnd(Co, Ci, k, k, seed=22, scale=(Ci * k * k) ** -0.5)
    c_small, n_small = 5, 9
    x[:, c_small] *= f
    gy[:, n_small] *= f
    x[:, :, : H // 2, : W // 2] *= f       # quadrant (the small channel there is 2^-2shift: its slice is scored on the other quadrants' rows)
    gy[:, :, : Ho // 2, : Wo // 2] *= f
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, stride=stride, padding=k // 2)
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), gy.double())
    Cip = ops.pad_to(Ci, 32)
    xh, gh, wg = to_nhwc(x, Cip), to_nhwc(gy, ops.pad_to(Co, 4)), w.to(DEV)
    m = 2  # rows / columns next to the quadrant's border mix both scales: scored with the large part
    slices = {
        "fwd quadrant": lambda t: t[:, :, : Ho // 2 - m, : Wo // 2 - m],
        "fwd rest": lambda t: t[:, :, Ho // 2:, :],
        "dgrad quadrant": lambda t: t[:, :, : H // 2 - m * stride, : W // 2 - m * stride],
        "dgrad rest": lambda t: t[:, :, H // 2:, :],
        "wgrad small input channel": lambda t: t[:, c_small],
        "wgrad small gout channel": lambda t: t[n_small],
        "wgrad rest": lambda t: t[n_small + 1:, c_small + 1:],
    }

    def run(mode):
        ops.set_conv_precision(mode)
        try:
            with torch.no_grad():
                y = to_nchw(ops.conv2d_forward_raw(xh, wg, None, stride))
                gx = to_nchw(ops.Conv2dInputGradFn.apply(gh, wg, stride, H, W, Cip))[:, :Ci]
                gw = ops.Conv2dWeightGradFn.apply(xh, gh, w.shape, stride).cpu()
        finally:
            ops.set_conv_precision("default")
        out = {}
        for name, sl in slices.items():
            got, ref = {"fwd": (y, yd.detach()), "dgrad": (gx, gxd), "wgrad": (gw, gwd)}[name.split()[0]]
            out[name] = rel_err(sl(got.double()), sl(ref))
        return out

    e_split, e_exact = run("fp32_split_f16"), run("fp32")
    for name in slices:
        assert e_split[name] < 2e-5 and e_split[name] <= 4.0 * e_exact[name] + 1e-6, (name, shift, e_split, e_exact)


def test_channel_absmax_matches_torch(ops):
    """lhg_channel_absmax: per-channel max|x| of an NHWC slice (the per-channel scales of the fp16-split weight-gradient GEMMs)."""
    from learned_hologram_gan_amd.native import call, ptr, stream_ptr

    for C, ld, pixels in ((64, 64, 5000), (32, 128, 777), (1024, 1024, 301), (96, 96, 64), (2048, 2048, 19)):
        t = torch.randn((pixels, ld), device=DEV) * torch.logspace(-6, 3, ld, device=DEV)
        t[3, 1] = -1e7
        out = torch.full((C,), 123.0, device=DEV)
        ws = torch.empty((2048 * C,), device=DEV)
        call("lhg_channel_absmax", ptr(t), pixels, C, ld, ptr(out), ptr(ws), stream_ptr())
        assert torch.equal(out, t[:, :C].abs().amax(dim=0)), (C, ld, pixels)


def test_bn_kernels_leave_the_per_channel_maxima_of_what_they_write(ops):
    """lhg_bn_apply_chanmax / lhg_bn_backward_chanmax + lhg_channel_absmax_finish (ABI 5; max|gres| ABI 7): the per-channel maxima of y / gx, bit for bit
    those of a pass over the tensor, and y / gx / the parameter gradients bit-identical to the plain calls'."""
    from learned_hologram_gan_amd.native import call, load, ptr, stream_ptr

    lib = load()
    for C, pixels, act in ((64, 4 * 96 * 96, 1), (256, 4 * 24 * 24, 2), (1024, 4 * 6 * 6, 1), (32, 1000, 0), (2048, 37, 2)):
        torch.manual_seed(C)
        x = torch.randn((pixels, C), device=DEV) * torch.logspace(-3, 2, C, device=DEV)
        gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        stats = torch.cat((x.mean(0), (x.var(0, unbiased=False) + 1e-5).rsqrt()))
        rows = int(lib.lhg_chanmax_partial_rows(pixels, C))
        assert 1 <= rows <= 2048
        y0, y1 = torch.empty_like(x), torch.empty_like(x)
        part = torch.full((rows * C,), float("nan"), device=DEV)
        call("lhg_bn_apply", ptr(x), C, pixels, C, ptr(stats), ptr(gamma), ptr(beta), None, 0, act, 0.2, ptr(y0), C, None, stream_ptr())
        call("lhg_bn_apply_chanmax", ptr(x), C, pixels, C, ptr(stats), ptr(gamma), ptr(beta), None, 0, act, 0.2, ptr(y1), C, None, ptr(part), stream_ptr())
        out = torch.full((C,), 123.0, device=DEV)
        call("lhg_channel_absmax_finish", ptr(part), pixels, C, ptr(out), stream_ptr())
        assert torch.equal(y0, y1) and torch.equal(out, y0.abs().amax(dim=0)), (C, pixels)
        gy = torch.randn((pixels, C), device=DEV) * torch.logspace(2, -4, C, device=DEV)
        res = []
        for fused in (False, True):
            gx, gr, gg, gb = torch.empty_like(x), torch.empty_like(x), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
            ws = torch.empty((8192 * C,), device=DEV)
            args = [ptr(gy), C, ptr(x), C, ptr(y0), C, pixels, C, ptr(stats), ptr(gamma), act, 0.2, ptr(gx), C, ptr(gr), C, ptr(gg), ptr(gb), 0, ptr(ws), None, ptr(beta)]
            if fused:
                part.fill_(float("nan"))
                part_res = torch.full_like(part, float("nan"))
                res_amax = torch.zeros(1, device=DEV)  # ABI 7: max|gres| max-accumulated into a zeroed slot
                call("lhg_bn_backward_chanmax", *args, ptr(part), ptr(part_res), ptr(res_amax), stream_ptr())
                for p_, t_ in ((part, gx), (part_res, gr)):
                    call("lhg_channel_absmax_finish", ptr(p_), pixels, C, ptr(out), stream_ptr())
                    assert torch.equal(out, t_.abs().amax(dim=0)), (C, pixels)
                assert res_amax.item() == gr.abs().max().item(), (C, pixels)
            else:
                call("lhg_bn_backward", *args, stream_ptr())
            res.append((gx, gr, gg, gb))
        assert all(torch.equal(a, b) for a, b in zip(*res)), (C, pixels)


FUSED_CASES = ((64, 4 * 96 * 96, 1), (256, 4 * 24 * 24, 2), (1024, 4 * 6 * 6, 1), (32, 1000, 0), (2048, 37, 2), (64, 8 * 192 * 192, 2), (96, 70001, 1),
               (4, 5000, 0), (512, 2 * 48 * 48, 1))


def test_fused_batch_norm_calls_equal_the_unfused_ones(ops):
    """ABI 9: lhg_bn_forward_train / lhg_bn_backward_fused / lhg_bn_backward_backward_fused / lhg_channel_sum_fused /
    lhg_channel_absmax_fused finish their reductions inside the launch that writes the partial rows (two-level last-arriver fold,
    fixed order, double accumulation).  Against the unfused entry points on the same data: statistics, sums and parameter gradients agree to
    fp32 rounding of the final cast (<= 2 ulp; the fold order of the partial rows differs), everything element-wise (y, gx, gres) is
    computed from them by the same kernel; per-channel maxima are EXACT; the tickets are back at zero after every call; fifty repeats of
    each call give the same bits (the result does not depend on which workgroup arrives last)."""
    from learned_hologram_gan_amd.native import call, load, ptr, stream_ptr

    lib = load()
    ws = torch.empty((int(lib.lhg_fused_workspace_floats()),), device=DEV)
    tickets = torch.zeros((int(lib.lhg_fused_ticket_count()),), dtype=torch.int32, device=DEV)

    def close(a, b, what, rtol=1e-6):
        scale = b.abs().max().item() + 1e-30
        err = (a.double() - b.double()).abs().max().item() / scale
        assert err <= rtol, (what, err)

    for C, pixels, act in FUSED_CASES:
        torch.manual_seed(C + pixels)
        ld = C if C % 32 else C + 32  # a channel slice of a wider buffer
        xb = torch.randn((pixels, ld), device=DEV) * torch.logspace(-3, 2, ld, device=DEV) + torch.linspace(-5, 5, ld, device=DEV)
        x = xb[:, :C]
        gamma, beta = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        res = torch.randn((pixels, C), device=DEV) if act == 1 else None
        # ---- forward: unfused reference
        stats0, rm0, rv0 = torch.empty(2 * C, device=DEV), torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        y0 = torch.empty((pixels, C), device=DEV)
        w0 = torch.empty((8192 * C,), device=DEV)
        call("lhg_bn_stats", ptr(x), pixels, C, ld, ptr(stats0), ptr(rm0), ptr(rv0), 0.1, 1e-5, ptr(w0), stream_ptr())
        call("lhg_bn_apply", ptr(x), ld, pixels, C, ptr(stats0), ptr(gamma), ptr(beta), ptr(res), C, act, 0.2, ptr(y0), C, None, stream_ptr())
        outs = []
        for rep in range(50 if pixels < 200000 else 5):
            stats1, rm1, rv1 = torch.empty(2 * C, device=DEV), torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
            y1, amax, cvec = torch.empty((pixels, C), device=DEV), torch.zeros(1, device=DEV), torch.full((C,), 123.0, device=DEV)
            call("lhg_bn_forward_train", ptr(x), ld, pixels, C, ptr(gamma), ptr(beta), ptr(rm1), ptr(rv1), 0.1, 1e-5, ptr(res), C, act, 0.2, ptr(y1), C,
                 ptr(stats1), ptr(amax), ptr(cvec) if C % 4 == 0 else None, 1, ptr(ws), ptr(tickets), stream_ptr())
            outs.append((stats1, rm1, rv1, y1, amax, cvec))
        assert int(tickets.abs().sum().item()) == 0
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(outs[0], o)), ("forward repeat", C, pixels)
        stats1, rm1, rv1, y1, amax, cvec = outs[0]
        close(stats1[:C], stats0[:C], ("mean", C, pixels))
        close(stats1[C:] / stats0[C:], torch.ones_like(stats0[C:]), ("invstd", C, pixels), 1e-6)
        close(rm1, rm0, ("running_mean", C, pixels))
        close(rv1, rv0, ("running_var", C, pixels), 1e-6)
        # the apply kernel is the same: with the reference's statistics it gives the reference's bits
        call("lhg_bn_apply", ptr(x), ld, pixels, C, ptr(stats1), ptr(gamma), ptr(beta), ptr(res), C, act, 0.2, ptr(y0), C, None, stream_ptr())
        assert torch.equal(y0, y1), (C, pixels)
        assert amax.item() == y1.abs().max().item() and torch.equal(cvec, y1.abs().amax(dim=0)), (C, pixels)
        if C % 4 == 0:  # chanmax_finish = 0 (the op layer's default): the apply launch leaves partial rows, a finish launch folds them later
            rows = int(lib.lhg_chanmax_partial_rows(pixels, C))
            part, y2, fin = torch.full((rows * C,), float("nan"), device=DEV), torch.empty_like(y1), torch.full((C,), 5.0, device=DEV)
            call("lhg_bn_forward_train", ptr(x), ld, pixels, C, ptr(gamma), ptr(beta), None, None, 0.1, 1e-5, ptr(res), C, act, 0.2, ptr(y2), C,
                 ptr(stats1.clone()), None, ptr(part), 0, ptr(ws), ptr(tickets), stream_ptr())
            call("lhg_channel_absmax_finish", ptr(part), pixels, C, ptr(fin), stream_ptr())
            assert torch.equal(y2, y1) and torch.equal(fin, cvec), (C, pixels)
        # ---- backward
        gy = torch.randn((pixels, C), device=DEV) * torch.logspace(2, -4, C, device=DEV)
        want_res = res is not None
        gx0, gr0, gg0, gb0 = torch.empty_like(y0), torch.empty_like(y0), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        call("lhg_bn_backward", ptr(gy), C, ptr(x), ld, ptr(y1), C, pixels, C, ptr(stats1), ptr(gamma), act, 0.2, ptr(gx0), C, ptr(gr0) if want_res else None, C,
             ptr(gg0), ptr(gb0), 0, ptr(w0), None, ptr(beta), stream_ptr())
        outs = []
        for rep in range(50 if pixels < 200000 else 5):
            gx1, gr1 = torch.empty_like(y0), torch.empty_like(y0)
            gg1, gb1 = torch.full((C,), 2.0, device=DEV), torch.full((C,), -3.0, device=DEV)  # accumulate = 1 adds to these
            am, amr = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
            cv, cvr = torch.full((C,), 7.0, device=DEV), torch.full((C,), 7.0, device=DEV)
            call("lhg_bn_backward_fused", ptr(gy), C, ptr(x), ld, ptr(y1), C, pixels, C, ptr(stats1), ptr(gamma), ptr(beta), act, 0.2, ptr(gx1), C,
                 ptr(gr1) if want_res else None, C, ptr(gg1), ptr(gb1), 1, ptr(am), ptr(amr) if want_res else None, ptr(cv),
                 ptr(cvr) if want_res else None, 1, ptr(ws), ptr(tickets), stream_ptr())
            outs.append((gx1, gr1 if want_res else gx1, gg1, gb1, am, cv, cvr if want_res else cv))
        assert int(tickets.abs().sum().item()) == 0
        for o in outs[1:]:
            assert all(torch.equal(a, b) for a, b in zip(outs[0], o)), ("backward repeat", C, pixels)
        gx1, gr1, gg1, gb1, am, cv, cvr = outs[0]
        close(gg1 - 2.0, gg0, ("ggamma", C, pixels), 2e-6)
        close(gb1 + 3.0, gb0, ("gbeta", C, pixels), 2e-6)
        close(gx1, gx0, ("gx", C, pixels), 2e-6)
        assert am.item() == gx1.abs().max().item() and torch.equal(cv, gx1.abs().amax(dim=0)), (C, pixels)
        if want_res:
            assert torch.equal(gr1, gr0) and torch.equal(cvr, gr1.abs().amax(dim=0)), (C, pixels)
        # ---- double backward (dense tensors)
        if pixels < 200000:
            xd, ggx = x.contiguous(), torch.randn((pixels, C), device=DEV)
            r0 = [torch.empty_like(y0), torch.empty_like(y0), torch.empty(C, device=DEV)]
            r1 = [torch.empty_like(y0), torch.empty_like(y0), torch.empty(C, device=DEV)]
            w2 = torch.empty(((5 * 4096 + 8) * C,), device=DEV)
            call("lhg_bn_backward_backward", ptr(ggx), ptr(gy), ptr(xd), ptr(y1), pixels, C, ptr(stats1), ptr(gamma), act, 0.2, *map(ptr, r0), ptr(w2), stream_ptr())
            call("lhg_bn_backward_backward_fused", ptr(ggx), ptr(gy), ptr(xd), ptr(y1), pixels, C, ptr(stats1), ptr(gamma), act, 0.2, *map(ptr, r1), ptr(ws),
                 ptr(tickets), stream_ptr())
            for a, b, what in zip(r1, r0, ("ggy", "gx2", "ggamma2")):
                close(a, b, (what, C, pixels), 5e-6)
        # ---- channel sum / channel maxima in one launch
        s0, s1 = torch.full((C,), 1.5, device=DEV), torch.full((C,), 1.5, device=DEV)
        call("lhg_channel_sum", ptr(x), pixels, C, ld, ptr(s0), 1, ptr(w0), stream_ptr())
        call("lhg_channel_sum_fused", ptr(x), pixels, C, ld, ptr(s1), 1, ptr(ws), ptr(tickets), stream_ptr())
        close(s1, s0, ("channel_sum", C, pixels), 1e-6)
        m1 = torch.full((C,), 9.0, device=DEV)
        call("lhg_channel_absmax_fused", ptr(x), pixels, C, ld, ptr(m1), ptr(ws), ptr(tickets), stream_ptr())
        assert torch.equal(m1, x.abs().amax(dim=0)), (C, pixels)
        assert int(tickets.abs().sum().item()) == 0


def test_fused_polar_jacobians_equal_the_torch_expressions():
    """lhg_polar_output_cotangent / lhg_polar_input_cotangent (ABI 9) against the differentiable torch expressions they replace in plain
    backward passes (asm_ops._output_cotangent / _input_cotangent), zeros of the field included, both scale factors, every mode."""
    from learned_hologram_gan_amd import asm_ops
    from learned_hologram_gan_amd.native import IN_PHASE, IN_POLAR, OUT_ABS, OUT_ABS_ANGLE

    g = torch.Generator().manual_seed(5)
    shape = (2, 3, 40, 56)
    z = torch.complex(torch.randn(shape, generator=g), torch.randn(shape, generator=g)).to(DEV)
    z[0, 0, :3] = 0  # zeros of the field: the cotangent is defined as zero there
    ga, gb = torch.randn(shape, generator=g).to(DEV), torch.randn(shape, generator=g).to(DEV)
    geom = asm_ops.Geometry(40, 56, 4, 4)
    for out_mode, g_b in ((OUT_ABS_ANGLE, gb), (OUT_ABS, None), (OUT_ABS_ANGLE, None)):
        for scale in (1.0, 1.0 / 4096.0):
            spec = asm_ops.Spec(geom, IN_POLAR, out_mode)
            with torch.enable_grad():
                want = asm_ops._output_cotangent(spec, z, ga, g_b, None, scale)
            with torch.no_grad():
                got = asm_ops._output_cotangent(spec, z, ga, g_b, None, scale)
            assert got.dtype == torch.complex64 and (got - want).abs().max().item() <= 2e-6 * want.abs().max().item(), (out_mode, scale)
            assert (got[0, 0, :3] == 0).all()
    a, phi = torch.rand(shape, generator=g).to(DEV) + 0.1, (torch.rand(shape, generator=g) * 6.28).to(DEV)
    for in_mode, b, ps in ((IN_POLAR, phi, 1.0), (IN_POLAR, phi, 6.2831853), (IN_PHASE, None, 1.0)):
        for pre in (1.0, 1048576.0):
            spec = asm_ops.Spec(geom, in_mode, OUT_ABS, ps)
            first = a if in_mode == IN_POLAR else phi
            with torch.enable_grad():
                want = asm_ops._input_cotangent(spec, first, b, z, pre)
            with torch.no_grad():
                got = asm_ops._input_cotangent(spec, first, b, z, pre)
            for w_, g_ in zip(want, got):
                assert (w_ is None) == (g_ is None)
                if w_ is not None:
                    assert (g_ - w_).abs().max().item() <= 3e-6 * w_.abs().max().item(), (in_mode, ps, pre)


def test_wgrad_slab_reduce_is_the_documented_sum_bit_for_bit(ops):
    """lhg_wgrad_reduce on nine-tap slabs:
    grad[d0][d1][t] (+)= the four (small weights: sixteen) interleaved slab slices summed in ascending order and combined in order — emulated here with fp32 torch
    adds, so the comparison is bit for bit; both orientations (conv weights: m is d1), ragged extents, accumulate on and off."""
    from learned_hologram_gan_amd.native import call, ptr, stream_ptr

    for (D0, D1, m_is_d1, S) in ((128, 64, 1, 7), (96, 80, 1, 4), (100, 90, 0, 3), (256, 128, 1, 1), (72, 130, 0, 9), (32, 16, 1, 5)):
        M, Nn = (D1, D0) if m_is_d1 else (D0, D1)
        m_pad, n_pad = (M + 63) // 64 * 64, (Nn + 63) // 64 * 64
        torch.manual_seed(D0 + S)
        slabs = torch.randn((S, 9, m_pad, n_pad), device=DEV) * torch.logspace(-3, 3, S, device=DEV).view(S, 1, 1, 1)
        total = 9 * D0 * D1
        slices = 16 if (total * 4 < S * 64 or total < 65536) else 4   # the small-weight kernel spends 16 threads per output on the slab axis
        parts = []
        for k in range(slices):
            a = torch.zeros((9, m_pad, n_pad), device=DEV)
            for s_ in range(k, S, slices):
                a = a + slabs[s_]
            parts.append(a)
        v = torch.zeros_like(parts[0])
        for a in parts:
            v = v + a
        v = v[:, :M, :Nn]                                           # [t][m][n]
        want = (v.permute(2, 1, 0) if m_is_d1 else v.permute(1, 2, 0)).contiguous()  # [d0][d1][t]
        for accumulate in (0, 1):
            grad = torch.full((D0, D1, 3, 3), 0.5, device=DEV)
            call("lhg_wgrad_reduce", ptr(slabs), S, 9, m_pad, n_pad, ptr(grad), D0, D1, m_is_d1, accumulate, stream_ptr())
            expect = want.reshape(D0, D1, 3, 3) + (0.5 if accumulate else 0.0)
            assert torch.equal(grad, expect), (D0, D1, m_is_d1, S, accumulate, (grad - expect).abs().max().item())


def test_nchw_to_nhwc_layout_kernels(ops):
    """lhg_nchw_to_nhwc: the pixel-per-thread form (ld <= 32) and the element-per-thread form, zero-filled padding channels."""
    from learned_hologram_gan_amd.native import call, ptr, stream_ptr

    torch.manual_seed(0)
    for (N, C, H, W, ld) in ((4, 4, 96, 96, 4), (2, 3, 64, 48, 4), (2, 6, 40, 40, 8), (2, 5, 33, 17, 32), (1, 1, 7, 9, 4), (2, 13, 20, 20, 16), (2, 40, 8, 8, 64)):
        x = torch.randn(N, C, H, W, device=DEV)
        y = torch.full((N, H, W, ld), 7.0, device=DEV)
        call("lhg_nchw_to_nhwc", ptr(x), ptr(y), N, C, H, W, ld, stream_ptr())
        want = torch.zeros((N, H, W, ld), device=DEV)
        want[..., :C] = x.permute(0, 2, 3, 1)
        assert torch.equal(y, want), (N, C, H, W, ld)


def test_default_gemm_mode_and_env_override():
    """The library starts in the fp32-faithful two-term fp16 split mode unless LHG_CONV_PRECISION names another one (read at load time)."""
    import os
    import subprocess
    import sys

    from learned_hologram_gan_amd import hip_ops

    want = os.environ.get("LHG_CONV_PRECISION", "fp32_split_f16")
    assert hip_ops.default_precision() == want and hip_ops.conv_precision() == want
    code = "from learned_hologram_gan_amd import hip_ops; print(hip_ops.conv_precision())"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, LHG_CONV_PRECISION="fp32"), capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "fp32", out.stderr[-500:]


@pytest.mark.parametrize("case", [(2, 64, 64, 24, 20, 3, 1), (1, 32, 128, 40, 40, 3, 1), (2, 128, 256, 16, 16, 3, 1), (2, 32, 64, 18, 22, 3, 2),
                                  (3, 256, 128, 8, 8, 1, 1), (2, 128, 128, 64, 64, 3, 1)], ids=lambda c: "x".join(map(str, c)))
def test_conv2d_bf16_operand_mode(ops, case):
    """lhg_set_conv_precision(bf16): operands rounded to bf16 (nearest even), fp32 accumulation.  Checked against fp32 convs of the
    bf16-ROUNDED operands (tight: only the summation order differs) and against the unrounded fp32 conv (loose: bf16 has 8 bits)."""
    N, Ci, Co, H, W, k, stride = case
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5)
    b = rnd(Co, seed=3, scale=0.1)
    rb = lambda t: t.bfloat16().float()  # noqa: E731
    y_exact = F.conv2d(x, w, b, stride=stride, padding=k // 2)
    y_ref = F.conv2d(rb(x), rb(w), b, stride=stride, padding=k // 2)
    proj = rnd(*y_ref.shape, seed=4)
    gx_ref = torch.autograd.grad((F.conv2d(xr := rb(x).requires_grad_(True), rb(w), b, stride=stride, padding=k // 2) * rb(proj)).sum(), xr)[0]
    ops.set_conv_precision("bf16")
    try:
        assert ops.conv_precision() == "bf16"
        xg = to_nhwc(x).requires_grad_(True)
        wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        yg = ops.Conv2dFn.apply(xg, wg, bg, stride, None)
        assert rel_err(to_nchw(yg), y_ref) < TOL
        assert rel_err(to_nchw(yg), y_exact) < 2e-2
        (yg * to_nhwc(proj)).sum().backward()
        assert rel_err(to_nchw(xg.grad), gx_ref) < TOL  # input gradient: gy and W rounded to bf16
        wgrad_ref = torch.autograd.grad((F.conv2d(rb(x), wr := w.clone().requires_grad_(True), b, stride=stride, padding=k // 2) * rb(proj)).sum(), wr)[0]
        assert rel_err(wg.grad.cpu(), wgrad_ref) < 5e-5  # weight gradient: x and gy rounded to bf16, fp32 accumulation
    finally:
        ops.set_conv_precision("default")
    # back in fp32 mode the packed panels are fp32 again
    y32 = ops.Conv2dFn.apply(to_nhwc(x), w.to(DEV), b.to(DEV), stride, None)
    assert rel_err(to_nchw(y32), y_exact) < TOL


def test_thin_conv_epilogues(ops):
    """Thin-input conv with bias / folded-BN affine / activation, and the planar sigmoid head (thin output)."""
    N, H, W = 2, 11, 19
    x, w, b = rnd(N, 4, H, W, seed=1), rnd(64, 4, 3, 3, seed=2, scale=0.2), rnd(64, seed=3)
    sc, sh = rnd(64, seed=4) + 1.5, rnd(64, seed=5)
    assert ops.thin_mode(4, 64, 3, 1) == 1 and ops.thin_mode(64, 6, 1, 1) == 2 and ops.thin_mode(64, 64, 3, 1) == 0
    ref = F.relu((F.conv2d(x, w, b, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)))
    y = ops.conv2d_forward_raw(to_nhwc(x, 32), w.to(DEV), b.to(DEV), 1, act=ops.ACT_RELU, scale=sc.to(DEV), shift=sh.to(DEV))
    assert rel_err(to_nchw(y), ref) < TOL
    x2, w2, b2 = rnd(N, 64, H, W, seed=6), rnd(6, 64, 1, 1, seed=7, scale=0.2), rnd(6, seed=8)
    ref2 = torch.sigmoid(F.conv2d(x2, w2, b2))
    y2 = ops.conv2d_forward_raw(to_nhwc(x2), w2.to(DEV), b2.to(DEV), 1, act=ops.ACT_SIGMOID, planar=True)
    assert y2.shape == (N, 6, H, W) and rel_err(y2.cpu(), ref2) < TOL
    # round 5: a thin-INPUT conv whose output feeds a GEMM directly (eval-mode chains) measures max|y| itself (lhg_conv2d_thin_forward_amax),
    # and the 64 -> 6 head finishes one output per lane (reduce-scatter): odd widths, every lane count of the tail
    if ops.conv_precision() == "fp32_split_f16":
        ym = ops.conv2d_forward_raw(to_nhwc(x, 32), w.to(DEV), b.to(DEV), 1, act=ops.ACT_RELU, scale=sc.to(DEV), shift=sh.to(DEV), measure_out=True)
        tag = ym.__dict__.get("_lhg_amax")
        assert tag is not None and tag[1].item() == ym.abs().max().item() and torch.equal(ym, y)
    for (n_, h_, w_) in ((1, 5, 3), (2, 7, 66), (1, 1, 257)):
        x3, w3, b3 = rnd(n_, 64, h_, w_, seed=9), rnd(6, 64, 1, 1, seed=10, scale=0.3), rnd(6, seed=11)
        for planar in (True, False):
            y3 = ops.conv2d_forward_raw(to_nhwc(x3), w3.to(DEV), b3.to(DEV), 1, act=ops.ACT_SIGMOID, planar=planar)
            got = y3.cpu() if planar else to_nchw(y3, 6)
            assert rel_err(got, torch.sigmoid(F.conv2d(x3, w3, b3))) < TOL, (n_, h_, w_, planar)


def test_thin_conv_reads_the_nchw_input_directly(ops):
    """lhg_conv2d_thin_forward_nchw (ABI 10): the thin-input convs of an eval-mode generator's first block on the reference's NCHW frame —
    the same multiply-adds as the NHWC(32) route (conversion pass, then lhg_conv2d_thin_forward), so the same bits: ragged extents,
    3x3 and 1x1, 4 and 3 input channels, folded affine + activation, max|y|."""
    for (N, Ci, Co, H, W, k, act) in ((2, 4, 64, 17, 37, 3, ops.ACT_RELU), (1, 4, 64, 9, 70, 1, ops.ACT_NONE), (2, 3, 32, 16, 23, 3, ops.ACT_LEAKY),
                                      (1, 4, 64, 40, 520, 3, ops.ACT_RELU)):
        X = rnd(N, Ci, H, W, seed=11).to(DEV)
        w = rnd(Co, Ci, k, k, seed=12, scale=0.3).to(DEV)
        b, sc, sh = rnd(Co, seed=13).to(DEV), (rnd(Co, seed=14) + 1.5).to(DEV), rnd(Co, seed=15).to(DEV)
        x = ops.ToNHWC.apply(X, 32)
        want = ops.conv2d_forward_raw(x, w, b, 1, act=act, slope=0.2, scale=sc, shift=sh, measure_out=True)
        got = ops.conv2d_thin_forward_nchw(X, w, b, act=act, slope=0.2, scale=sc, shift=sh, measure_out=True)
        assert torch.equal(got, want), (N, Ci, Co, H, W, k)
        assert float(got.__dict__["_lhg_amax"][1]) == float(want.abs().max()), (N, Ci, Co, H, W, k)


def test_thin_residual_epilogue_gives_the_two_kernel_routes_bits(ops):
    """lhg_conv2d_forward_thin_res (ABI 10): the 3x3 conv of an eval-mode first ResidualBlock with the 1x1 shortcut of the NCHW frame evaluated
    inside the GEMM's epilogue, against shortcut tensor + lhg_conv2d_forward(res=...): bit-identical output and max|y| — ragged extents (tiles
    that straddle two images, padding columns of the strip kernels), 4 and 3 frame channels, fewer than 64 output channels."""
    for (N, Cx, Co, H, W, act) in ((3, 4, 64, 24, 23, ops.ACT_RELU), (2, 3, 64, 40, 36, ops.ACT_NONE), (1, 4, 32, 33, 130, ops.ACT_RELU), (2, 4, 64, 96, 96, ops.ACT_LEAKY)):
        X = rnd(N, Cx, H, W, seed=21).to(DEV)
        y1 = to_nhwc(rnd(N, 64, H, W, seed=22))
        w2 = rnd(Co, 64, 3, 3, seed=23, scale=0.05).to(DEV)
        b2, sc, sh = rnd(Co, seed=24).to(DEV), (rnd(Co, seed=25) + 1.5).to(DEV), rnd(Co, seed=26).to(DEV)
        w3, b3 = rnd(Co, Cx, 1, 1, seed=27, scale=0.5).to(DEV), rnd(Co, seed=28).to(DEV)
        skip = ops.conv2d_thin_forward_nchw(X, w3, b3)
        want = ops.conv2d_forward_raw(y1, w2, b2, 1, act=act, slope=0.1, scale=sc, shift=sh, res=skip, measure_out=True)
        got = ops.conv2d_forward_thin_res(y1, w2, b2, X, w3, b3, act=act, slope=0.1, scale=sc, shift=sh, measure_out=True)
        assert got is not None and torch.equal(got, want), (N, Cx, Co, H, W, (got - want).abs().max().item() if got is not None else None)
        assert float(got.__dict__["_lhg_amax"][1]) == float(want.abs().max())


def test_thin_conv_double_backward(ops):
    """Gradient-penalty pattern through a thin-input conv and a thin-output conv: d/dw of || d sum(y) / d x ||^2."""
    N, H, W = 2, 10, 12
    x = rnd(N, 3, H, W, seed=1)
    w1, b1 = rnd(32, 3, 3, 3, seed=2, scale=0.3), rnd(32, seed=3, scale=0.1)
    w2, b2 = rnd(1, 32, 3, 3, seed=4, scale=0.1), rnd(1, seed=5, scale=0.1)

    def penalty(conv1, conv2, xin):
        s = conv2(F.leaky_relu(conv1(xin), 0.2)) if conv1.__name__ == "ref1" else conv2(conv1(xin))
        (g,) = torch.autograd.grad(s.sum(), xin, create_graph=True)
        return (g * g).sum()

    xr = x.clone().requires_grad_(True)
    pr = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]

    def ref1(t):
        return F.conv2d(t, pr[0], pr[1], padding=1)

    penalty(ref1, lambda t: F.conv2d(t, pr[2], pr[3], padding=1), xr).backward()

    xg = to_nhwc(x, 32).requires_grad_(True)
    pg = [t.to(DEV).requires_grad_(True) for t in (w1, b1, w2, b2)]

    def hip1(t):
        return ops.ConvBiasActFn.apply(t, pg[0], pg[1], 1, ops.ACT_LEAKY, 0.2)

    penalty(hip1, lambda t: ops.Conv2dFn.apply(t, pg[2], pg[3], 1, None), xg).backward()
    assert rel_err(pg[0].grad.cpu(), pr[0].grad) < 1e-4 and rel_err(pg[2].grad.cpu(), pr[2].grad) < 1e-4
    assert rel_err(to_nchw(xg.grad, 3), xr.grad) < 1e-4


def test_conv2d_fused_epilogue(ops):
    N, Ci, Co, H, W = 2, 64, 64, 12, 12
    x, w, b = rnd(N, Ci, H, W, seed=1), rnd(Co, Ci, 3, 3, seed=2, scale=0.05), rnd(Co, seed=3)
    sc, sh, res = rnd(Co, seed=4) + 1.5, rnd(Co, seed=5), rnd(N, Co, H, W, seed=6)
    ref = F.leaky_relu((F.conv2d(x, w, b, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)) + res, 0.2)
    y = ops.conv2d_forward_raw(to_nhwc(x), w.to(DEV), b.to(DEV), 1, act=ops.ACT_LEAKY, slope=0.2, scale=sc.to(DEV), shift=sh.to(DEV),
                               res=to_nhwc(res))
    assert rel_err(to_nchw(y), ref) < TOL
    # planar sigmoid head
    w1, b1 = rnd(6, Ci, 1, 1, seed=7, scale=0.2), rnd(6, seed=8)
    yp = ops.conv2d_forward_raw(to_nhwc(x), w1.to(DEV), b1.to(DEV), 1, act=ops.ACT_SIGMOID, planar=True)
    assert rel_err(yp.cpu(), torch.sigmoid(F.conv2d(x, w1, b1))) < TOL
    # output into a channel slice of a wider buffer
    buf = torch.full((N, H, W, 128), 7.0, device=DEV)
    ops.conv2d_forward_raw(to_nhwc(x), w.to(DEV), b.to(DEV), 1, out=ops.OutSlot(buf[..., 64:]))
    assert rel_err(to_nchw(buf[..., 64:]), F.conv2d(x, w, b, padding=1)) < TOL
    assert (buf[..., :64] == 7.0).all()


@pytest.mark.parametrize("k,stride", [(3, 1), (1, 1), (3, 2)], ids=["3x3", "1x1", "3x3s2"])
@pytest.mark.parametrize("act", [0, 1, 2, 3], ids=["none", "relu", "leaky", "sigmoid"])
def test_gather_gemm_epilogue_writes_its_window_and_nothing_else(ops, k, stride, act):
    """The tile store goes through a buffer descriptor based at the tile's first pixel (csrc/gg_epilogue.inc): padding rows of the strip
    kernel, rows past the end of the last tile and columns past Co get offsets the descriptor rejects.  Output into a channel slice of a
    wider NaN-filled buffer at odd extents, Co = 48 (a partial 32-column block), residual from a slice with another row pitch, every
    activation: the slice equals the float64 reference, max|y| is exact, and every byte outside the slice is still NaN."""
    N, Ci, Co, H, W = 3, 64, 48, 13, 11
    x, w, b = rnd(N, Ci, H, W, seed=11), rnd(Co, Ci, k, k, seed=12, scale=0.05), rnd(Co, seed=13)
    Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
    res = rnd(N, Co, Ho, Wo, seed=14)
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=k // 2) + res.double()
    ref = {0: ref, 1: F.relu(ref), 2: F.leaky_relu(ref, 0.2), 3: torch.sigmoid(ref)}[act]
    buf = torch.full((N, Ho, Wo, 160), float("nan"), device=DEV)
    rbuf = torch.zeros((N, Ho, Wo, 96), device=DEV)
    rbuf[..., 32:80] = res.permute(0, 2, 3, 1).to(DEV)
    y = ops.conv2d_forward_raw(to_nhwc(x), w.to(DEV), b.to(DEV), stride, act=act, slope=0.2, res=rbuf[..., 32:80],
                               out=ops.OutSlot(buf[..., 64:112]), measure_out=True)
    torch.cuda.synchronize()
    got = buf[..., 64:112].permute(0, 3, 1, 2).cpu().double()
    assert rel_err(got, ref) < TOL, (k, stride, act)
    assert torch.isnan(buf[..., :64]).all() and torch.isnan(buf[..., 112:]).all(), "stores outside the output slice"
    known = y.__dict__.get("_lhg_amax")
    if known is not None:  # (the fp16-split mode measures max|y| on the way out)
        assert known[1].item() == buf[..., 64:112].abs().max().item()


@pytest.mark.parametrize("case", [(2, 64, 32, 8, 10), (1, 128, 64, 12, 12), (2, 1024, 512, 2, 2)], ids=str)
def test_conv_transpose2x2(ops, case):
    N, Ci, Co, H, W = case
    x, w, b = rnd(N, Ci, H, W, seed=1), rnd(Ci, Co, 2, 2, seed=2, scale=Ci ** -0.5), rnd(Co, seed=3, scale=0.1)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, stride=2)
    proj = rnd(*yr.shape, seed=4)
    (yr * proj).sum().backward()
    xg = to_nhwc(x).requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    yg = ops.ConvTranspose2x2Fn.apply(xg, wg, bg, None)
    assert rel_err(to_nchw(yg), yr.detach()) < TOL
    (yg * to_nhwc(proj)).sum().backward()
    assert rel_err(to_nchw(xg.grad), xr.grad) < TOL
    assert rel_err(wg.grad.cpu(), wr.grad) < 5e-5
    assert rel_err(bg.grad.cpu(), br.grad) < 5e-5


@pytest.mark.parametrize("act,slope,with_res", [(1, 0.0, True), (2, 0.2, False), (0, 0.0, False)])
def test_batch_norm_train_forward_backward(ops, act, slope, with_res):
    N, C, H, W = 3, 64, 10, 14
    x = rnd(N, C, H, W, seed=1) * 2 + 0.7
    gam, bet = rnd(C, seed=2) + 1.5, rnd(C, seed=3)
    res = rnd(N, C, H, W, seed=4) if with_res else None
    rm, rv = rnd(C, seed=5), rnd(C, seed=6).abs() + 0.5
    xr, gr, br = x.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if with_res else None
    rm_r, rv_r = rm.clone(), rv.clone()
    pre = F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.1, 1e-5)
    if with_res:
        pre = pre + rr
    yr = F.relu(pre) if act == 1 else (F.leaky_relu(pre, slope) if act == 2 else pre)
    proj = rnd(*yr.shape, seed=7)
    (yr * proj).sum().backward()

    xg = to_nhwc(x).requires_grad_(True)
    gg, bg = gam.to(DEV).requires_grad_(True), bet.to(DEV).requires_grad_(True)
    rg = to_nhwc(res).requires_grad_(True) if with_res else None
    rm_g, rv_g = rm.to(DEV), rv.to(DEV)
    yg = ops.BatchNormTrainFn.apply(xg, gg, bg, rm_g, rv_g, rg, act, slope, None)
    assert rel_err(to_nchw(yg), yr.detach()) < TOL
    assert rel_err(rm_g.cpu(), rm_r) < 1e-5 and rel_err(rv_g.cpu(), rv_r) < 1e-5
    (yg * to_nhwc(proj)).sum().backward()
    assert rel_err(to_nchw(xg.grad), xr.grad) < 1e-4
    assert rel_err(gg.grad.cpu(), gr.grad) < 1e-4
    assert rel_err(bg.grad.cpu(), br.grad) < 1e-4
    if with_res:
        assert rel_err(to_nchw(rg.grad), rr.grad) < TOL


def test_batch_norm_large_mean_stability(ops):
    """Shifted one-pass variance: mean >> std must not lose the variance."""
    N, C, H, W = 2, 32, 64, 64
    x = rnd(N, C, H, W, seed=1) * 0.01 + 100.0
    ref = F.batch_norm(x, None, None, None, None, True)
    y = ops.BatchNormTrainFn.apply(to_nhwc(x), torch.ones(C, device=DEV), torch.zeros(C, device=DEV), None, None, None, 0, 0.0, None)
    assert rel_err(to_nchw(y), ref) < 2e-3


def test_maxpool(ops):
    x = rnd(2, 64, 12, 16, seed=1)
    x[0, :, 0, 0] = x[0, :, 0, 1]  # a tie: first maximum wins
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2, 2)
    proj = rnd(*yr.shape, seed=2)
    (yr * proj).sum().backward()
    xg = to_nhwc(x).requires_grad_(True)
    yg = ops.MaxPool2x2Fn.apply(xg)
    assert torch.equal(to_nchw(yg), yr.detach())
    (yg * to_nhwc(proj)).sum().backward()
    assert torch.equal(to_nchw(xg.grad), xr.grad)


def test_maxpool_with_skip_adds_the_second_consumers_gradient_in_its_backward_kernel(ops):
    """maxpool2x2_with_skip (lhg_maxpool2x2_backward_add, ABI 8): (pool(x), alias of x); the gradient through the alias — here a STRIDED
    channel slice, as the skip concatenation's gradient is — is added by the pool's backward kernel: equal to autograd's own sum bit
    for bit (one fp32 add per element either way); each output alone works too."""
    x = rnd(2, 64, 12, 16, seed=1)
    proj, skip = rnd(2, 64, 6, 8, seed=2), rnd(2, 64, 12, 16, seed=3)
    xr = x.clone().requires_grad_(True)
    (F.max_pool2d(xr, 2, 2) * proj).sum().backward()
    wide = torch.zeros(2, 12, 16, 128, device=DEV)
    wide[..., 32:96] = to_nhwc(skip)
    for use_pool, use_skip in ((True, True), (True, False), (False, True)):
        xg = to_nhwc(x).requires_grad_(True)
        yg, xs = ops.maxpool2x2_with_skip(xg)
        assert torch.equal(to_nchw(yg), F.max_pool2d(x, 2, 2)) and xs.data_ptr() == xg.data_ptr()
        loss = (yg * to_nhwc(proj)).sum() if use_pool else 0
        if use_skip:
            loss = loss + (xs * wide[..., 32:96]).sum()  # d loss / d xs is the strided slice itself
        loss.backward()
        want = (xr.grad if use_pool else 0) + (skip if use_skip else 0)
        assert torch.equal(to_nchw(xg.grad), want), (use_pool, use_skip)


def test_layout_roundtrip_and_grad(ops):
    x = rnd(2, 3, 8, 12, seed=1)
    xg = x.to(DEV).requires_grad_(True)
    h = ops.ToNHWC.apply(xg, 32)
    assert h.shape == (2, 8, 12, 32) and torch.equal(to_nchw(h, 3), x) and h[..., 3:].abs().max() == 0
    back = ops.ToNCHW.apply(h, 3)
    assert torch.equal(back.cpu(), x)
    (back * 2).sum().backward()
    assert torch.equal(xg.grad.cpu(), torch.full_like(x, 2.0))


def _critic_like(ops, x, w1, b1, w2, b2, gam, bet, w3, b3, on_gpu):
    """conv+leaky -> conv(s2)+BN+leaky -> conv(->1): the op sequence of the critic, both backends."""
    if on_gpu:
        h = ops.ToNHWC.apply(x, 32)
        h = ops.ConvBiasActFn.apply(h, w1, b1, 1, ops.ACT_LEAKY, 0.2)
        y = ops.Conv2dFn.apply(h, w2, b2, 2, None)
        h = ops.BatchNormTrainFn.apply(y, gam, bet, None, None, None, ops.ACT_LEAKY, 0.2, None)
        s = ops.Conv2dFn.apply(h, w3, b3, 1, None)
        return s.reshape(s.shape[0], -1)
    h = F.leaky_relu(F.conv2d(x, w1, b1, padding=1), 0.2)
    h = F.leaky_relu(F.batch_norm(F.conv2d(h, w2, b2, stride=2, padding=1), None, None, gam, bet, True, 0.1, 1e-5), 0.2)
    return F.conv2d(h, w3, b3, padding=1).flatten(1)


def test_gradient_penalty_double_backward(ops):
    """WGAN-GP: d/dtheta of (||d sum D(x)/dx|| - 1)^2 through conv, train-mode BN and LeakyReLU."""
    g = torch.Generator().manual_seed(5)
    x = torch.rand((2, 3, 16, 16), generator=g)
    shapes = dict(w1=(32, 3, 3, 3), b1=(32,), w2=(64, 32, 3, 3), b2=(64,), gam=(64,), bet=(64,), w3=(1, 64, 3, 3), b3=(1,))
    P = {k: (torch.rand(s, generator=g) - 0.5) * (0.4 if k.startswith("w") else 0.2) + (1.0 if k == "gam" else 0.0) for k, s in shapes.items()}

    def run(on_gpu):
        dev = DEV if on_gpu else "cpu"
        p = {k: v.clone().to(dev).requires_grad_(True) for k, v in P.items()}
        xi = x.clone().to(dev).requires_grad_(True)
        score = _critic_like(ops, xi, on_gpu=on_gpu, **p)
        (gx,) = torch.autograd.grad(score, xi, torch.ones_like(score), create_graph=True, retain_graph=True)
        gp = ((gx.view(2, -1).norm(2, dim=1) - 1) ** 2).mean()
        loss = score.mean() + 10 * gp
        loss.backward()
        return score.detach().cpu(), gx.detach().cpu(), gp.item(), {k: v.grad.cpu() for k, v in p.items()}

    s_r, gx_r, gp_r, gr = run(False)
    s_g, gx_g, gp_g, gg = run(True)
    assert rel_err(s_g, s_r) < 1e-4
    assert rel_err(gx_g, gx_r) < 1e-4
    assert abs(gp_g - gp_r) <= 1e-4 * abs(gp_r)
    for k in gr:
        if k == "b2":  # bias feeding a train-mode BN: analytically zero gradient, noise on both sides
            continue
        assert rel_err(gg[k], gr[k]) < 5e-4, k


def test_adam_matches_torch(ops):
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(1000, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    pg, m, v = p0.clone().to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for step in range(1, 4):
        grad = torch.randn(1000, generator=g)
        pr.grad = grad.clone()
        opt.step()
        ops.adam_step_(pg, grad.to(DEV), m, v, 1e-3, 0.9, 0.999, 1e-8, step)
        assert rel_err(pg.cpu(), pr.detach()) < 1e-6


def test_poh_encode_forward_backward_matches_composite():
    """AP2POH tail (stencil, per-plane max normalisation, angle, acos, checkerboard) vs the tensor-op composite."""
    from learned_hologram_gan_amd.poh_ops import PohEncodeFn
    from oracle import nets

    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 20, 28
    field = torch.complex(torch.randn((B, 3, H, W), generator=g), torch.randn((B, 3, H, W), generator=g))
    taps = torch.rand((3, 3), generator=g) + 0.2
    bias = torch.randn((3,), generator=g) * 0.1
    proj = torch.randn((B, 3, H, W), generator=g)

    fr, tr, br = field.clone().requires_grad_(True), taps.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    sd = {f"p.conv_{c}.params": tr[i] for i, c in enumerate("rgb")}
    sd.update({f"p.conv_{c}.bias": br[i:i + 1] for i, c in enumerate("rgb")})
    mod = torch.complex(nets.channelwise_symmetric_conv(sd, "p.", fr.real), nets.channelwise_symmetric_conv(sd, "p.", fr.imag))
    poh_r = nets.double_phase_encode(nets.normalize_amplitude(mod.abs()), mod.angle())
    (torch.cos(poh_r) * proj).sum().backward()

    fg, tg, bg = field.to(DEV).requires_grad_(True), taps.to(DEV).requires_grad_(True), bias.to(DEV).requires_grad_(True)
    poh_g = PohEncodeFn.apply(fg, tg, bg)
    assert (torch.exp(1j * poh_g.detach().cpu()) - torch.exp(1j * poh_r.detach())).abs().max() < 1e-4
    (torch.cos(poh_g) * proj.to(DEV)).sum().backward()
    assert rel_err(torch.view_as_real(fg.grad.cpu()), torch.view_as_real(fr.grad)) < 2e-4
    assert rel_err(tg.grad.cpu(), tr.grad) < 2e-4 and rel_err(bg.grad.cpu(), br.grad) < 2e-4


def test_recon_losses_match_reference_definitions():
    from learned_hologram_gan_amd.poh_ops import ReconLossFn
    from oracle import losses as L

    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 24, 20
    ha, ta = torch.rand((B, 3, H, W), generator=g), torch.rand((B, 3, H, W), generator=g)
    hp, tp = torch.rand((B, 3, H, W), generator=g) * 6.28 - 3.14, torch.rand((B, 3, H, W), generator=g) * 6.28 - 3.14
    wts = torch.tensor([1.0, 5.0, 3.0])
    har, hpr = ha.clone().requires_grad_(True), hp.clone().requires_grad_(True)
    ref = torch.stack((L.focal_sincos_phase_gradient_loss(hpr, tp), L.pixel_loss(har, ta), L.total_variation_loss(har, ta)))
    (ref * wts).sum().backward()
    hag, hpg = ha.to(DEV).requires_grad_(True), hp.to(DEV).requires_grad_(True)
    out = ReconLossFn.apply(hag, ta.to(DEV), hpg, tp.to(DEV))
    assert rel_err(out.detach().cpu(), ref.detach()) < 1e-5
    (out * wts.to(DEV)).sum().backward()
    assert rel_err(hag.grad.cpu(), har.grad) < 1e-4
    assert rel_err(hpg.grad.cpu(), hpr.grad) < 1e-4


def test_stale_gradient_slot_is_never_inherited(ops):
    """A weight-gradient slot registered for a parameter must not be picked up by an unrelated tensor that reuses its address."""
    w = torch.nn.Parameter(rnd(64, 64, 3, 3, seed=1, scale=0.05).to(DEV))
    slot = torch.zeros_like(w)
    ops.register_grad_slot(w, slot)
    addr = w.data_ptr()
    assert ops._grad_slot(w) is slot
    del w
    torch.cuda.empty_cache()
    other = torch.empty((64, 64, 3, 3), device=DEV)  # may or may not land on the same address
    if other.data_ptr() == addr:
        assert ops._grad_slot(other) is None
    assert addr not in ops._GRAD_SLOTS or ops._GRAD_SLOTS[addr][0]() is None or other.data_ptr() != addr
    x = to_nhwc(rnd(1, 64, 8, 8, seed=2)).requires_grad_(True)
    w2 = rnd(64, 64, 3, 3, seed=3, scale=0.05).to(DEV).requires_grad_(True)
    ops.Conv2dFn.apply(x, w2, None, 1, None).sum().backward()
    assert w2.grad is not None and torch.isfinite(w2.grad).all()


def test_conv2d_randomised_shape_sweep(ops):
    """Seeded sweep over ragged extents, channel counts that are not tile multiples, both kernel sizes and strides: forward, input
    gradient, weight gradient and bias gradient against PyTorch fp32 on the CPU (covers the MFMA engine's padding / masking paths and
    the dispatch to the thin kernels)."""
    import random

    rng = random.Random(1234)
    channels_in, channels_out = (1, 3, 4, 6, 32, 64, 96, 128), (1, 4, 6, 32, 64, 96, 160)
    for case in range(28):
        N, H, W = rng.randint(1, 3), rng.randint(5, 40), rng.randint(5, 40)
        Ci, Co, k = rng.choice(channels_in), rng.choice(channels_out), rng.choice((1, 3))
        stride = rng.choice((1, 2)) if k == 3 else 1
        x = rnd(N, Ci, H, W, seed=10 * case + 1)
        w = rnd(Co, Ci, k, k, seed=10 * case + 2, scale=(Ci * k * k) ** -0.5)
        b = rnd(Co, seed=10 * case + 3, scale=0.1)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = F.conv2d(xr, wr, br, stride=stride, padding=k // 2)
        proj = rnd(*yr.shape, seed=10 * case + 4)
        (yr * proj).sum().backward()
        ld = ops.pad_to(Ci, 32)
        xg = to_nhwc(x, ld).requires_grad_(True)
        wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        yg = ops.Conv2dFn.apply(xg, wg, bg, stride, None)
        tag = (case, N, Ci, Co, H, W, k, stride)
        assert yg.shape == (N, yr.shape[2], yr.shape[3], Co), tag
        assert rel_err(to_nchw(yg), yr.detach()) < TOL, tag
        (yg * to_nhwc(proj)).sum().backward()
        assert rel_err(to_nchw(xg.grad, Ci), xr.grad) < TOL, tag
        assert ld == Ci or xg.grad[..., Ci:].abs().max().item() == 0, tag
        assert rel_err(wg.grad.cpu(), wr.grad) < 5e-5, tag
        assert rel_err(bg.grad.cpu(), br.grad) < 5e-5, tag


@pytest.mark.parametrize("shape", [(4, 3, 384, 384), (2, 3, 37, 53), (5, 1, 64, 12)])
def test_psnr_ssim_kernels_match_definitions(shape):
    """csrc/metrics.hip against the oracle's float64 restatement of the torchmetrics defaults and against the host-side expression."""
    from learned_hologram_gan_amd.poh_ops import psnr_ssim
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import psnr, ssim
    from oracle import losses

    g = torch.Generator().manual_seed(11)
    x = torch.rand(shape, generator=g)
    y = (x + 0.15 * (torch.rand(shape, generator=g) - 0.4)).clamp(0, 1.4)
    got = psnr_ssim(x.to(DEV), y.to(DEV)).cpu()
    assert abs(got[0].item() - losses.psnr(x, y).item()) < 1e-3
    assert abs(got[1].item() - losses.ssim(x, y).item()) < 2e-5
    assert abs(got[1].item() - ssim(x, y).item()) < 2e-5 and abs(got[0].item() - psnr(x, y).item()) < 1e-3
    same = psnr_ssim(x.to(DEV), x.to(DEV)).cpu()
    assert abs(same[1].item() - 1.0) < 1e-6 and torch.isinf(same[0])


# ----------------------------------------------------------------------------- bf16 activation storage (BASELINE configs[2] / [4])
@pytest.fixture
def bf16_storage(ops):
    ops.set_activation_storage("bf16")
    try:
        yield ops
    finally:
        ops.set_activation_storage("fp32")
    assert ops.activation_storage() == "fp32" and ops.conv_precision() == ops.default_precision()


def _rb(t):
    return t.bfloat16().float()


@pytest.mark.parametrize("case", [(2, 64, 64, 24, 20, 3, 1), (1, 32, 128, 40, 40, 3, 1), (2, 128, 256, 16, 16, 3, 1), (2, 32, 64, 18, 22, 3, 2),
                                  (3, 256, 128, 8, 8, 1, 1), (2, 4, 64, 32, 32, 3, 1), (2, 64, 6, 32, 32, 1, 1)], ids=lambda c: "x".join(map(str, c)))
def test_bf16_storage_conv_family(bf16_storage, case):
    """bf16 storage mode: NHWC activations and their gradients are bf16 in HBM, the GEMMs read them without conversion, accumulate in
    fp32 and round once on the way out.  Against fp32 convolutions of the SAME bf16 values: only the output rounding (2^-9) and the
    summation order differ.  The thin layers (4 -> 64, 64 -> 6) take the MFMA path here."""
    ops = bf16_storage
    N, Ci, Co, H, W, k, stride = case
    x, w, b = _rb(rnd(N, Ci, H, W, seed=1)), rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5), rnd(Co, seed=3, scale=0.1)
    xr, wr = x.clone().requires_grad_(True), _rb(w).requires_grad_(True)
    y_ref = F.conv2d(xr, wr, b, stride=stride, padding=k // 2)
    proj = _rb(rnd(*y_ref.shape, seed=4))
    gx_ref, gw_ref = torch.autograd.grad((y_ref * proj).sum(), (xr, wr))
    xg = to_nhwc(x, ops.pad_to(Ci, 32)).bfloat16().requires_grad_(True)
    wg, bg = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    yg = ops.Conv2dFn.apply(xg, wg, bg, stride, None)
    assert yg.dtype == torch.bfloat16
    assert rel_err(to_nchw(yg.float()), y_ref.detach()) < 6e-3
    (yg.float() * to_nhwc(proj)).sum().backward()
    assert xg.grad.dtype == torch.bfloat16 and wg.grad.dtype == torch.float32
    assert rel_err(to_nchw(xg.grad.float())[:, :Ci], gx_ref) < 6e-3
    assert rel_err(wg.grad.cpu(), gw_ref) < 2e-3  # fp32 accumulation of bf16 x bf16 products, fp32 result
    assert rel_err(bg.grad.cpu(), proj.sum(dim=(0, 2, 3))) < 1e-4


def test_bf16_storage_batch_norm_pool_and_layout(bf16_storage):
    ops = bf16_storage
    N, C, H, W = 2, 64, 12, 20
    x = _rb(rnd(N, C, H, W, seed=5) * 2 + 0.3)
    gamma, beta = rnd(C, seed=6) + 1.5, rnd(C, seed=7)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = F.relu(F.batch_norm(xr, None, None, gr, br, True, 0.1, 1e-5))
    proj = _rb(rnd(N, C, H, W, seed=8))
    gx_ref, gg_ref, gb_ref = torch.autograd.grad((y_ref * proj).sum(), (xr, gr, br))
    # layout kernels: NCHW fp32 <-> NHWC bf16
    xg = ops.ToNHWC.apply(x.to(DEV), C)
    assert xg.dtype == torch.bfloat16 and torch.equal(ops.ToNCHW.apply(xg, C).cpu(), x)
    xg = xg.detach().requires_grad_(True)
    gg, bb = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yg = ops.BatchNormTrainFn.apply(xg, gg, bb, rm, rv, None, ops.ACT_RELU, 0.0, None)
    assert yg.dtype == torch.bfloat16 and rel_err(to_nchw(yg.float()), y_ref.detach()) < 6e-3
    (yg.float() * to_nhwc(proj)).sum().backward()
    assert rel_err(to_nchw(xg.grad.float()), gx_ref) < 2e-2  # the mask comes from the ROUNDED output: a few borderline pixels flip
    assert rel_err(gg.grad.cpu(), gg_ref) < 2e-2 and rel_err(bb.grad.cpu(), gb_ref) < 2e-2
    assert rel_err(rm.cpu(), 0.1 * x.mean(dim=(0, 2, 3))) < 1e-4
    # max pool: exact on bf16 values, gradient routed to the first maximum
    p = ops.MaxPool2x2Fn.apply(xg.detach())
    assert torch.equal(to_nchw(p.float()), F.max_pool2d(x, 2))


# --------------------------------------------------------------------------- tensor scales of the fp16-split GEMM mode
def test_absmax_kernel_and_fused_measurements(ops):
    """lhg_absmax max-accumulates max|x| over an NHWC slice (vector and scalar paths, non-finite values order above finite ones); the
    BN kernels measure the same quantity for the tensor they write (y_absmax / gx_absmax) and tag it on the tensor object."""
    if ops.conv_precision() != "fp32_split_f16":
        pytest.skip("tensor scales belong to the fp32_split_f16 mode")
    g = torch.Generator().manual_seed(3)
    t = (torch.randn((2, 9, 7, 96), generator=g) * 3).to(DEV)
    assert float(ops.operand_absmax(t)[0]) == float(t.abs().max())
    sl = t[..., 32:64]                                  # channel slice: ld 96, C 32
    assert float(ops.operand_absmax(sl)[0]) == float(sl.abs().max())
    odd = torch.randn((1, 5, 3, 6), generator=g).to(DEV)  # C % 4 != 0: scalar path
    assert float(ops.operand_absmax(odd)[0]) == float(odd.abs().max())
    z = torch.zeros((1, 4, 4, 32), device=DEV)
    assert float(ops.operand_absmax(z)[0]) == 0.0
    bad = t.clone()
    bad[1, 2, 3, 4] = float("inf")
    assert torch.isinf(ops.operand_absmax(bad)).all()
    bad[0, 0, 0, 0] = float("nan")
    assert torch.isnan(ops.operand_absmax(bad)).all()
    # the slot is remembered on the tensor together with its version counter ...
    a1 = ops.operand_absmax(t)
    assert ops.operand_absmax(t).data_ptr() == a1.data_ptr()
    t.mul_(2.0)                                         # ... and an in-place update invalidates it
    a2 = ops.operand_absmax(t)
    assert a2.data_ptr() != a1.data_ptr() and float(a2[0]) == float(t.abs().max())
    # fused measurements: BatchNormTrainFn output and its input-gradient
    x = (torch.randn((3, 10, 14, 64), generator=g) * 2 + 0.7).to(DEV).requires_grad_(True)
    gam, bet = torch.rand(64, generator=g).to(DEV) + 0.5, torch.randn(64, generator=g).to(DEV)
    rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    y = ops.BatchNormTrainFn.apply(x, gam.requires_grad_(True), bet.requires_grad_(True), rm, rv, None, 1, 0.0, None)
    tag = y.__dict__.get("_lhg_amax")
    assert tag is not None and float(tag[1][0]) == float(y.detach().abs().max())
    pooled = ops.MaxPool2x2Fn.apply(y)
    assert pooled.__dict__["_lhg_amax"][1].data_ptr() == tag[1].data_ptr()  # inherits the input's bound
    gy = torch.randn(y.shape, generator=g).to(DEV)
    gx, _ = ops.bn_backward_raw(gy, x.detach(), y.detach(), gam.detach(), *_bn_stats(ops, x.detach()), 1, 0.0, False, None, None, False)
    assert float(gx.__dict__["_lhg_amax"][1][0]) == float(gx.abs().max())


def test_absmax_slot_ring_recycles_and_invalidates_tags(ops, monkeypatch):
    """The slot pools form a ring that is never freed (a weight-gradient GEMM on the side stream may still read a slot after Python let
    go of it); a recycled pool starts from zero again and the tags that pointed into it are not honoured any more."""
    if ops.conv_precision() != "fp32_split_f16":
        pytest.skip("tensor scales belong to the fp32_split_f16 mode")
    monkeypatch.setattr(ops, "_RING_POOLS", 2)
    monkeypatch.setattr(ops, "_POOL_SLOTS", 4)
    monkeypatch.setattr(ops, "_AMAX_POOL", {})
    t = torch.full((1, 2, 2, 32), 3.0, device=DEV)
    first = ops.operand_absmax(t)
    assert float(first[0]) == 3.0 and ops.operand_absmax(t).data_ptr() == first.data_ptr()
    ptrs = {first.data_ptr()}
    for i in range(8):                                  # two full turns of the ring
        o = ops.operand_absmax(torch.full((1, 2, 2, 32), float(i + 10), device=DEV))
        assert float(o[0]) == float(i + 10)             # a recycled slot starts from zero again
        ptrs.add(o.data_ptr())
    again = ops.operand_absmax(t)                       # the old tag points into a recycled pool: measured afresh
    assert float(again[0]) == 3.0 and not ops._slot_alive(first)
    assert len(ptrs | {again.data_ptr()}) <= 8          # 2 pools x 4 slots, reused in place


def _bn_stats(ops, x):
    from learned_hologram_gan_amd.native import call, ptr, stream_ptr

    px, N, H, W, Cc, ld = ops.nhwc(x)
    stats = torch.empty((2 * Cc,), dtype=torch.float32, device=x.device)
    ws = torch.empty((4104 * Cc,), dtype=torch.float32, device=x.device)
    call("lhg_bn_stats", px, N * H * W, Cc, ld, ptr(stats), None, None, 0.1, 1e-5, ptr(ws), stream_ptr())
    return (stats,)


def test_autotune_choices_persist_in_the_cache_file(tmp_path):
    """LHG_TUNE_CACHE=<file>: a process appends its tiling choices, a later process reads them back (and times nothing it finds there)."""
    import os
    import subprocess
    import sys

    cache = tmp_path / "tune.txt"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch; from learned_hologram_gan_amd import hip_ops as ops\n"
            "torch.manual_seed(0)\n"
            "x = torch.randn(2, 24, 24, 64, device='cuda'); w = torch.randn(64, 64, 3, 3, device='cuda')\n"
            "y = ops.conv2d_forward_raw(x, w, None, 1); torch.cuda.synchronize(); print(float(y.abs().sum()))")
    env = dict(os.environ, LHG_TUNE_CACHE=str(cache), LHG_AUTOTUNE="1")  # (an inherited LHG_AUTOTUNE=0 is honoured by every launcher now)
    a = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True)
    assert a.returncode == 0, a.stderr[-800:]
    lines = cache.read_text().strip().splitlines()
    assert lines and all(ln.startswith("lhg-tune-") for ln in lines)
    b = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True)
    assert b.returncode == 0 and b.stdout.strip().splitlines()[-1] == a.stdout.strip().splitlines()[-1]
    assert cache.read_text().strip().splitlines() == lines  # nothing new was tuned


@pytest.mark.parametrize("scale", [1e-4, 1.0, 3e4], ids=lambda s: f"x{s:g}")
def test_f16_split_tracks_exact_fp32_over_operand_scales(ops, scale):
    """A conv -> conv -> conv-transpose chain (no normalisation in between, so the operand magnitudes really differ by `scale`^k from
    layer to layer) and its gradients in the default two-term fp16 mode against the exact fp32 MFMA kernels on the same data: the
    per-tensor power-of-two scales must keep the result at fp32 accuracy whether the activations are ~1e-8 or ~1e9."""
    if ops.conv_precision() != "fp32_split_f16":
        pytest.skip("the default mode was overridden")
    g = torch.Generator().manual_seed(17)
    x0 = (torch.randn((2, 20, 24, 64), generator=g) * scale).to(DEV)
    w1 = (torch.randn((128, 64, 3, 3), generator=g) * (64 * 9) ** -0.5 * scale).to(DEV)
    w2 = (torch.randn((128, 128, 3, 3), generator=g) * (128 * 9) ** -0.5).to(DEV)
    wt = (torch.randn((128, 64, 2, 2), generator=g) * 128 ** -0.5 * scale).to(DEV)
    proj = torch.randn((2, 40, 48, 64), generator=g).to(DEV)

    def run(mode):
        ops.set_conv_precision(mode)
        try:
            x = x0.clone().requires_grad_(True)
            ws = [w.clone().requires_grad_(True) for w in (w1, w2, wt)]
            h = ops.Conv2dFn.apply(x, ws[0], None, 1, None)
            h = ops.Conv2dFn.apply(h, ws[1], None, 1, None)
            y = ops.ConvTranspose2x2Fn.apply(h, ws[2], None, None)
            (y * proj).sum().backward()
            torch.cuda.synchronize()
            ops.join_side_stream()
            return [y.detach().clone(), x.grad.clone()] + [w.grad.clone() for w in ws]
        finally:
            ops.set_conv_precision("default")

    got, want = run("fp32_split_f16"), run("fp32")
    for a, b in zip(got, want):
        assert torch.isfinite(a).all() and rel_err(a.cpu().double(), b.cpu().double()) < 3e-6, scale


@pytest.mark.parametrize("variant", [5, 8], ids=lambda v: f"gg4s_variant{v}")
def test_strip_kernel_is_exercised_when_forced(variant):
    """The autotuner decides per geometry whether a 3x3 stride-1 launch runs on the strip kernel (gg4s) — on the small shapes of this
    file it may never win.  Force it (64 x 64 and 128 x 128 tiles; LHG_GGS_VARIANT is read once per process, hence the child process)
    and run the convolution parity tests, ragged extents and the fp64 comparison included."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LHG_AUTOTUNE="0", LHG_GGS_VARIANT=str(variant))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_ops.py"), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider",
                          "-k", "conv2d_forward_and_gradients or split_gemm or randomised_shape_sweep or fused_epilogue"],
                         cwd=root, env=env, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-500:]


@pytest.mark.parametrize("mode", ["fp32_split_f16", "fp32_split", "fp32", "bf16"])
def test_batched_weight_pack_equals_per_weight_packs(ops, mode):
    """lhg_pack_weights (one call after an optimiser step) writes bit-for-bit what the lazy per-weight lhg_pack_weight writes, for both
    panel orientations and more weights than one batch of kernel arguments holds; forms that are still current are left alone."""
    prev = ops.conv_precision()
    ops.set_conv_precision(mode)
    try:
        torch.manual_seed(21)
        code = ops._mode()
        shapes = [(64, 32, 3, 3), (128, 64, 3, 3), (32, 64, 2, 2), (70, 33, 1, 1), (1024, 512, 3, 3)] + [(64, 64, 3, 3)] * 34
        ws = [torch.randn(s, device=DEV) * (10.0 ** (i % 5 - 2)) for i, s in enumerate(shapes)]
        for w in ws:  # first use: the lazy path
            ops.pack_weight(w, True)
            ops.pack_weight(w, False)
        assert ops.repack_weights(ws) == 0  # nothing is stale
        for w in ws[:-1]:
            w.mul_(1.5).add_(0.01)  # what an optimiser step does: new values, new version, same storage
        held = [w.__dict__["_lhg_packed"][(True, ops.pad_to(w.shape[1], 32), code)][1] for w in ws]
        untouched = held[-1].clone()
        assert ops.repack_weights(ws) == 2 * (len(ws) - 1)  # two batches of kernel arguments
        got = [(ops.pack_weight(w, True), ops.pack_weight(w, False)) for w in ws]  # cache hits now: the same buffers, rewritten in place
        assert all(g[0] is h for g, h in zip(got, held)) and torch.equal(held[-1], untouched)
        for w, (a, b) in zip(ws, got):
            fresh = w.clone()  # no cache on the clone: packed by lhg_pack_weight
            fa, fb = ops.pack_weight(fresh, True), ops.pack_weight(fresh, False)
            bits = lambda t: t.reshape(-1).view(torch.int32)[: t.numel() // 2] if mode == "bf16" else t.view(torch.int32)  # noqa: E731  (bf16 panels fill half of the fp32-sized buffer)
            assert torch.equal(bits(a), bits(fa)) and torch.equal(bits(b), bits(fb))
            if mode == "fp32_split_f16":  # max|w| behind the panels
                tail = lambda t: torch.as_strided(t, (1,), (1,), t.numel())  # noqa: E731
                assert float(tail(a)) == float(w.abs().max()) == float(tail(fa))
        assert ops.repack_weights(ws) == 0
    finally:
        ops.set_conv_precision(prev)


def test_optimizer_step_leaves_every_packed_form_current(ops):
    """FusedAdam.step re-packs, in its batched call, the forms the convs hold: the next forward / backward find them in the cache."""
    from learned_hologram_gan_amd import optim

    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(32, 64, 3, padding=1), torch.nn.Conv2d(64, 32, 3, padding=1)).to(DEV)
    opt = optim.FusedAdam(optim.FlatParams(net), lr=1e-2)
    x = torch.randn(2, 12, 12, 32, device=DEV, requires_grad=True)

    def run():
        h = ops.Conv2dFn.apply(x, net[0].weight, net[0].bias, 1, None)
        y = ops.Conv2dFn.apply(h, net[1].weight, net[1].bias, 1, None)
        opt.zero_grad()
        y.square().mean().backward()
        return y

    run()
    opt.step()
    for conv in net:
        w = conv.weight
        stamp = (w.data_ptr(), w._version, tuple(w.shape))
        forms = w.__dict__["_lhg_packed"]
        assert len(forms) == 2 and all(s == stamp for s, _ in forms.values())
    before = {id(v[1]) for conv in net for v in conv.weight.__dict__["_lhg_packed"].values()}
    y1 = run()
    assert before == {id(v[1]) for conv in net for v in conv.weight.__dict__["_lhg_packed"].values()}  # no re-pack on use
    ref = F.conv2d(F.conv2d(x.detach().permute(0, 3, 1, 2), net[0].weight, net[0].bias, padding=1), net[1].weight, net[1].bias, padding=1)
    assert rel_err(y1.detach().permute(0, 3, 1, 2).cpu(), ref.cpu()) < TOL


@pytest.mark.parametrize("stride,with_1x1", [(1, True), (2, True), (1, False)])
def test_shared_input_conv_adds_the_skip_gradient_in_the_epilogue(ops, stride, with_1x1):
    """Conv2dSharedInputFn: d/dx of conv3x3(x) and of the skip path (conv1x1(x) or x itself) summed inside the input-gradient GEMM —
    the same gradient as the two-consumer graph gives with autograd's own accumulation (ResidualBlock, ref neural_network_components.py:22-31)."""
    torch.manual_seed(17)
    N, C, H, W, Co = 2, 64, 20, 24, 64 if not with_1x1 else 96
    x0 = torch.randn(N, C, H, W)
    w1, b1 = torch.randn(Co, C, 3, 3) * 0.05, torch.randn(Co) * 0.1
    k3 = 1 if stride == 1 else 3  # (a strided 1x1 input-gradient leaves pixels without taps: not part of the model, not supported)
    w3, b3 = torch.randn(Co, C, k3, k3) * 0.1, torch.randn(Co) * 0.1
    gy = torch.randn(N, Co, (H + stride - 1) // stride, (W + stride - 1) // stride)
    gs = torch.randn_like(gy) if with_1x1 else torch.randn(N, C, H, W)

    xr = x0.clone().requires_grad_(True)
    yr = F.conv2d(xr, w1, b1, stride=stride, padding=1)
    sr = F.conv2d(xr, w3, b3, stride=stride, padding=k3 // 2) if with_1x1 else xr
    ((yr * gy).sum() + (sr * gs).sum()).backward()

    x = to_nhwc(x0).requires_grad_(True)
    W1, B1, W3, B3 = (t.to(DEV).requires_grad_(True) for t in (w1, b1, w3, b3))
    y, xs = ops.conv2d_shared_input(x, W1, B1, stride, None)
    assert xs is not x and xs.data_ptr() == x.data_ptr()
    s = ops.Conv2dFn.apply(xs, W3, B3, stride, None) if with_1x1 else xs
    ((y * to_nhwc(gy)).sum() + (s * to_nhwc(gs)).sum()).backward()
    assert rel_err(to_nchw(y), yr.detach()) < TOL
    assert rel_err(to_nchw(x.grad), xr.grad) < TOL
    assert rel_err(W1.grad.cpu(), torch.autograd.grad(F.conv2d(x0, w1.requires_grad_(True), b1, stride=stride, padding=1), w1, gy)[0]) < TOL


_VARIANT_CHILD = r"""
import hashlib, torch
from learned_hologram_gan_amd import hip_ops as ops
torch.manual_seed(3)
x = torch.randn(2, 48, 40, 128, device="cuda"); w = torch.randn(128, 128, 3, 3, device="cuda") * 0.05
gy = torch.randn(2, 48, 40, 128, device="cuda"); gy2 = torch.randn(2, 24, 20, 128, device="cuda")
with torch.no_grad():
    y = ops.conv2d_forward_raw(x, w, None, 1)
    gx = ops.Conv2dInputGradFn.apply(gy, w, 1, 48, 40, 128)
    gx2 = ops.Conv2dInputGradFn.apply(gy2, w, 2, 48, 40, 128)
    slot = torch.zeros(128, 128, 3, 3, device="cuda")
    ops.conv2d_weight_grad_raw(x, gy, (128, 128, 3, 3), 1, slot)
    # the tile store's window (csrc/gg_epilogue.inc) under THIS variant: odd extents, Co = 80 (a partial column block), output into a
    # channel slice of a NaN-filled buffer, residual with another row pitch, ReLU — outside the slice every byte must still be NaN
    xs = torch.randn(3, 13, 11, 128, device="cuda"); ws = torch.randn(80, 128, 3, 3, device="cuda") * 0.05
    buf = torch.full((3, 13, 11, 192), float("nan"), device="cuda"); rbuf = torch.randn(3, 13, 11, 96, device="cuda")
    ops.conv2d_forward_raw(xs, ws, None, 1, act=ops.ACT_RELU, res=rbuf[..., 8:88], out=ops.OutSlot(buf[..., 64:144]))
    torch.cuda.synchronize()
    assert torch.isnan(buf[..., :64]).all() and torch.isnan(buf[..., 144:]).all() and not torch.isnan(buf[..., 64:144]).any(), "stray stores"
    ys = buf[..., 64:144].contiguous()
torch.cuda.synchronize()
print("HASH", " ".join(hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16] for t in (y, gx, gx2, slot, ys)))
"""


_BF16_CLASSES_CHILD = r"""
import hashlib, torch
from learned_hologram_gan_amd import hip_ops as ops
torch.manual_seed(5)
out = []
for storage in ("fp32", "bf16"):
    with ops.precision("bf16", storage=storage):
        dt = torch.bfloat16 if storage == "bf16" else torch.float32
        w = torch.randn(128, 64, 3, 3, device="cuda") * 0.05
        gy2 = torch.randn(2, 24, 20, 128, device="cuda").to(dt)
        x = torch.randn(2, 12, 10, 128, device="cuda").to(dt)
        wt = torch.randn(128, 64, 2, 2, device="cuda") * 0.1
        with torch.no_grad():
            gx2 = ops.Conv2dInputGradFn.apply(gy2, w, 2, 48, 40, 64)                      # stride-2 input gradient: four parity classes
            up = ops.ConvTranspose2x2Fn.apply(x, wt, torch.randn(64, device="cuda"), None)  # 2x2 transposed conv: four one-tap classes
        torch.cuda.synchronize()
        out += [gx2.float(), up.float()]
print("HASH", " ".join(hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16] for t in out))
"""


def test_bf16_modes_merge_the_parity_classes_bit_identically():
    """Round 5: the bf16 operand / storage modes launch the output-parity classes of a stride-2 input gradient and of a 2x2 transposed conv as
    ONE merged gg3s launch (as the fp16-split mode does); LHG_MERGE_CLASSES=0 launches them one by one — the same bits when both run the
    same kernel (LHG_GGB_VARIANT=8: gg3s 64 x 64 with 64-k steps; the bf16 kernels do not share one K order — gg2b walks taps outermost,
    gg3s chunks outermost — so the tuner's free choice per class would differ in rounding)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    for merge in ("1", "0"):
        env = dict(os.environ, LHG_MERGE_CLASSES=merge, LHG_GGB_VARIANT="8", LHG_AUTOTUNE="0", PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-c", _BF16_CLASSES_CHILD], cwd=root, env=env, capture_output=True, text=True)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("HASH")]
        assert out.returncode == 0 and lines, out.stdout[-800:] + out.stderr[-1500:]
        seen[merge] = lines[-1]
    assert seen["1"] == seen["0"], seen


def test_every_tiling_variant_gives_the_same_bits():
    """What the autotuner picks must not show in the results: a 3x3 forward, its input gradient (flipped taps), a stride-2 input
    gradient (merged parity classes, and one launch per class) and a weight gradient are bit-identical across the gather variants —
    gg3s tiles, the 64 x 128 tile, strips — and across the weight-gradient kernels wg2s / wg4s / wg5p: every kernel accumulates per
    32-channel chunk in ascending tap order, two sub-steps of 16, product groups a1 b0, a0 b1, a0 b0.  (One child process per forced
    pair: LHG_GGS_VARIANT / LHG_WG_VARIANT are read once per process.)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = {}
    # 10, 11: twelve-wave strips (single-class launches); 12, 13 (round 5, LHG_GG_EXPERIMENTAL): strips with two taps per barrier (128 x 64 /
    # 64 x 64; on the merged stride-2 launch the same numbers mean one launch per class of variants 2 / 3); 15: swizzled unpadded LDS rows
    for gv, wv in ((2, 21), (0, 18), (9, 22), (5, 25), (8, 10), (12, 13), (13, 21), (15, 18), (10, 21), (11, 18)):
        # LHG_WG6=0: the per-tap weight-gradient kernels this test forces (the tap-fused kernel's variants: tests/test_gpu_wgrad6.py)
        env = dict(os.environ, LHG_AUTOTUNE="0", LHG_GGS_VARIANT=str(gv), LHG_WG_VARIANT=str(wv), LHG_WG6="0", LHG_GG_EXPERIMENTAL="1", PYTHONPATH=root)
        out = subprocess.run([sys.executable, "-c", _VARIANT_CHILD], cwd=root, env=env, capture_output=True, text=True)
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("HASH")]
        assert out.returncode == 0 and lines, out.stdout[-800:] + out.stderr[-800:]
        seen[(gv, wv)] = lines[-1]
    assert len(set(seen.values())) == 1, seen
