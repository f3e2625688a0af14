"""Does a streaming kernel run slower right after matrix-pipe work (power state)?  bn_apply on 151 MB timed alone, and directly behind a
burst of gather-GEMMs on the same stream."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from learned_hologram_gan_amd import hip_ops as ops
from learned_hologram_gan_amd.hip_ops import call, ptr, stream_ptr
dev = "cuda:0"
ops.set_conv_precision("fp32_split_f16")
N, HW, C = 4, 384, 64
pixels = N * HW * HW
x = torch.rand(N, HW, HW, C, device=dev) * 2 - 1
y = torch.empty_like(x)
stats = torch.empty(2 * C, device=dev); gamma = torch.rand(C, device=dev) + 0.5; beta = torch.rand(C, device=dev) - 0.5
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev); ws = torch.empty(8192 * C, device=dev); amax = torch.zeros(4, device=dev)
call("lhg_bn_stats", ptr(x), pixels, C, C, ptr(stats), ptr(rm), ptr(rv), 0.1, 1e-5, ptr(ws), stream_ptr())
f_apply = lambda: call("lhg_bn_apply", ptr(x), C, pixels, C, ptr(stats), ptr(gamma), ptr(beta), None, 0, 1, 0.0, ptr(y), C, ptr(amax), stream_ptr())
xa = torch.rand(4, 96, 96, 256, device=dev) * 2 - 1
w = (torch.rand(512, 256, 3, 3, device=dev) * 2 - 1) * 0.05
def gemm(n):
    with torch.no_grad():
        for _ in range(n): ops.conv2d_forward_raw(xa, w, None, 1)
gemm(3); f_apply(); torch.cuda.synchronize()
def timed(pre):
    ts = []
    for _ in range(12):
        pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f_apply(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2], ts[0], ts[-1]
print("alone (idle before):      median %.1f us (min %.1f max %.1f)" % timed(lambda: torch.cuda.synchronize()))
print("behind 1 gather-GEMM:     median %.1f us (min %.1f max %.1f)" % timed(lambda: gemm(1)))
print("behind 10 gather-GEMMs:   median %.1f us (min %.1f max %.1f)" % timed(lambda: gemm(10)))
print("behind 40 gather-GEMMs:   median %.1f us (min %.1f max %.1f)" % timed(lambda: gemm(40)))
z = torch.empty(64 * 1024 * 1024, device=dev)
print("behind a 256 MB fill:     median %.1f us (min %.1f max %.1f)" % timed(lambda: z.fill_(1.0)))
