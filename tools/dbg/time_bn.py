"""BatchNorm kernels alone on the step's shapes: microseconds and TB/s of the bytes each MUST move, next to torch's add on the same bytes.
    python tools/dbg/time_bn.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from learned_hologram_gan_amd import hip_ops as ops
from learned_hologram_gan_amd.hip_ops import call, ptr, stream_ptr

dev = "cuda:0"
def t(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (N, HW, C) in [(4, 384, 64), (8, 384, 32), (4, 192, 128), (4, 96, 256), (4, 48, 512), (4, 24, 1024)]:
    pixels = N * HW * HW
    x = torch.rand(N, HW, HW, C, device=dev) * 2 - 1
    gy = torch.rand_like(x) * 2 - 1
    y, gx = torch.empty_like(x), torch.empty_like(x)
    stats = torch.empty(2 * C, device=dev); gamma = torch.rand(C, device=dev) + 0.5; beta = torch.rand(C, device=dev) - 0.5
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    ws = torch.empty(8192 * C, device=dev)
    gg, gb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    amax = torch.zeros(4, device=dev)
    mb = pixels * C * 4 / 1e6
    f_stats = lambda: call("lhg_bn_stats", ptr(x), pixels, C, C, ptr(stats), ptr(rm), ptr(rv), 0.1, 1e-5, ptr(ws), stream_ptr())
    f_apply = lambda: call("lhg_bn_apply", ptr(x), C, pixels, C, ptr(stats), ptr(gamma), ptr(beta), None, 0, 1, 0.0, ptr(y), C, ptr(amax), stream_ptr())
    f_bwd = lambda: call("lhg_bn_backward", ptr(gy), C, ptr(x), C, None, C, pixels, C, ptr(stats), ptr(gamma), 1, 0.0, ptr(gx), C, None, C,
                         ptr(gg), ptr(gb), 0, ptr(ws), ptr(amax), ptr(beta), stream_ptr())
    f_bwdy = lambda: call("lhg_bn_backward", ptr(gy), C, ptr(x), C, ptr(y), C, pixels, C, ptr(stats), ptr(gamma), 1, 0.0, ptr(gx), C, None, C,
                          ptr(gg), ptr(gb), 0, ptr(ws), ptr(amax), None, stream_ptr())
    f_stats()
    a = t(lambda: torch.add(x, gy, out=gx))
    s, ap, bw, bwy = t(f_stats), t(f_apply), t(f_bwd), t(f_bwdy)
    print(f"{N}x{HW}^2x{C} ({mb:6.1f} MB/tensor): torch add (3 passes) {a:6.1f} us {3 * mb / a:5.2f} TB/s | stats (1) {s:6.1f} us {mb / s:5.2f} | apply (2) {ap:6.1f} us {2 * mb / ap:5.2f} | "
          f"backward, mask from x (2 + 3) {bw:6.1f} us {5 * mb / bw:5.2f} | backward with y (3 + 4) {bwy:6.1f} us {7 * mb / bwy:5.2f}", flush=True)
