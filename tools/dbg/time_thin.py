"""Time the thin-convolution kernels of the 384^2 x 4 step (forward, input gradient, weight gradient).  python tools/dbg/time_thin.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from learned_hologram_gan_amd import hip_ops as ops

dev = "cuda:0"
def t(fn, reps=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (Ci, Co, k, HW) in [(4, 64, 3, 384), (64, 6, 3, 384), (3, 32, 3, 384), (1024, 1, 3, 24), (64, 2, 1, 384)]:
    ldx = ops.pad_to(Ci, 32) if Ci > 8 else 4
    x = (torch.rand(4, HW, HW, ldx, device=dev) * 2 - 1).requires_grad_(True)
    w = ((torch.rand(Co, Ci, k, k, device=dev) * 2 - 1) * 0.1).requires_grad_(True)
    if not ops.thin_mode(Ci, Co, k, 1):
        print(Ci, Co, k, "not thin"); continue
    with torch.no_grad():
        fwd = t(lambda: ops.conv2d_forward_raw(x, w, None, 1))
    y = ops.Conv2dFn.apply(x, w, None, 1, None)
    gy = torch.rand_like(y)
    def bw_w():
        w.grad = None
        y.backward(gy, retain_graph=True, inputs=[w])
    def bw_x():
        x.grad = None
        y.backward(gy, retain_graph=True, inputs=[x])
    print(f"{Ci:4d}>{Co:3d} k{k} @{HW}: fwd {fwd:7.1f} us   wgrad (autograd) {t(bw_w):7.1f} us   dgrad (autograd) {t(bw_x):7.1f} us", flush=True)
