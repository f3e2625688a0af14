"""Thin-convolution kernels alone at the step's shapes (fan-out 4 -> 64 and 3 -> 32 at 384^2): microseconds and TB/s of the bytes written.
    python tools/time_thin.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learned_hologram_gan_amd import hip_ops as ops

dev = "cuda:0"
def t(fn, reps=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (N, HW, Ci, Co, k) in [(4, 384, 4, 64, 3), (4, 384, 4, 64, 1), (8, 384, 3, 32, 3), (4, 384, 3, 32, 3), (1, 2160, 4, 64, 3)]:
    W = HW if HW != 2160 else 3840
    x = torch.rand(N, HW, W, 32, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.1
    b = torch.randn(Co, device=dev)
    y = torch.empty(N, HW, W, Co, device=dev)
    us = t(lambda: ops.conv2d_forward_raw(x, w, b, 1, out=ops.OutSlot(y)))
    print(f"fan-out {Ci}->{Co} k{k} {N}x{HW}x{W}: {us:7.1f} us  {y.numel() * 4 / us / 1e6:5.2f} TB/s written", flush=True)
