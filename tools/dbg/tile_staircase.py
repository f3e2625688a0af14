"""Gather-GEMM time against the number of output tiles (rows of the image varied): a staircase says grid quantisation, a line says none.
    python tools/dbg/tile_staircase.py Ci Co W"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from learned_hologram_gan_amd import hip_ops as ops

Ci, Co, W = (int(a) for a in sys.argv[1:4])
ops.set_conv_precision("fp32_split_f16")
dev = "cuda:0"
w = (torch.rand(Co, Ci, 3, 3) * 2 - 1).to(dev) * 0.05
def t(fn, reps=8):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for H in list(range(16, 4 * W + 1, max(4, W // 12))):
    x = (torch.rand(1, H, W, Ci) * 2 - 1).to(dev)
    with torch.no_grad():
        us = t(lambda: ops.conv2d_forward_raw(x, w, None, 1))
    fl = 2.0 * H * W * Ci * Co * 9
    print(f"H={H:4d} M={H*W:7d} tiles128={-(-H*W//128)*(-(-Co//128)):5d} {us:8.1f} us {fl/us/1e6:7.1f} TFLOP/s", flush=True)
