"""Do a gather-GEMM and a weight-gradient GEMM finish sooner side by side than one after the other?  Pairs of the step's layers, each
alone (events on its own stream), back to back on one stream, and concurrently on two streams.
    python tools/dbg/mix_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from learned_hologram_gan_amd import hip_ops as ops

dev = "cuda:0"
ops.set_conv_precision("fp32_split_f16")
ops.SIDE_WGRAD = False
def rnd(*s): return (torch.rand(*s, device=dev) * 2 - 1)
def wall(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
side = torch.cuda.Stream()
for (Ci, Co, HW, Cw_i, Cw_o, HWw) in [(256, 512, 96, 256, 512, 96), (128, 128, 192, 128, 128, 192), (64, 64, 384, 64, 64, 384), (512, 256, 96, 128, 128, 192), (64, 64, 384, 256, 512, 96)]:
    x = rnd(4, HW, HW, Ci); w = rnd(Co, Ci, 3, 3) * 0.05
    xw = rnd(4, HWw, HWw, Cw_i); gyw = rnd(4, HWw, HWw, Cw_o)
    with torch.no_grad():
        f_gg = lambda: ops.conv2d_forward_raw(x, w, None, 1)
        f_wg = lambda: ops.conv2d_weight_grad_raw(xw, gyw, (Cw_o, Cw_i, 3, 3), 1)
        for _ in range(3): f_gg(); f_wg()
        a, b = wall(f_gg), wall(f_wg)
        def serial(): f_gg(); f_wg()
        def both():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side): f_wg()
            f_gg()
            torch.cuda.current_stream().wait_stream(side)
        s, c = wall(serial), wall(both)
    print(f"gg {Ci}>{Co}@{HW} {a:7.1f} us | wg {Cw_i}>{Cw_o}@{HWw} {b:7.1f} us | one stream {s:7.1f} | two streams {c:7.1f}  ({100 * (s - c) / s:4.1f} % saved)", flush=True)
