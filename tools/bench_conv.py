"""Per-layer timing of the conv GEMM kernels at the benchmark shapes (B=4, 384x384).
Usage (GPU box): python tools/bench_conv.py [reps] [fp32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd import hip_ops as ops
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if len(sys.argv) > 2:
    ops.set_conv_precision(sys.argv[2])  # "bf16": bf16 operands in the forward / input-gradient GEMMs
B = 4
# (name, Ci, Co, HW, k, stride)
LAYERS = [("enc1.c1", 4, 64, 384, 3, 1), ("enc1.c2", 64, 64, 384, 3, 1), ("enc1.c3", 4, 64, 384, 1, 1),
          ("enc2.c1", 64, 128, 192, 3, 1), ("enc2.c2", 128, 128, 192, 3, 1), ("enc2.c3", 64, 128, 192, 1, 1),
          ("enc3.c1", 128, 256, 96, 3, 1), ("enc3.c2", 256, 256, 96, 3, 1),
          ("enc4.c1", 256, 512, 48, 3, 1), ("enc4.c2", 512, 512, 48, 3, 1),
          ("bott.c1", 512, 1024, 24, 3, 1), ("bott.c2", 1024, 1024, 24, 3, 1), ("bott.c3", 512, 1024, 24, 1, 1),
          ("dec1.c1", 1024, 512, 48, 3, 1), ("dec2.c1", 512, 256, 96, 3, 1), ("dec3.c1", 256, 128, 192, 3, 1),
          ("dec4.c1", 128, 64, 384, 3, 1), ("dec4.c3", 128, 64, 384, 1, 1), ("head", 64, 6, 384, 1, 1),
          ("D.b1", 3, 32, 384, 3, 1), ("D.b2", 32, 64, 384, 3, 2), ("D.b3", 64, 128, 192, 3, 1), ("D.b4", 128, 256, 192, 3, 2),
          ("D.b5", 256, 512, 96, 3, 1), ("D.b6", 512, 1024, 96, 3, 2), ("D.head", 1024, 1, 48, 3, 1)]

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

tot = [0, 0, 0]
print(f"{'layer':9s} {'Ci':>5s} {'Co':>5s} {'HW':>4s} k s | {'fwd us':>8s} {'TF':>6s} | {'dgrad us':>8s} {'TF':>6s} | {'wgrad us':>8s} {'TF':>6s}")
for name, Ci, Co, HW, k, s in LAYERS:
    Cip = ops.pad_to(Ci, 32)
    x = torch.randn(B, HW, HW, Cip, device=DEV)
    if Cip > Ci: x[..., Ci:] = 0
    w = torch.randn(Co, Ci, k, k, device=DEV) * 0.05
    Ho = (HW + 2 * (k // 2) - k) // s + 1
    gy = torch.randn(B, Ho, Ho, Co, device=DEV)
    flops = 2.0 * B * Ho * Ho * Co * Ci * k * k
    with torch.no_grad():
        tf = timeit(lambda: ops.conv2d_forward_raw(x, w, None, s))
        td = timeit(lambda: ops.Conv2dInputGradFn.apply(gy, w, s, HW, HW, Cip))
        tw = timeit(lambda: ops.Conv2dWeightGradFn.apply(x, gy, w.shape, s))
    tot[0] += tf; tot[1] += td; tot[2] += tw
    print(f"{name:9s} {Ci:5d} {Co:5d} {HW:4d} {k} {s} | {tf*1e3:8.1f} {flops/tf/1e9:6.1f} | {td*1e3:8.1f} {flops/td/1e9:6.1f} | {tw*1e3:8.1f} {flops/tw/1e9:6.1f}")
print("sum ms: fwd %.2f dgrad %.2f wgrad %.2f" % tuple(tot))
