"""Time the stride-2 input gradient of a 3x3 conv (four output-parity classes of 1 / 2 / 2 / 4 taps).  args: Ci Co HW(of x) reps [batch]
LHG_GGS_VARIANT < 10: one merged launch; 10 + v: one launch per class with variant v (read once per process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from learned_hologram_gan_amd import hip_ops as ops
Ci, Co, HW, reps = map(int, sys.argv[1:5]); B = int(sys.argv[5]) if len(sys.argv) > 5 else 4
ops.set_conv_precision("fp32_split_f16")
gy = torch.randn(B, HW // 2, HW // 2, Co, device="cuda"); w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
with torch.no_grad():
    f = lambda: ops.Conv2dInputGradFn.apply(gy, w, 2, HW, HW, Ci)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print("%.1f us  %.1f TFLOP/s" % (us, 2.0 * B * (HW // 2) ** 2 * 9 * Ci * Co / us / 1e6))
