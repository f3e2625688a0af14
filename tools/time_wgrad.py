"""Time the weight gradient of one 3x3 conv layer (args: Ci Co HW reps); LHG_WG_VARIANT / LHG_AUTOTUNE select the kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd import hip_ops as ops
Ci, Co, HW = map(int, sys.argv[1:4]); reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
x = torch.randn(4, HW, HW, Ci, device="cuda"); gy = torch.randn(4, HW, HW, Co, device="cuda")
xa, ga = ops.operand_absmax(x), ops.operand_absmax(gy)
slot = torch.zeros(Co, Ci, 3, 3, device="cuda")
with torch.no_grad():
    for _ in range(3): ops.conv2d_weight_grad_raw(x, gy, (Co, Ci, 3, 3), 1, slot)
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(reps): ops.conv2d_weight_grad_raw(x, gy, (Co, Ci, 3, 3), 1, slot)
    e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print("%.1f us  %.1f TFLOP/s (incl. the slab reduction)" % (us, 2.0 * 4 * HW * HW * 9 * Ci * Co / us / 1e6))
