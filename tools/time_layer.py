"""Time one conv forward (args: Ci Co HW k stride precision reps).  Used with the LHG_* ablation switches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd import hip_ops as ops
Ci, Co, HW, k, s = map(int, sys.argv[1:6]); ops.set_conv_precision(sys.argv[6]); reps = int(sys.argv[7])
x = torch.randn(4, HW, HW, Ci, device="cuda"); w = torch.randn(Co, Ci, k, k, device="cuda") * 0.05
with torch.no_grad():
    for _ in range(3): ops.conv2d_forward_raw(x, w, None, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
    for _ in range(reps): ops.conv2d_forward_raw(x, w, None, s)
    e1.record(); torch.cuda.synchronize()
print("%.1f us" % (e0.elapsed_time(e1) * 1e3 / reps))
