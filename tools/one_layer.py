"""Run ONE conv layer shape repeatedly (for rocprofv3 --pmc).  args: Ci Co HW k stride mode(fwd|dgrad|wgrad) reps [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd import hip_ops as ops
Ci, Co, HW, k, s = map(int, sys.argv[1:6]); mode = sys.argv[6]; reps = int(sys.argv[7]); B = 4
if len(sys.argv) > 8: ops.set_conv_precision(sys.argv[8])  # fp32 | fp32_split | fp32_split2 | bf16
Cip = ops.pad_to(Ci, 32)
x = torch.randn(B, HW, HW, Cip, device="cuda"); w = torch.randn(Co, Ci, k, k, device="cuda") * 0.05
Ho = (HW + 2 * (k // 2) - k) // s + 1
gy = torch.randn(B, Ho, Ho, Co, device="cuda")
with torch.no_grad():
    for _ in range(reps):
        if mode == "fwd": ops.conv2d_forward_raw(x, w, None, s)
        elif mode == "dgrad": ops.Conv2dInputGradFn.apply(gy, w, s, HW, HW, Cip)
        else: ops.Conv2dWeightGradFn.apply(x, gy, w.shape, s)
torch.cuda.synchronize()
