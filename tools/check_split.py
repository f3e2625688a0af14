"""Accuracy of the split (fp32-faithful) conv GEMMs against a float64 convolution, next to the exact fp32 MFMA kernels.
Usage (GPU box): python tools/check_split.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from learned_hologram_gan_amd import hip_ops as ops
DEV = "cuda:0"
torch.manual_seed(0)
CASES = [(64, 64, 48, 3, 1, 2), (128, 256, 24, 3, 1, 2), (512, 512, 12, 3, 1, 2), (64, 128, 24, 1, 1, 2), (64, 128, 32, 3, 2, 2), (1024, 1024, 8, 3, 1, 4)]
for Ci, Co, HW, k, s, B in CASES:
    x = torch.randn(B, Ci, HW, HW)
    w = torch.randn(Co, Ci, k, k) * (1.0 / (Ci * k * k) ** 0.5)
    Ho = (HW + 2 * (k // 2) - k) // s + 1
    gy = torch.randn(B, Co, Ho, Ho)
    xd, wd, gd = x.double().requires_grad_(True), w.double().requires_grad_(True), gy.double()
    yd = F.conv2d(xd, wd, None, s, k // 2)
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), gd)
    xh = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    gh = gy.permute(0, 2, 3, 1).contiguous().to(DEV)
    wg = w.to(DEV)
    line = f"Ci {Ci:4d} Co {Co:4d} HW {HW:3d} k{k} s{s}:"
    for mode in ("fp32", "fp32_split", "fp32_split2", "bf16"):
        ops.set_conv_precision(mode)
        with torch.no_grad():
            y = ops.conv2d_forward_raw(xh, wg, None, s).permute(0, 3, 1, 2).cpu().double()
            gx = ops.Conv2dInputGradFn.apply(gh, wg, s, HW, HW, Ci).permute(0, 3, 1, 2).cpu().double()
            gw = ops.Conv2dWeightGradFn.apply(xh, gh, w.shape, s).cpu().double()
        e = [((a - b).abs().max() / b.abs().max()).item() for a, b in ((y, yd.detach()), (gx, gxd), (gw, gwd))]
        line += f"  {mode}: fwd {e[0]:.1e} dgrad {e[1]:.1e} wgrad {e[2]:.1e} |"
    print(line)
ops.set_conv_precision("default")
