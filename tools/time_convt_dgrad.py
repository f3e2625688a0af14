"""Time the input gradient of ConvTranspose2d(Ci, Co, 2, 2) (a 4-tap stride-2 gather-GEMM).  args: Ci Co H(out of the convT = 2x in) reps [ld of gy]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd import hip_ops as ops
from learned_hologram_gan_amd.native import call, ptr, stream_ptr
Ci, Co, H2, reps = map(int, sys.argv[1:5])
ld = int(sys.argv[5]) if len(sys.argv) > 5 else Co
gy = torch.randn(4, H2, H2, ld, device="cuda")[..., :Co]; w = torch.randn(Ci, Co, 2, 2, device="cuda") * 0.05
wp = ops.pack_weight(w, True)
gx = ops.new_nhwc(4, H2 // 2, H2 // 2, Ci, gy.device)
pg, N, _, _, Cg, ldg = ops.nhwc(gy); pgx, _, _, _, _, ldgx = ops.nhwc(gx)
am = ops.operand_absmax(gy)
def run(): call("lhg_conv_transpose2x2_backward_input", pg, N, H2 // 2, H2 // 2, Cg, ldg, ptr(wp), wp.shape[1], pgx, Ci, ldgx, ptr(am), stream_ptr())
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True); e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
print("%.1f us  %.1f TFLOP/s" % (us, 2.0 * 4 * (H2 // 2) ** 2 * 4 * Ci * Co / us / 1e6))
