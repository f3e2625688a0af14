"""lhg_bn_stats_finish alone on random partial rows: microseconds per (rows, C) — the row counts the step's conv epilogues leave.
    LHG_BN_FINISH_FORM=<n> python tools/time_bn_finish.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from learned_hologram_gan_amd.native import call, ptr, stream_ptr

dev = "cuda:0"
def t(fn, reps=50):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = []
for rows, C in [(144, 1024), (576, 256), (576, 512), (2304, 128), (2316, 128), (9216, 64), (9264, 64), (18500, 64)]:
    part = torch.randn(rows * 2 * C, device=dev)
    stats, rm, rv, b = torch.empty(2 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.randn(C, device=dev)
    big = torch.empty(64 << 20, device=dev)  # flush: the rows come from another kernel's writes in the step
    def run():
        call("lhg_bn_stats_finish", ptr(part), rows, ptr(b), rows * 64, C, ptr(stats), ptr(rm), ptr(rv), 0.1, 1e-5, stream_ptr())
    out.append(f"{rows}x{C}: {t(run):5.1f}")
print(os.environ.get("LHG_BN_FINISH_FORM", "0"), " | ".join(out))
