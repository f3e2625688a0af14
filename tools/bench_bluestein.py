"""GPU box: the fused HIP operator on Bluestein extents vs the torch.fft (rocFFT) route of the same module.  usage: bench_bluestein.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd.angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_single_fixed_distance as Fx
WL = torch.tensor([638e-9, 520e-9, 450e-9])
for (r0, c0, pad, B) in ((192, 192, 320, 4), (2160, 3840, 320, 1)):
    fx = Fx(r0, c0, pad, 0.45, 3.74e-6, WL, False, True, torch.tensor([1e-3]))
    a = torch.rand((B, 3, r0, c0), device="cuda") + 0.1; p = torch.rand((B, 3, r0, c0), device="cuda") * 6
    def t(fn, n=5):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    hip = t(lambda: fx.propagate_AP2C_backward(a, p))
    from learned_hologram_gan_amd import asm_ops
    sup = asm_ops.Geometry.supported
    asm_ops.Geometry.supported = lambda self: False  # force the torch.fft route
    try:
        roc = t(lambda: fx.propagate_AP2C_backward(a, p))
    finally:
        asm_ops.Geometry.supported = sup
    print(f"{r0}x{c0} pad {pad} -> {fx.samplingRowNum}x{fx.samplingColNum}, {3*B} planes: fused HIP operator {hip:.2f} ms, torch.fft route {roc:.2f} ms")
