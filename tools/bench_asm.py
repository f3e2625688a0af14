"""Time the fused angular-spectrum operator at the benchmark geometry (384^2, pad 320 -> 1024^2, B=4, 20-plane stack).
Usage (GPU box): python tools/bench_asm.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd.angular_spectrum_method import (bandLimitedAngularSpectrumMethod_for_multiple_distances as Mu,
                                                              bandLimitedAngularSpectrumMethod_for_single_fixed_distance as Fx)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
DEV, B, R0 = "cuda:0", 4, 384
WL = torch.tensor([638e-9, 520e-9, 450e-9])
stack = torch.linspace(-4e-4, 0.0, 21)[:-1]
fx = Fx(R0, R0, 320, 0.45, 3.74e-6, WL, False, True, torch.tensor([1e-3]))
mu = Mu(R0, R0, stack, 320, 0.45, 3.74e-6, WL, False, True)
g = torch.Generator().manual_seed(0)
amp, phs, poh = (torch.rand((B, 3, R0, R0), generator=g).to(DEV) for _ in range(3))
idx = torch.tensor([3, 17, 9, 12])


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def recon_fb():
    q = poh.clone().requires_grad_(True)
    ha, hp, ta, tp = mu.reconstruct_planes(fx, q, amp, phs, idx)
    (ha.sum() + hp.sum()).backward()


with torch.no_grad():
    t_back = timeit(lambda: fx.propagate_AP2C_backward(amp, phs))
    t_rec = timeit(lambda: mu.reconstruct_planes(fx, poh, amp, phs, idx))
t_fb = timeit(recon_fb)
planes_back, planes_rec = 3 * B, 2 * 3 * B
mb = lambda planes: planes * 2 * 32 * 1024 * 1024 / 1e6  # 32*R*C bytes per 2-D transform, fft2 + ifft2 per plane
print(f"A5 back-propagation   (12 planes): {t_back:8.1f} us  = {mb(planes_back)/t_back:7.2f} TB/s of SURVEY-algorithmic bytes ({mb(planes_back):.0f} MB)")
print(f"A8+A9 reconstruction  (24 planes): {t_rec:8.1f} us  = {mb(planes_rec)/t_rec:7.2f} TB/s of SURVEY-algorithmic bytes ({mb(planes_rec):.0f} MB)")
print(f"A8+A9 forward+backward           : {t_fb:8.1f} us")
