"""Loss / PSNR / SSIM over a few hundred training steps on smooth synthetic frames (does it learn?).
Usage (GPU box): python tests/diagnostics/learn_curve.py [rows] [gan|nogan] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon, watermelon_without_GAN
from learned_hologram_gan_amd.poh_ops import psnr_ssim
from oracle import seeded
dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cls = watermelon if (len(sys.argv) > 2 and sys.argv[2] == "gan") else watermelon_without_GAN
torch.manual_seed(0)
W = cls(filter_radius_coefficient=0.45, pad_size=R // 2, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, R, R))
W.generator.to(dev).train(); W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 1e-1, 1e-3, 1e-3, 1 if cls is watermelon else 0, 10)
batches = [tuple(t.to(dev) for t in seeded.smooth_batch(4, R, R, seed=200 + i)) for i in range(4)]
for it in range(int(sys.argv[3]) if len(sys.argv) > 3 else 120):
    rgbd, tamp, tphs = batches[it % 4]
    out = W.train_step(rgbd, tamp, tphs)
    if it % 20 == 0 or it == 119:
        m = psnr_ssim(out["hat_amps"], out["target_amps"]).tolist()
        print(f"step {it:4d} G_loss {out['G_loss'].item():.4f} D_loss {float(out['D_loss']):.3f} PSNR {m[0]:.2f} dB SSIM {m[1]:.4f}", flush=True)
