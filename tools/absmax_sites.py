"""GPU box: which call sites still measure max|x| with a separate lhg_absmax launch in one train step (bytes and count per site)."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd import hip_ops
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon
dev = torch.device("cuda", 0)
W = watermelon(filter_radius_coefficient=0.45, pad_size=320, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, 384, 384))
W.generator.to(dev).train(); W.discriminator.to(dev).train()
W.configure(phs_gradient_loss_weight=1, perceptual_loss_weight=0, pixel_loss_weight=1, TV_loss_weight=1e-3, discriminator_loss_weight=1e-1,
            lr_G=1e-3, lr_D=1e-3, discriminator_train_ratio=1, discriminator_lambda=10)
g = torch.Generator().manual_seed(1)
rgbd = torch.rand((4, 4, 384, 384), generator=g).to(dev); tamp = torch.rand((4, 3, 384, 384), generator=g).to(dev); tphs = torch.rand((4, 3, 384, 384), generator=g).to(dev)
for _ in range(2): W.train_step(rgbd, tamp, tphs)
sites = collections.defaultdict(lambda: [0, 0])
orig = hip_ops._amax_slot
real_call = hip_ops.call
def call(name, *a):
    if name == "lhg_absmax":
        st = traceback.extract_stack(limit=8)
        key = " < ".join(f"{f.name}:{f.lineno}" for f in reversed(st[:-2]) if "hip_ops" in f.filename or "neural" in f.filename or "discriminator" in f.filename)[:150]
        sites[key][0] += 1; sites[key][1] += a[1] * a[2] * 4
    return real_call(name, *a)
hip_ops.call = call
W.train_step(rgbd, tamp, tphs)
torch.cuda.synchronize()
tot = sum(v[1] for v in sites.values())
print("separate absmax launches:", sum(v[0] for v in sites.values()), "bytes read: %.2f GB" % (tot / 1e9))
for k, v in sorted(sites.items(), key=lambda kv: -kv[1][1]): print("%4d  %7.1f MB  %s" % (v[0], v[1] / 1e6, k))
