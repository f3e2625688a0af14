"""Which weight-gradient operands still pay a per-channel-maxima pass of their own (hip_ops.operand_chanmax) in one train step?
Usage (GPU box): python tools/chanmax_sites.py"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from learned_hologram_gan_amd import hip_ops
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

dev = torch.device("cuda", 0)
R = 384
W = watermelon(filter_radius_coefficient=0.45, pad_size=320, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, R, R))
W.generator.to(dev).train()
W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 1e-1, 1e-3, 1e-3, 1, 10)
g = torch.Generator().manual_seed(0)
rgbd, tamp, tphs = (torch.rand((4, c, R, R), generator=g).to(dev) for c in (4, 3, 3))
W.train_step(rgbd, tamp, tphs)
hip_ops.CHANMAX_STATS.update(fused=0, pass_bytes=0, **{"pass": 0})
hip_ops.CHANMAX_PASS_LOG = []
W.train_step(rgbd, tamp, tphs)
torch.cuda.synchronize()
print("per step:", hip_ops.CHANMAX_STATS)
agg = collections.Counter()
mb = collections.Counter()
for shape, stack, producer in hip_ops.CHANMAX_PASS_LOG:
    key = (shape, producer + "  " + " < ".join(reversed(stack[-3:])))
    agg[key] += 1
    mb[key] += shape[0] * shape[1] * shape[2] * shape[4] * 4 / 1e6
for key, n in sorted(agg.items(), key=lambda kv: -mb[kv[0]]):
    print(f"{n:3d} x {key[0]}  {mb[key]:8.1f} MB  {key[1]}")
