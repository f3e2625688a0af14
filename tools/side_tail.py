"""How long does the main stream wait for the weight-gradient stream at the end of each backward pass?  (events around join_side_stream)
Usage (GPU box): python tools/side_tail.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from learned_hologram_gan_amd import hip_ops
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
R = 384
W = watermelon(filter_radius_coefficient=0.45, pad_size=320, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, R, R))
W.generator.to(dev).train()
W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 1e-1, 1e-3, 1e-3, 1, 10)
g = torch.Generator().manual_seed(0)
rgbd, tamp, tphs = (torch.rand((4, c, R, R), generator=g).to(dev) for c in (4, 3, 3))
for _ in range(3):
    W.train_step(rgbd, tamp, tphs)
torch.cuda.synchronize()
records = []
orig = hip_ops.join_side_stream


def join(device=None):
    main = torch.cuda.current_stream(dev)
    side = hip_ops.side_stream(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    b.record(side)
    records.append((a, b))
    orig(device)


hip_ops.join_side_stream = join
t0 = torch.cuda.Event(enable_timing=True)
t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(K):
    W.train_step(rgbd, tamp, tphs)
t1.record()
torch.cuda.synchronize()
waits = [a.elapsed_time(b) for a, b in records]
per = len(waits) // K
print(f"step {t0.elapsed_time(t1) / K:.2f} ms; {per} joins per step; main waits for the side stream (ms, >0 = side finishes later):")
for s in range(K):
    print("   ", [round(w, 2) for w in waits[s * per:(s + 1) * per]])
