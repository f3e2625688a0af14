"""Where does the HOST spend the launch loop of a training step?  cProfile over K steps of the bench workload (no device sync inside).
Usage (GPU box): python tools/host_profile.py [steps] [top N] [sort: tottime|cumtime]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
TOP = int(sys.argv[2]) if len(sys.argv) > 2 else 45
SORT = sys.argv[3] if len(sys.argv) > 3 else "tottime"
R = 384
dev = torch.device("cuda", 0)
W = watermelon(filter_radius_coefficient=0.45, pad_size=320, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, R, R))
W.generator.to(dev).train()
W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 1e-1, 1e-3, 1e-3, 1, 10)
g = torch.Generator().manual_seed(0)
rgbd, tamp, tphs = (torch.rand((4, c, R, R), generator=g).to(dev) for c in (4, 3, 3))
for _ in range(4):
    W.train_step(rgbd, tamp, tphs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    W.train_step(rgbd, tamp, tphs)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"unprofiled host launch loop {1e3 * (t1 - t0) / K:.1f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    W.train_step(rgbd, tamp, tphs)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats(SORT)
print(f"per-step figures = totals / {K}")
st.print_stats(TOP)
