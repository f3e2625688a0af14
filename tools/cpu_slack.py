"""How far ahead of the GPU does the host run?  (launch-loop time vs completed time of K training steps)
Usage (GPU box): python tools/cpu_slack.py [steps] [f32|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd import hip_ops
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R = int(sys.argv[3]) if len(sys.argv) > 3 else 384
if len(sys.argv) > 2 and sys.argv[2] == "bf16":
    hip_ops.set_conv_precision("bf16")
dev = torch.device("cuda", 0)
W = watermelon(filter_radius_coefficient=0.45, pad_size=(320 if R == 384 else R // 2), distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, R, R))
W.generator.to(dev).train(); W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 1e-1, 1e-3, 1e-3, 1, 10)
g = torch.Generator().manual_seed(0)
rgbd, tamp, tphs = (torch.rand((4, c, R, R), generator=g).to(dev) for c in (4, 3, 3))  # noqa
for _ in range(4): W.train_step(rgbd, tamp, tphs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K): W.train_step(rgbd, tamp, tphs)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host launch loop {1e3*(t1-t0)/K:.1f} ms/step, completed {1e3*(t2-t0)/K:.1f} ms/step")
