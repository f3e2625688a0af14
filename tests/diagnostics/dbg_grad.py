import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import seeded
from learned_hologram_gan_amd.neural_network_components import UNet
sd32 = {k[len("part1.part1."):]: v for k, v in seeded.generator_state_dict().items() if k.startswith("part1.part1.")}
net = UNet(6, 4); net.load_state_dict(sd32); net.to("cuda").train()
rgbd, _, _ = seeded.smooth_batch(2, 32, 32, seed=21)
x = rgbd.to("cuda").requires_grad_(True)
print("leaf", x.is_leaf, x.requires_grad)
y = net(x)
print("y grad_fn", y.grad_fn)
g = y.grad_fn
y.sum().backward()
print("x.grad", None if x.grad is None else x.grad.abs().sum().item())
