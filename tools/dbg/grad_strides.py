"""Which backward functions hand autograd a gradient of a given shape, and with what strides (a strided gradient makes autograd's accumulation slow)."""
import sys, os, inspect
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from learned_hologram_gan_amd import hip_ops
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon
SHAPE = tuple(int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (4, 96, 96, 256)
log = []
for name, cls in inspect.getmembers(hip_ops, inspect.isclass):
    if issubclass(cls, torch.autograd.Function) and "backward" in cls.__dict__:
        orig = cls.__dict__["backward"].__func__
        def make(orig, name):
            def wrapped(ctx, *gs):
                ins = [(tuple(g.shape), g.stride(), g.is_contiguous()) for g in gs if isinstance(g, torch.Tensor) and tuple(g.shape) == SHAPE]
                out = orig(ctx, *gs)
                outs = out if isinstance(out, tuple) else (out,)
                o = [(tuple(g.shape), g.stride(), g.is_contiguous()) for g in outs if isinstance(g, torch.Tensor) and tuple(g.shape) == SHAPE]
                if ins or o:
                    log.append((name, "in", ins, "out", o))
                return out
            return staticmethod(wrapped)
        setattr(cls, "backward", make(orig, name))
dev = torch.device("cuda", 0)
W = watermelon(filter_radius_coefficient=0.45, pad_size=320, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, 384, 384))
W.generator.to(dev).train(); W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 1e-1, 1e-3, 1e-3, 1, 10)
g = torch.Generator().manual_seed(1)
rgbd, tamp, tphs = (torch.rand((4, c, 384, 384), generator=g).to(dev) for c in (4, 3, 3))
W.train_step(rgbd, tamp, tphs)
log.clear()
W.train_step(rgbd, tamp, tphs)
torch.cuda.synchronize()
for r in log: print(r)
