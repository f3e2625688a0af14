"""GPU box: the tensors that still get a separate lhg_absmax launch in one train step, with shape and how autograd produced them."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
orig = hip_ops.operand_absmax
rows = []
def probe(t):
    known = t.__dict__.get("_lhg_amax")
    hit = known is not None and known[0] == t._version and hip_ops._slot_alive(known[1])
    if not hit and hip_ops._mode() == hip_ops._F16_SPLIT:
        st = traceback.extract_stack(limit=7)
        where = " < ".join(f"{f.name}:{f.lineno}" for f in reversed(st[:-1]) if "hip_ops" in f.filename)[:90]
        rows.append((tuple(t.shape), t.is_contiguous(), type(t.grad_fn).__name__ if t.grad_fn is not None else "-", where))
    return orig(t)
hip_ops.operand_absmax = probe
W.train_step(rgbd, tamp, tphs)
torch.cuda.synchronize()
for r in rows: print(r)
