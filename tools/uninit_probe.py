"""GPU box: does any kernel's result depend on memory nobody wrote?  torch.empty / empty_like (every workspace, output and slab the
ops allocate) are patched to return pattern-filled tensors; a small train step must give bit-identical results for every pattern."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd import hip_ops
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

PATTERN = [None]
real_empty, real_empty_like = torch.empty, torch.empty_like


def fill(t):
    p = PATTERN[0]
    if p is not None and t.is_cuda and t.numel():
        if t.is_floating_point() or t.is_complex():
            t.fill_(p)
        else:
            t.fill_(0x5a if t.dtype == torch.uint8 else 12345)
    return t


torch.empty = lambda *a, **k: fill(real_empty(*a, **k))
torch.empty_like = lambda *a, **k: fill(real_empty_like(*a, **k))

dev = "cuda:0"
rows = cols = 64
torch.manual_seed(5)
stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
W = watermelon(filter_radius_coefficient=0.45, pad_size=rows // 2, distance_stack=stack, input_shape=(1, 4, rows, cols))
W.generator.to(dev).train(); W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10)
g = torch.Generator().manual_seed(91)
x = (torch.rand((2, 4, rows, cols), generator=g).to(dev), torch.rand((2, 3, rows, cols), generator=g).to(dev), torch.rand((2, 3, rows, cols), generator=g).to(dev),
     torch.tensor([5, 2]), [torch.tensor([0.3, 0.8]).view(2, 1, 1, 1).to(dev)])
grabbed = []
for opt in (W._opt_D, W._opt_G):
    def step(grad_scale=1.0, opt=opt):
        hip_ops.join_side_stream(); torch.cuda.synchronize(); grabbed.append(opt.flat.grad.detach().clone())
    opt.step = step


def run(pat):
    PATTERN[0] = pat
    out = W.train_step(*x)
    torch.cuda.synchronize()
    PATTERN[0] = None
    return {"POH": out["POH"].clone(), "hat_amps": out["hat_amps"].clone(), "target_amps": out["target_amps"].clone(), "G_loss": out["G_loss"].clone(),
            "D_loss": out["D_loss"].clone(), "gradD": grabbed[-2], "gradG": grabbed[-1]}


run(None); run(None)
base = run(None)
bad = 0
for pat in (0.0, float("nan"), 1e30, -3.0e-3, float("inf")):
    r = run(pat)
    diffs = [k for k in base if not torch.equal(base[k], r[k])]
    print("empty() filled with %-6s: %s" % (pat, "identical" if not diffs else "DIFFERENT: " + ", ".join(
        "%s (%d elements, max |d| %.3g)" % (k, int((base[k] != r[k]).sum()), float((base[k] - r[k]).abs().nan_to_num(nan=1e30).max())) for k in diffs)), flush=True)
    bad += bool(diffs)
sys.exit(1 if bad else 0)
