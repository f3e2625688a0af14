"""GPU box: does any kernel read memory it (or its producer) never wrote?  A train step is run on clean cached memory, then the caching
allocator's free blocks are filled with a poison value (NaN, 1e30, 1.0) and the SAME step is run again: every output must repeat bit for
bit.  usage: poison_probe.py [rows=64] [pad=32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.manual_seed(122731)  # random-init weights: the probe compares the step with itself
from learned_hologram_gan_amd import hip_ops
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = "cuda:0"
stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
W = watermelon(filter_radius_coefficient=0.45, pad_size=pad, distance_stack=stack, input_shape=(1, 4, rows, rows))
W.generator.to(dev).train(); W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10)
g = torch.Generator().manual_seed(200)
rgbd, tamp, tphs = (torch.rand((2, c, rows, rows), generator=g) for c in (4, 3, 3))
idx = torch.tensor([5, 2]); alphas = [torch.tensor([0.3, 0.8]).view(2, 1, 1, 1).to(dev)]
x = (rgbd.to(dev), tamp.to(dev), tphs.to(dev), idx, alphas)
grabbed = {}
def grab(name, opt):
    def step(grad_scale=1.0):
        hip_ops.join_side_stream(); torch.cuda.synchronize()
        grabbed.setdefault(name, []).append(opt.flat.grad.detach().clone())
    return step
W._opt_G.step = grab("G", W._opt_G); W._opt_D.step = grab("D", W._opt_D)
def names(model, flat, a, b):
    nm = {id(p): n for n, p in model.named_parameters()}; out = []
    for p_, o in zip(flat.params, flat.offsets):
        da, db = a[o:o + p_.numel()], b[o:o + p_.numel()]
        if not torch.equal(da, db): out.append((nm.get(id(p_), "?"), float((da - db).norm() / (db.norm() + 1e-30))))
    return sorted(out, key=lambda t: -t[1])[:8]
def poison(value):
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info()
    n = min(int(free * 0.25), 24 << 30) // 4
    big = torch.full((n,), value, dtype=torch.float32, device=dev); torch.cuda.synchronize(); del big  # stays in the cache, poisoned
for _ in range(2): out0 = W.train_step(*x)
base = {k: v[-1] for k, v in grabbed.items()}
for value in (float("nan"), 1e30, 1.0, -3.0e-5):
    poison(value)
    out = W.train_step(*x)
    bad = False
    for k in ("G", "D"):
        cur = grabbed[k][-1]
        if not torch.equal(cur, base[k]):
            bad = True
            model, flat = (W.generator, W._opt_G.flat) if k == "G" else (W.discriminator, W._opt_D.flat)
            print(f"poison {value}: {k} gradients differ:", names(model, flat, cur, base[k]))
    print(f"poison {value}: {'DIFFERENT' if bad else 'identical'}")
