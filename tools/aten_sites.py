"""Which Python call sites launch the ATen kernels of one train step (torch.profiler, one step): op name, count, CUDA time, innermost repo frame."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon
dev = torch.device("cuda", 0)
W = watermelon(filter_radius_coefficient=0.45, pad_size=320, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, 384, 384))
W.generator.to(dev).train(); W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 1e-1, 1e-3, 1e-3, 1, 10)
g = torch.Generator().manual_seed(1)
rgbd, tamp, tphs = (torch.rand((4, c, 384, 384), generator=g).to(dev) for c in (4, 3, 3))
for _ in range(3): W.train_step(rgbd, tamp, tphs)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    W.train_step(rgbd, tamp, tphs)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    st = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if not ev.key.startswith("aten::") or st <= 0:
        continue
    frames = [f for f in (ev.stack or []) if "learned_hologram_gan_amd" in f or "bench.py" in f]
    where = " < ".join(f.split("learned_hologram_gan_amd/")[-1].strip() for f in frames[:3])
    rows.append((ev.count, st, ev.key, str(ev.input_shapes)[:70], where[:200]))
# stacks of the LARGE accumulations (autograd adds over activation-sized tensors): per-event, since key_averages drops the frames here
seen = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::add_", "aten::add", "aten::copy_", "aten::fill_", "aten::zero_") and ev.input_shapes and ev.input_shapes[0] and len(ev.input_shapes[0]) == 4:
        n = 1
        for d in ev.input_shapes[0]:
            n *= d
        if n >= 1 << 20:
            frames = [f for f in (ev.stack or []) if "learned_hologram_gan_amd" in f or "autograd" in f][:4]
            seen[(ev.name, tuple(ev.input_shapes[0]), " < ".join(f.split("/")[-1].strip() for f in frames))] += 1
for (name, shp, where), n in seen.most_common(30):
    print(f"LARGE {n:3d} {name:12s} {shp} {where}")
print("device-launching ATen ops of one train step, by call site:", sum(r[0] for r in rows), "launching ops")
for n, st, name, shp, where in sorted(rows, reverse=True)[:80]:
    print(f"{n:4d} {st:9.1f} us  {name:22s} {shp:70s} {where}")
