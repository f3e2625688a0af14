"""GPU box: where the host time of hip_ops.repack_weights goes (line timers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LHG_BATCHED_REPACK"] = "0"
import gc
import torch
from learned_hologram_gan_amd import hip_ops, native
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

dev = torch.device("cuda", 0)
S = int(os.environ.get("PROBE_SIZE", "96"))
trainer = watermelon(filter_radius_coefficient=0.45, pad_size=80, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, S, S))
trainer.generator.to(dev).train(); trainer.discriminator.to(dev).train()
trainer.configure(phs_gradient_loss_weight=1, perceptual_loss_weight=0, pixel_loss_weight=1, TV_loss_weight=1e-3, discriminator_loss_weight=1e-1,
                  lr_G=1e-3, lr_D=1e-3, discriminator_train_ratio=1, discriminator_lambda=10)
g = torch.Generator().manual_seed(1)
data = tuple(torch.rand((4, c, S, S), generator=g).to(dev) for c in (4, 3, 3))
for _ in range(4):
    trainer.train_step(*data)
torch.cuda.synchronize()
params = [p for p in trainer.generator.parameters()]
T = time.perf_counter
for rep in range(4):
    for p in params:
        hip_ops.bump_version(p)
    t = [T()]
    mode, items, fresh = hip_ops._mode(), [], []
    for w in params:
        cache = w.__dict__.get("_lhg_packed")
        if not cache or w.dim() != 4 or not w.is_cuda or not w.is_contiguous():
            continue
        stamp = (w.data_ptr(), w._version, tuple(w.shape))
        D0, D1, KH, KW = w.shape
        for key, (old, out) in list(cache.items()):
            rows_from_d0, k_pad, m = key
            if m != mode or old == stamp or old[0] != stamp[0] or old[2] != stamp[2]:
                continue
            rows_pad = hip_ops.pad_to(D0 if rows_from_d0 else D1, 64)
            items.append(native.PackItem(w.data_ptr(), out.data_ptr(), D0, D1, KH, KW, int(rows_from_d0), rows_pad, k_pad))
            fresh.append((cache, key, stamp, out))
    t.append(T())
    arr = (native.PackItem * len(items))(*items)
    t.append(T())
    native.call("lhg_pack_weights", arr, len(items), native.stream_ptr())
    t.append(T())
    for cache, key, stamp, out in fresh:
        cache[key] = (stamp, out)
    t.append(T())
    torch.cuda.synchronize()
    t.append(T())
    print(len(params), len(items), "scan %.3f  array %.3f  call %.3f  stamps %.3f  sync %.3f ms" % tuple((b - a) * 1e3 for a, b in zip(t, t[1:])))
