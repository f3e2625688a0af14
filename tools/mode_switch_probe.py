"""GPU box: does the bf16-storage step cost the same after an fp32 phase in the same process?  usage: mode_switch_probe.py [fp32_first=1]"""
import sys, time, torch
sys.path.insert(0, ".")
from learned_hologram_gan_amd import hip_ops, native
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

dev = torch.device("cuda", 0)
def build():
    T = watermelon(filter_radius_coefficient=0.45, pad_size=320, distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, 384, 384))
    T.generator.to(dev).train(); T.discriminator.to(dev).train()
    T.configure(phs_gradient_loss_weight=1, perceptual_loss_weight=0, pixel_loss_weight=1, TV_loss_weight=1e-3, discriminator_loss_weight=1e-1,
                lr_G=1e-3, lr_D=1e-3, discriminator_train_ratio=1, discriminator_lambda=10)
    return T
g = torch.Generator().manual_seed(1)
rgbd = torch.rand((4, 4, 384, 384), generator=g).to(dev); tamp = torch.rand((4, 3, 384, 384), generator=g).to(dev); tphs = torch.rand((4, 3, 384, 384), generator=g).to(dev)
def timeit(W, n=6):
    for _ in range(3): W.train_step(rgbd, tamp, tphs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): W.train_step(rgbd, tamp, tphs)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
if int(sys.argv[1]) if len(sys.argv) > 1 else 1:
    W = build(); print("fp32 phase %.2f ms" % timeit(W)); del W; torch.cuda.empty_cache()
hip_ops.set_activation_storage("bf16")
W = build()
for i in range(4): print("bf16 storage, block %d: %.2f ms" % (i, timeit(W)))
with native.kernel_profile() as prof:
    for _ in range(3): W.train_step(rgbd, tamp, tphs)
    torch.cuda.synchronize()
r = prof.result
print({k: (round(v["total_ms"] / 3, 3) if isinstance(v, dict) and "total_ms" in v else v) for k, v in r.items()} if isinstance(r, dict) else r)
print("mem GB", torch.cuda.memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): W.train_step(rgbd, tamp, tphs)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.2f ms/step, +sync tail %.2f ms" % ((t1 - t0) / 4 * 1e3, (t2 - t1) * 1e3))
import cProfile, pstats, gc
print("gc counts", gc.get_count(), "objects", len(gc.get_objects()))
pr = cProfile.Profile(); pr.enable()
for _ in range(4): W.train_step(rgbd, tamp, tphs)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
