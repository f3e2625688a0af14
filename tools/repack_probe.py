"""GPU box: host-side cost of hip_ops.repack_weights inside the train step and how far the host runs ahead of the GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from learned_hologram_gan_amd import hip_ops
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

dev = torch.device("cuda", 0)
S = int(os.environ.get("PROBE_SIZE", "384"))
trainer = watermelon(filter_radius_coefficient=0.45, pad_size=int(os.environ.get("PROBE_PAD", "320")), distance_stack=torch.linspace(-4e-4, 0.0, 21)[:-1], input_shape=(1, 4, S, S))
trainer.generator.to(dev).train(); trainer.discriminator.to(dev).train()
trainer.configure(phs_gradient_loss_weight=1, perceptual_loss_weight=0, pixel_loss_weight=1, TV_loss_weight=1e-3, discriminator_loss_weight=1e-1,
                  lr_G=1e-3, lr_D=1e-3, discriminator_train_ratio=1, discriminator_lambda=10)
g = torch.Generator().manual_seed(1)
data = tuple(torch.rand((4, c, S, S), generator=g).to(dev) for c in (4, 3, 3))
for _ in range(6):
    trainer.train_step(*data)
torch.cuda.synchronize()
spent = [0.0, 0]
real = hip_ops.repack_weights
def timed(params):
    t0 = time.perf_counter()
    n = real(params)
    spent[0] += time.perf_counter() - t0
    spent[1] += n
    return n
hip_ops.repack_weights = timed
in_call = [0.0]
real_call = hip_ops.call
def call(name, *a):
    if name != "lhg_pack_weights":
        return real_call(name, *a)
    t0 = time.perf_counter()
    r = real_call(name, *a)
    in_call[0] += time.perf_counter() - t0
    return r
hip_ops.call = call
N = 10
if os.environ.get("PROBE_GC", "1") == "0":
    import gc
    gc.collect(); gc.disable()
t0 = time.perf_counter()
for _ in range(N):
    trainer.train_step(*data)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("inside lhg_pack_weights: %.2f ms per step" % (in_call[0] / N * 1e3))
print("per step: host enqueue %.2f ms, wall %.2f ms, repack_weights host %.2f ms for %d panels" % (t_host / N * 1e3, t_all / N * 1e3, spent[0] / N * 1e3, spent[1] / N))
