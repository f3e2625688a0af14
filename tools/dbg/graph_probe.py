"""Where does the host time of a graphed train step go?  replay() alone (idle GPU, back to back), staging alone, with / without the
second stream."""
import os, sys, time, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from learned_hologram_gan_amd import hip_ops
from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon
side = os.environ.get("PROBE_SIDE", "1") == "1"
hip_ops.SIDE_WGRAD = side
if os.environ.get("PROBE_BF16", "0") == "1":
    hip_ops.set_activation_storage("bf16")
dev = "cuda:0"
rows = int(os.environ.get("PROBE_ROWS", "384")); B = 4
stack = torch.linspace(-4e-4, 0.0, 21)[:-1]
W = watermelon(filter_radius_coefficient=0.45, pad_size=320 if rows == 384 else 32, distance_stack=stack, input_shape=(1, 4, rows, rows))
W.generator.to(dev).train(); W.discriminator.to(dev).train()
W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10)
g = torch.Generator().manual_seed(1)
x = [torch.rand((B, c, rows, rows), generator=g).to(dev) for c in (4, 3, 3)]
W.use_graph = True
for _ in range(3): W.train_step(*x)
torch.cuda.synchronize()
G = W._graphed
def t(fn, n=5, sync_each=False):
    torch.cuda.synchronize(); hs = []; t0 = time.perf_counter()
    for _ in range(n):
        h0 = time.perf_counter(); fn(); hs.append(time.perf_counter() - h0)
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, sum(hs) / n * 1e3
print("side stream", side, "rows", rows)
print("replay only, sync each      : wall %.2f ms host %.2f ms" % t(G.graph.replay, sync_each=True))
print("replay only, back to back   : wall %.2f ms host %.2f ms" % t(G.graph.replay))
print("stage only                  : wall %.2f ms host %.2f ms" % t(lambda: (G._stage(None, None), G._stage_consts())))
print("full graphed step           : wall %.2f ms host %.2f ms" % t(lambda: W.train_step(*x)))
W.use_graph = False
W.train_step(*x)
print("eager step                  : wall %.2f ms host %.2f ms" % t(lambda: W.train_step(*x)))
