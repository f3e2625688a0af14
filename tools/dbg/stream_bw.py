"""What streaming kernels reach on this chip: torch copy / add / mul-add on tensors of the step's sizes (GB/s of algorithmic bytes)."""
import torch
dev = "cuda:0"
def t(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for n in (4 * 384 * 384 * 64, 4 * 192 * 192 * 128, 4 * 96 * 96 * 256, 4 * 48 * 48 * 512):
    a, b, c = (torch.rand(n, device=dev) for _ in range(3))
    mb = n * 4 / 1e6
    us = t(lambda: c.copy_(a)); print(f"{mb:7.1f} MB  copy 1R+1W {us:7.1f} us {2 * mb / us:6.2f} TB/s", end="")
    us = t(lambda: torch.add(a, b, out=c)); print(f"   add 2R+1W {us:7.1f} us {3 * mb / us:6.2f} TB/s", end="")
    us = t(lambda: a.sum()); print(f"   sum 1R {us:7.1f} us {mb / us:6.2f} TB/s", end="")
    us = t(lambda: torch.mul(a, b).sum()) ; print(f"   (2R+1W, 1R) {us:7.1f} us")
