"""Worker of tests/test_gpu_path.py::test_gradient_buckets_are_reduced_inside_backward_two_ranks and ::test_rccl_world1 (started by
``python -m torch.distributed.run``; not collected by pytest).  Prints one JSON line per rank."""

import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def rccl_world1():
    """backend="nccl" IS RCCL on ROCm: initialise it for a world of one and run the collectives the trainer uses."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
    want = t.clone()
    w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
    w.wait()
    m = torch.tensor([3.5], device="cuda")
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    dist.broadcast(t, src=0)
    dist.barrier()
    torch.cuda.synchronize()
    ok = bool(torch.equal(t, want)) and m.item() == 3.5
    print(json.dumps({"rccl": ok, "backend": dist.get_backend(), "nccl_version": list(torch.cuda.nccl.version())}), flush=True)
    dist.destroy_process_group()


def overlap():
    """Two ranks on one GPU over gloo: a generator-only train step with different data per rank.  Pass 0 and pass 2 keep the gradients
    local (they must repeat bit for bit), pass 1 reduces them (buckets must be launched from inside backward): both ranks must end with
    the same reduced gradients, equal to the mean over ranks of the local ones."""
    from learned_hologram_gan_amd import distributed, hip_ops
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon
    from oracle import seeded  # seeded weights / inputs only (test infrastructure)

    rank, world, _ = distributed.init_from_env("gloo")
    dev = "cuda:0"
    rows = cols = 64
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
    W = watermelon(filter_radius_coefficient=0.45, pad_size=32, distance_stack=stack, input_shape=(1, 4, rows, cols))
    W.generator.load_state_dict(seeded.generator_state_dict())
    W.generator.to(dev).train()
    W.discriminator.to(dev).train()
    W.configure(1, 0.0, 1, 1e-3, 0.0, 1e-3, 1e-3, 0, 10, grad_buckets=4)  # no critic: one backward pass per step
    rgbd, tamp, tphs = seeded.smooth_batch(2, rows, cols, seed=200 + rank)
    idx = torch.tensor([5, 2])
    grabbed = []

    def no_update():  # capture instead of Adam: both passes see the same weights
        hip_ops.join_side_stream()
        torch.cuda.synchronize()
        grabbed.append(W._opt_G.flat.grad.detach().clone())

    W._opt_G.step = no_update
    sync = W._sync_G
    assert sync.enabled and len(sync.ranges) >= 3
    x = (rgbd.to(dev), tamp.to(dev), tphs.to(dev), idx)
    sync.enabled = False
    W.train_step(*x)           # pass 0: local, first sight of every geometry (the GEMM launcher times its tilings here)
    sync.enabled = True
    before = hip_ops.CONTRIBUTIONS
    W.train_step(*x)           # pass 1: reduced across the two ranks, buckets launched from inside backward
    total = hip_ops.CONTRIBUTIONS - before
    log = [(b, c - before, ff) for b, c, ff in sync.launch_log]
    sync.enabled = False
    W.train_step(*x)           # pass 2: local again
    local0, reduced, local = grabbed
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean = sum(gathered) / world
    err = ((reduced - mean).norm() / mean.norm()).item()
    both = [torch.empty_like(reduced) for _ in range(world)]
    dist.all_gather(both, reduced)
    print(json.dumps({"rank": rank, "err": err, "launch_log": log, "contributions": total, "buckets": len(sync.ranges),
                      "local_repeatable": bool(torch.equal(local0, local)), "ranks_agree": bool(torch.equal(both[0], both[1]))}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    {"rccl": rccl_world1, "overlap": overlap}[sys.argv[1]]()
