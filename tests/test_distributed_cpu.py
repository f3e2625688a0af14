"""world_size-2 gloo test of the data-parallel layer (distributed.GradSynchronizer + optim.FlatParams):
bucketed, hook-driven all-reduce of a flat gradient buffer must equal the average of the per-rank gradients,
including parameters that receive no gradient in a pass and repeated backward passes.  CPU only — the layer is
device agnostic; on the GPU box the same code runs over RCCL."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.b = torch.nn.Linear(16, 16)
        self.unused = torch.nn.Linear(4, 4)  # never gets a gradient
        self.c = torch.nn.Linear(16, 2)

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.tanh(self.a(x)))))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from learned_hologram_gan_amd import distributed
    from learned_hologram_gan_amd.optim import FlatParams

    r, w, _ = distributed.init_from_env("gloo")
    assert (r, w) == (rank, world) and distributed.world_size() == world
    torch.manual_seed(100 + rank)  # replicas are built from different seeds ...
    net = _Net()
    flat = FlatParams(net)
    distributed.broadcast_module_state(net, flat.data)  # ... and start from rank 0's weights
    torch.manual_seed(100)
    want = torch.cat([p.detach().reshape(-1) for p in _Net().parameters()])
    got = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.equal(got, want)
    assert all(p.data_ptr() == flat.data.data_ptr() + 4 * o for p, o in zip(flat.params, flat.offsets))
    sync = distributed.GradSynchronizer(flat.params, flat.offsets, flat.grad, n_buckets=3)
    assert len(sync.ranges) >= 2 and sum(sync.expected) == len(flat.params)
    assert sorted(sync.ranges)[0][0] == 0 and max(hi for _, hi in sync.ranges) == flat.grad.numel()

    data = [torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + k)) for k in range(world)]
    ok = True
    for it in range(2):  # repeated passes: counters must re-arm
        # expected: average over ranks of the local gradients (every rank can compute all of them)
        expect = torch.zeros_like(flat.grad)
        for k in range(world):
            ref = _Net()
            ref.load_state_dict(net.state_dict())
            ((ref(data[k]) ** 2).mean() * (it + 1)).backward()
            for p_ref, p, o in zip([q_ for q_ in ref.parameters()], flat.params, flat.offsets):
                if p_ref.grad is not None:
                    expect[o:o + p.numel()] += p_ref.grad.reshape(-1) / world
        flat.zero_grad()
        sync.start()
        ((net(data[rank]) ** 2).mean() * (it + 1)).backward()
        sync.finish()
        ok = ok and torch.allclose(flat.grad, expect, rtol=1e-5, atol=1e-7)
        ok = ok and all(sync._launched)
    # outside start()/finish() the hooks must be inert (G backward also reaches the critic's parameters)
    flat.zero_grad()
    (net(data[rank]) ** 2).mean().backward()
    ok = ok and sync._work == [] or not sync._armed
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _tracked_base():
    from learned_hologram_gan_amd import hip_ops

    return hip_ops.TrackedFunction


class _SlotScaleFn(_tracked_base()):
    """y = x * w.sum() + bias * 0 — a stand-in for the conv ops' contract: the weight gradient is ACCUMULATED into the flat slot by
    hip_ops._weight_grad (autograd receives None) and the bias gradient is an exact zero reported through note_contribution."""

    @staticmethod
    def forward(ctx, x, w, bias):
        from learned_hologram_gan_amd import hip_ops

        ctx.save_for_backward(x, w)
        ctx.bias = bias
        hip_ops.note_use(w, ctx.needs_input_grad[1])
        hip_ops.note_use(bias, ctx.needs_input_grad[2])
        return x * w.sum()

    @staticmethod
    def backward(ctx, g):
        from learned_hologram_gan_amd import hip_ops

        x, w = ctx.saved_tensors
        def wgrad(slot):  # the ops' contract: a new tensor without a slot, accumulation into it with one
            val = (g * x).sum() * torch.ones_like(w)
            if slot is None:
                return val
            slot.add_(val)
            return None

        gw = hip_ops._weight_grad(w, (x, g), wgrad)
        hip_ops.note_contribution(ctx.bias)
        return g * w.sum(), gw, None


class _SlotNet(torch.nn.Module):
    """lin0 -> scale (used TWICE: two contributions to one slot) -> lin1.  Parameter order is w, zero_bias, pad, lin0.*, lin1.*, so
    with three buckets (filled from the last parameter backwards) the slot-accumulated weight sits alone in the last bucket, the
    zero-gradient bias shares one with the never-used `pad`, and the first bucket completes through autograd's hooks."""

    def __init__(self):
        super().__init__()
        self.lin0 = torch.nn.Linear(8, 8)
        self.w = torch.nn.Parameter(torch.full((2, 2, 1, 1), 0.3))  # 4-D: FlatParams registers a gradient slot for it
        self.zero_bias = torch.nn.Parameter(torch.zeros(8))
        self.pad = torch.nn.Parameter(torch.zeros(40))  # never used: only finish() can complete its bucket
        self.lin1 = torch.nn.Linear(8, 4)

    def forward(self, x):
        h = torch.tanh(self.lin0(x))
        h = _SlotScaleFn.apply(h, self.w, self.zero_bias)
        h = _SlotScaleFn.apply(torch.tanh(h), self.w, self.zero_bias)
        return self.lin1(h)


def _overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from learned_hologram_gan_amd import distributed, hip_ops
    from learned_hologram_gan_amd.optim import FlatParams

    distributed.init_from_env("gloo")
    torch.manual_seed(7)
    net = _SlotNet()
    flat = FlatParams(net)
    names = [n for n, _ in net.named_parameters()]
    sync = distributed.GradSynchronizer(flat.params, flat.offsets, flat.grad, n_buckets=3)
    data = [torch.randn(6, 8, generator=torch.Generator().manual_seed(50 + k)) for k in range(world)]
    ok = True
    for it in range(2):
        expect = torch.zeros_like(flat.grad)
        for k in range(world):  # reference: plain autograd on a copy with the same maths
            ref = _SlotNet()
            ref.load_state_dict(net.state_dict())
            h = torch.tanh(ref.lin0(data[k]))
            h = h * ref.w.sum()
            h = torch.tanh(h) * ref.w.sum()
            (ref.lin1(h) ** 2).mean().backward()
            for n, p_ref, o in zip(names, ref.parameters(), flat.offsets):
                if p_ref.grad is not None:
                    expect[o:o + p_ref.numel()] += p_ref.grad.reshape(-1) / world
        flat.zero_grad()
        sync.start()
        before = hip_ops.CONTRIBUTIONS
        (net(data[rank]) ** 2).mean().backward()
        in_backward = [(b, c - before) for b, c, from_finish in sync.launch_log if not from_finish]
        sync.finish()
        ok = ok and torch.allclose(flat.grad, expect, rtol=1e-5, atol=1e-7)
        # the slot weight's bucket was launched from INSIDE backward, by the weight's SECOND contribution (the third contribution of
        # the pass: weight, bias, weight) and not by its first; the bucket holding the unused parameter only by finish()
        b_w, b_pad = sync.bucket_of[names.index("w")], sync.bucket_of[names.index("pad")]
        ok = ok and (b_w, 3) in in_backward and len(in_backward) == 2
        ok = ok and [b for b, _, from_finish in sync.launch_log if from_finish] == [b_pad] and all(sync._launched)
        ok = ok and net.w.grad.data_ptr() == flat.grad.data_ptr() + 4 * flat.offsets[names.index("w")]
    # A parameter that receives one contribution through the slot path (counted: its bucket is launched when the count returns to
    # zero) and ANOTHER through autograd's AccumulateGrad afterwards would lose the second one to the all-reduce already in flight:
    # GradSynchronizer must refuse it loudly (the guard used to sit behind the "already seen" early return).
    sync.remove()
    mixed = _SlotNet()
    flat2 = FlatParams(mixed)
    sync2 = distributed.GradSynchronizer(flat2.params, flat2.offsets, flat2.grad, n_buckets=3)
    flat2.zero_grad()
    sync2.start()
    h = _MixedPathFn.apply(torch.tanh(mixed.lin0(data[rank])), mixed.w)
    raised = False
    try:
        (mixed.lin1(h) ** 2).mean().backward()
    except RuntimeError as e:
        raised = "already been launched" in str(e)
    sync2.finish()
    ok = ok and raised
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


class _MixedPathFn(_tracked_base()):
    """Broken on purpose: accumulates into the slot AND hands autograd a gradient for the same parameter."""

    @staticmethod
    def forward(ctx, x, w):
        from learned_hologram_gan_amd import hip_ops

        ctx.save_for_backward(x, w)
        hip_ops.note_use(w, ctx.needs_input_grad[1])
        return x * w.sum()

    @staticmethod
    def backward(ctx, g):
        from learned_hologram_gan_amd import hip_ops

        x, w = ctx.saved_tensors
        val = (g * x).sum() * torch.ones_like(w)
        hip_ops._grad_slot(w).add_(val)
        hip_ops.note_contribution(w)   # the slot path says: complete -> the bucket's all-reduce goes out
        return g * w.sum(), val        # ... and autograd adds a second contribution after it


@pytest.mark.timeout(300)
def test_slot_accumulated_gradients_launch_their_bucket_inside_backward_world2():
    """The conv weight gradients never pass through autograd (hip_ops._weight_grad accumulates them into the flat buffer and returns
    None) and biases in front of a BatchNorm have an exact-zero gradient: their buckets must still be reduced from INSIDE backward."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


@pytest.mark.timeout(300)
def test_bucketed_grad_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _payload_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from learned_hologram_gan_amd import distributed
    from learned_hologram_gan_amd.optim import FlatParams

    distributed.init_from_env("gloo")
    data = [torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + k)) for k in range(world)]
    got = {}
    for payload, defer in (("fp32", False), ("fp32", True), ("bf16", True)):
        torch.manual_seed(100)
        net = _Net()
        flat = FlatParams(net)
        sync = distributed.GradSynchronizer(flat.params, flat.offsets, flat.grad, n_buckets=3, payload=payload, defer_scale=defer)
        flat.zero_grad()
        sync.start()
        (net(data[rank]) ** 2).mean().backward()
        sync.finish()
        got[(payload, defer)] = (flat.grad.clone(), sync.grad_scale)
        sync.remove()
    mean, one = got[("fp32", False)]
    summed, scale = got[("fp32", True)]
    wire, scale16 = got[("bf16", True)]
    ok = one == 1.0 and scale == scale16 == 1.0 / world
    ok = ok and torch.equal(summed * scale, mean)  # the deferred form: the buffer holds the SUM, 1 / world goes to the optimiser kernel
    # the bf16 wire: each rank's share is rounded to bf16 once and so is the ring's sum — relative error <= 2 * 2^-9 per element of the
    # sum's magnitude scale; the result is fp32 in the flat buffer and identical on every rank
    torch.manual_seed(100)
    net = _Net()
    (net(data[rank]) ** 2).mean().backward()
    local = torch.zeros_like(summed)
    for ref, o in zip(net.parameters(), flat.offsets):  # FlatParams keeps module.parameters() order
        if ref.grad is not None:
            local[o:o + ref.numel()] = ref.grad.reshape(-1)
    mags = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(mags, local.abs())
    tol = 2.0 ** -8 * (sum(mags) + summed.abs()) + 1e-12
    ok = ok and bool(((wire - summed).abs() <= tol).all()) and wire.dtype == torch.float32 and not torch.equal(wire, summed)
    both = [torch.empty_like(wire) for _ in range(world)]
    dist.all_gather(both, wire)
    ok = ok and torch.equal(both[0], both[1])
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bf16_gradient_payload_and_deferred_scale_world2():
    """GradSynchronizer(payload="bf16", defer_scale=True) — the bf16 data-parallel configs' wire format — equals the fp32 reduce within
    bf16 rounding, lands in the fp32 flat buffer, and leaves the division by the world size to the optimiser kernel (grad_scale)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_payload_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


@pytest.mark.timeout(300)
def test_bench_gpus_n_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` launched PLAINLY (no torchrun, WORLD_SIZE unset) must start two ranks itself — child processes, before
    anything touches a GPU — relay rank 0's line and fail loudly otherwise (it used to run one rank and report n_gpus 1).
    --spawn-check keeps the ranks to the rendezvous + one all-reduce (gloo here: no GPU in this container)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--spawn-check", "1"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 1.0 and out["rccl_ranks"] is None  # gloo: not RCCL ranks
    # a mismatch between --gpus and the ranks that exist is refused, never reported as a smaller run
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--spawn-check", "1"], cwd=root,
                         env=dict(env, WORLD_SIZE="1", RANK="0"), capture_output=True, text=True, timeout=240)
    assert res.returncode != 0 and "WORLD_SIZE=1" in (res.stderr + res.stdout)


def _equal_batches_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from learned_hologram_gan_amd import distributed, hip_ops

    distributed.init_from_env("gloo")
    ok = True
    hip_ops.set_sync_batch_stats(False)
    hip_ops.assert_equal_batches(3 + rank, "cpu")  # per-replica statistics: nothing to check
    hip_ops.set_sync_batch_stats(True)
    ok = ok and hip_ops.sync_world() == world
    hip_ops.assert_equal_batches(4, "cpu")  # equal batches pass
    try:
        hip_ops.assert_equal_batches(4 - rank, "cpu")  # a ragged last batch: rank 1 holds 3 samples
        ok = False
    except RuntimeError as e:
        ok = ok and "equal per-rank batches" in str(e)
    # a sub-group: statistics follow the group they were given
    sub = dist.new_group([0, 1])
    hip_ops.set_sync_batch_stats(True, group=sub)
    t = torch.tensor([float(rank + 1)])
    ok = ok and hip_ops.all_reduce_(t).item() == 3.0
    hip_ops.set_sync_batch_stats(False)
    ok = ok and hip_ops.sync_world() == 1
    dist.barrier()
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_synchronised_statistics_refuse_ragged_batches_world2():
    """sync_batch_stats weights the replicas equally: unequal per-rank batches must raise, and the reductions follow the
    process group handed to set_sync_batch_stats."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_equal_batches_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_spawn_local_ranks_reports_a_failing_rank():
    import sys

    from learned_hologram_gan_amd.distributed import spawn_local_ranks

    code, out0 = spawn_local_ranks([sys.executable, "-c", "import os, sys; print('r' + os.environ['RANK'] + '/' + os.environ['WORLD_SIZE']); "
                                    "sys.exit(3 if os.environ['RANK'] == '1' else 0)"], 2, timeout=60)
    assert code == 3 and out0.strip() in ("r0/2", "")  # rank 0 may have been terminated before printing
    code, out0 = spawn_local_ranks([sys.executable, "-c", "import os; print(os.environ['MASTER_ADDR'], os.environ['LOCAL_RANK'])"], 1, timeout=60)
    assert code == 0 and out0.split() == ["127.0.0.1", "0"]


def test_single_process_sync_is_a_noop():
    from learned_hologram_gan_amd.distributed import GradSynchronizer
    from learned_hologram_gan_amd.optim import FlatParams

    net = _Net()
    flat = FlatParams(net)
    sync = GradSynchronizer(flat.params, flat.offsets, flat.grad, 4)
    assert not sync.enabled
    flat.zero_grad()
    sync.start()
    net(torch.randn(3, 8)).sum().backward()
    sync.finish()
    assert flat.grad.abs().sum() > 0 and net.unused.weight.grad.abs().sum() == 0
