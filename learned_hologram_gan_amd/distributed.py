"""Single-node data parallelism: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI
on ROCm; "gloo" in the CPU tests).  The reference is single-device, so this layer
is new: the batch is sharded across ranks, every rank holds a full replica, and gradients are
averaged with bucketed all-reduces that start while backward is still running.

Semantics (documented): BatchNorm statistics, the focal-loss max normalisers and the
TV means are per replica; parameter gradients are averaged over ranks.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  No-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # LHG_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box); RCCL needs one GPU per rank
            backend = os.environ.get("LHG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def spawn_local_ranks(argv, n: int, env_extra: dict | None = None, timeout: float | None = None):
    """Start ``n`` rank processes of ``argv`` on this node (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT in
    their environment: the torchrun contract init_from_env reads), wait for all of them and return (exit code, rank 0's stdout).

    Children are plain child processes (never os.exec*), and the caller must not have touched the GPU: a process that has initialised
    HIP cannot be the parent of ranks that each own a GPU on this pool.  Rank 0's stdout is captured and returned (the caller relays the
    one JSON line); the other ranks' stdout goes to stderr of the parent so that nothing but rank 0's line reaches stdout.  If any rank
    exits non-zero the rest are terminated and that code is returned."""
    import socket
    import subprocess
    import sys
    import time

    if n < 1:
        raise ValueError("spawn_local_ranks: n must be >= 1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **(env_extra or {}))
        procs.append(subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading

    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)  # keeps rank 0's pipe drained
    reader.start()
    deadline = None if timeout is None else time.monotonic() + timeout
    code = 0
    try:
        pending = set(range(n))
        while pending and code == 0:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is not None:
                    pending.discard(r)
                    if rc != 0 and code == 0:
                        code = rc
            if pending and code == 0:
                if deadline is not None and time.monotonic() > deadline:
                    code = 124  # timed out
                else:
                    time.sleep(0.1)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    out0 = "".join(c for c in chunks if c)
    return code, out0


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def broadcast_module_state(module: torch.nn.Module, flat_data: torch.Tensor | None = None, src: int = 0, group=None) -> None:
    """Replicas start from rank `src`'s weights and buffers (as DistributedDataParallel does at construction): one broadcast of the
    flat parameter buffer (or one per parameter without it) plus one per floating-point buffer.  No-op for a single process."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        if flat_data is not None:
            dist.broadcast(flat_data, src=src, group=group)
        else:
            for p in module.parameters():
                dist.broadcast(p.data, src=src, group=group)
        for b in module.buffers():
            if b.dtype.is_floating_point or b.dtype == torch.int64:
                dist.broadcast(b, src=src, group=group)


class GradSynchronizer:
    """Average the gradients held in one flat buffer across ranks, overlapped with backward.

    The flat gradient buffer (optim.FlatParams.grad) is cut into ``n_buckets`` contiguous buckets in REVERSE parameter order (the
    last layers' gradients are produced first by backward).  A bucket's all-reduce is launched — asynchronously, from inside
    backward — as soon as the gradient of its last parameter has been ENQUEUED.  Two kinds of arrival feed that count:

    * gradients that autograd accumulates (BatchNorm affine parameters, non-zero biases): a post-accumulate hook;
    * gradients that never pass through autograd (hip_ops._weight_grad accumulates conv / conv-transpose weight gradients straight
      into the flat buffer from the weight-gradient stream and hands autograd ``None``; biases in front of a train-mode BatchNorm
      have an exactly-zero gradient): hip_ops counts the recorded uses of the parameter and calls the listener registered here when
      the last contribution has been enqueued.

    On the GPU the collective is issued with the weight-gradient stream current, after that stream has been made to wait for the main
    stream's work so far: RCCL's stream then starts the reduction behind this bucket's last weight-gradient GEMM while both compute
    streams carry on with the rest of backward.  ``finish()`` launches what is left, waits and divides by the world size.  xGMI is
    point-to-point (7 links x ~153 GB/s per GPU), so a few large buckets (tens of MB) are preferred over many small ones.
    Every rank records the same graph, so buckets complete — and collectives are issued — in the same order everywhere.
    """

    def __init__(self, params, offsets, flat_grad: torch.Tensor, n_buckets: int = 4, group=None, first_bucket_elems: int = 1 << 18,
                 payload: str = "fp32", defer_scale: bool = False):
        """``payload="bf16"`` (the bf16 data-parallel configs, BASELINE configs[2] / [4]): a bucket travels as bf16 — half the bytes per
        xGMI link — and comes back into the fp32 flat buffer (the fp32 master gradient Adam reads); rounding: 2^-9 relative per element
        and per partial sum of the ring.  ``defer_scale``: finish() leaves the SUM in the buffer and ``grad_scale`` = 1 / world for the
        optimiser kernel to apply (FusedAdam.step(grad_scale=...)): one pass over the buffer less per step."""
        if payload not in ("fp32", "bf16"):
            raise ValueError("payload must be 'fp32' or 'bf16'")
        self.params, self.flat_grad, self.group = list(params), flat_grad, group
        self.payload, self.defer_scale, self.grad_scale = payload, defer_scale, 1.0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = len(self.params)
        total = flat_grad.numel()
        target = max(1, total // max(1, n_buckets))
        # Walk parameters from last to first, closing a bucket every `target` elements.  The FIRST bucket (the parameters next to the
        # loss) is kept small — it closes before a parameter that would take it past `first_bucket_elems` (1 MiB, as DDP's first
        # bucket) — so that communication starts early and so that an arrival the op layer cannot observe holds back little: the
        # critic's head conv is visited by the gradient penalty's inner autograd.grad only (hip_ops.only_input_gradients), its recorded
        # use is never answered by a contribution, and its bucket is left to finish().  With the head alone in a 37 KB bucket the
        # three large buckets of the critic are reduced from inside backward.
        self.bucket_of, self.ranges, self.expected = [0] * n, [], []
        hi, count, b = total, 0, 0
        for i in range(n - 1, -1, -1):
            lo = offsets[i]
            if b == 0 and count > 0 and first_bucket_elems and hi - lo > first_bucket_elems and hi - offsets[i + 1] < target:
                self.ranges.append((offsets[i + 1], hi))  # close the small first bucket in front of this parameter
                self.expected.append(count)
                hi, count, b = offsets[i + 1], 0, 1
            self.bucket_of[i] = b
            count += 1
            if hi - lo >= target or i == 0:
                self.ranges.append((lo, hi))
                self.expected.append(count)
                hi, count, b = lo, 0, b + 1
        self._arrived = [0] * len(self.ranges)
        self._seen = [False] * n
        self._real = [False] * n
        self._launched = [False] * len(self.ranges)
        self._work = []
        self._wire = []  # (lo, hi, bf16 copy) of the buckets in flight (payload "bf16")
        self._hooks = []
        self.launch_log = []  # (bucket, hip_ops.CONTRIBUTIONS at launch, launched from finish()?) of the last pass: read by the tests
        self.enabled = self.world > 1
        if self.enabled:
            from . import hip_ops

            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
                self._hooks.append(p.register_hook(self._make_grad_seen(i)))  # tells a defined gradient from None
                hip_ops.set_arrival_listener(p, self._make_listener(i))

    def _arrive(self, i, via_autograd):
        if not self._armed:
            return
        b = self.bucket_of[i]
        # A contribution that lands after the bucket's all-reduce went out is a lost update.  The post-accumulate hook also fires when
        # autograd had NOTHING to add (an op handed back None because it accumulated into the slot itself), so a hook arrival counts as
        # a contribution only if the tensor hook saw a defined gradient for the parameter in this pass.
        if self._launched[b] and (not via_autograd or self._real[i]):
            raise RuntimeError("a gradient arrived for a bucket whose all-reduce had already been launched: a parameter received "
                               "contributions both through autograd and through the flat-buffer path in one backward pass "
                               "(every op must deliver ALL of a parameter's contributions one way: hip_ops.note_use / note_contribution)")
        if self._seen[i]:
            return
        self._seen[i] = True
        self._arrived[b] += 1
        if self._arrived[b] == self.expected[b]:
            self._launch(b, False)

    def _make_hook(self, i):
        return lambda _param: self._arrive(i, True)

    def _make_grad_seen(self, i):
        def seen(grad):
            if self._armed and grad is not None:  # the engine calls tensor hooks with None for an undefined gradient
                self._real[i] = True
        return seen

    def _make_listener(self, i):
        return lambda: self._arrive(i, False)

    _armed = False

    def _launch(self, b, from_finish):
        lo, hi = self.ranges[b]
        from . import hip_ops

        side = hip_ops.side_stream(self.flat_grad.device) if self.flat_grad.is_cuda else None

        def reduce():
            if self.payload == "bf16":
                wire = self.flat_grad[lo:hi].to(torch.bfloat16)  # on the stream the collective is issued on
                self._wire.append((lo, hi, wire))
                return dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            return dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        if side is not None:
            side.wait_stream(torch.cuda.current_stream(self.flat_grad.device))  # gradients accumulated by autograd on the main stream
            with torch.cuda.stream(side):  # the collective queues behind this bucket's last weight-gradient GEMM
                work = reduce()
        else:
            work = reduce()
        self._work.append(work)
        self._launched[b] = True
        self.launch_log.append((b, hip_ops.CONTRIBUTIONS, from_finish))

    def start(self):
        """Call right before ``loss.backward()``."""
        self._arrived = [0] * len(self.ranges)
        self._seen = [False] * len(self.params)
        self._real = [False] * len(self.params)
        self._launched = [False] * len(self.ranges)
        self._work = []
        self._wire = []
        self.launch_log = []
        self._armed = self.enabled
        if self.flat_grad.is_cuda:
            from . import hip_ops

            hip_ops.reset_backward_state()

    def finish(self):
        """Call after backward: launches buckets whose parameters got no gradient this pass and waits for all reductions.

        Contract of the flat gradient buffer afterwards: with ``defer_scale`` (watermelon's choice) it holds the SUM over the ranks and
        ``self.grad_scale`` = 1 / world is what turns it into the data-parallel mean — ``FusedAdam.step(grad_scale=...)`` multiplies on the
        fly; any other reader must use ``mean_gradient()``.  Without ``defer_scale`` the buffer holds the mean and grad_scale is 1.
        With ``payload="bf16"`` every partial sum of the ring all-reduce is rounded to bf16 (2^-9 relative, no error feedback): a wire
        format for the bf16 configs, recorded in the bench line's ``config.grad_payload``."""
        if not self.enabled:
            self.grad_scale = 1.0  # a local pass: the buffer holds this rank's own gradient
            return
        self._armed = False
        from . import hip_ops

        hip_ops.clear_pending_uses(self.params)
        for b, done in enumerate(self._launched):
            if not done:
                self._launch(b, True)
        for w in self._work:
            w.wait()  # the current stream waits for the collective
        if self.flat_grad.is_cuda:
            hip_ops.join_side_stream(self.flat_grad.device)  # weight gradients of buckets that autograd never completed
        for lo, hi, wire in self._wire:  # reduced bf16 sums back into the fp32 master gradient
            if wire.is_cuda:
                wire.record_stream(torch.cuda.current_stream(wire.device))
            self.flat_grad[lo:hi].copy_(wire)
        self._wire = []
        if self.defer_scale:
            self.grad_scale = 1.0 / self.world  # applied by the optimiser kernel (lhg_adam_step_scaled)
        else:
            self.flat_grad.div_(self.world)

    def mean_gradient(self):
        """The data-parallel MEAN gradient as a new tensor (flat_grad * grad_scale): for readers other than the fused optimiser."""
        return self.flat_grad * self.grad_scale

    def remove(self):
        from . import hip_ops

        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            hip_ops.set_arrival_listener(p, None)
