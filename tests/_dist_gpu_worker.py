"""Worker of tests/test_gpu_path.py::test_gradient_buckets_are_reduced_inside_backward_two_ranks and ::test_rccl_world1 (started by
``python -m torch.distributed.run``; not collected by pytest).  Prints one JSON line per rank."""

import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
# which tiling variant the autotuner picked for every geometry goes into the kept record (the library appends to this file)
os.environ.setdefault("LHG_TUNE_CACHE", os.path.join("/tmp", f"lhg_tune_rank{os.environ.get('RANK', '0')}_{os.getpid()}.txt"))


def emit(text, flush=True):
    """One write() per record: print() writes the text and the newline separately, and the ranks share the launcher's pipe."""
    sys.stdout.write(text + "\n")
    sys.stdout.flush()


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
    emit(json.dumps({"rccl": ok, "backend": dist.get_backend(), "nccl_version": list(torch.cuda.nccl.version())}), flush=True)
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

    def no_update(grad_scale=1.0):  # capture instead of Adam: both passes see the same weights (the buffer holds the SUM, Adam would scale it)
        hip_ops.join_side_stream()
        torch.cuda.synchronize()
        grabbed.append(W._opt_G.flat.grad.detach().clone() * grad_scale)

    W._opt_G.step = no_update
    sync = W._sync_G
    assert sync.enabled and len(sync.ranges) >= 3
    x = (rgbd.to(dev), tamp.to(dev), tphs.to(dev), idx)
    fwd = []  # (G loss, checksums of the hologram, the reconstruction and the propagated target) of every pass: tells a forward difference from a backward one

    notes = []

    kept = []  # (POH, hat_amps, target_amps) of every pass: the shape of the damage when a pass does not repeat

    def damage(a, b):
        d = (a - b).abs()
        nz = torch.nonzero(d > 0)
        if len(nz) == 0:
            return None
        return {"elements": int(len(nz)), "max_abs": float(d.max()), "planes": sorted({(int(u), int(v)) for u, v in nz[:, :2].tolist()}),
                "rows": sorted(set(nz[:, 2].tolist())), "cols": [int(nz[:, 3].min()), int(nz[:, 3].max()), len(set(nz[:, 3].tolist()))],
                "samples": [{"at": q, "this": float(a[tuple(q)]), "pass0": float(b[tuple(q)])} for q in nz[:: max(1, len(nz) // 6)][:6].tolist()]}

    def one_pass():
        out = W.train_step(*x)
        kept.append(tuple(out[k].detach().clone() for k in ("POH", "hat_amps", "target_amps")))
        fwd.append((float(out["G_loss"]), float(out["POH"].double().sum()), float(out["hat_amps"].double().sum()),
                    float(out["target_amps"].double().sum())))
        with torch.no_grad():  # the reconstruction of the step against fresh evaluations from the same hologram
            for k in range(3):
                again = W.propagator.reconstruct_planes(W.generator.part2.propagator, out["POH"], x[1], x[2], idx)[0]
                if not torch.equal(again, out["hat_amps"]):
                    d = (again - out["hat_amps"]).abs()
                    nz = torch.nonzero(d > 0)
                    note = {"pass": len(fwd) - 1, "recompute": k, "elements": int((d > 0).sum()), "max_abs": float(d.max()),
                            "where": nz[:4].tolist()}
                    if k == 0:  # the SHAPE of the damage and what the differing values look like: which buffer / kernel granularity it matches
                        tgt = out["target_amps"]
                        note.update(planes=sorted({(int(a), int(b)) for a, b in nz[:, :2].tolist()}), rows=sorted(set(nz[:, 2].tolist())),
                                    cols=[int(nz[:, 3].min()), int(nz[:, 3].max()), len(set(nz[:, 3].tolist()))],
                                    flat_offset_bytes=int((((nz[0, 0] * 3 + nz[0, 1]) * rows + nz[0, 2]) * cols + nz[0, 3]) * 4),
                                    data_ptr_mod_2MiB=int(out["hat_amps"].data_ptr() % (1 << 21)),
                                    samples=[{"at": p, "step": float(out["hat_amps"][tuple(p)]), "again": float(again[tuple(p)]),
                                              "target": float(tgt[tuple(p)])} for p in nz[:: max(1, len(nz) // 6)][:6].tolist()],
                                    step_equals_target_there=bool(torch.equal(out["hat_amps"][d > 0], tgt[d > 0])),
                                    finite=bool(torch.isfinite(out["hat_amps"]).all()))
                    notes.append(note)

    def asm_probe(poh, reps=200):
        """Localise a non-repeating reconstruction: the fused operator (three kernels per call) and its two halves — to_spectrum (row
        pass + column FFT x filter) and from_spectrum (column IFFT + row pass) — evaluated `reps` times on fixed inputs while the OTHER
        rank does the same on this GPU; every call is compared bit for bit with the first.  Counts, first offenders and the element
        positions go into the kept JSON record."""
        prop, fixed = W.propagator, W.generator.part2.propagator
        found = {"full": [], "to_spectrum": [], "from_spectrum": []}
        with torch.no_grad():
            ref = prop.reconstruct_planes(fixed, poh, x[1], x[2], idx)
            S0 = torch.cat((fixed.propagate_POH2Freq_forward(poh), prop.filter_AP2filteredFreq(x[1], x[2])), 0)
            back0 = prop.propagate_multiple_samples_with_random_fixed_multiple_distances_freq2amp(S0, idx)
            for r in range(reps):
                again = prop.reconstruct_planes(fixed, poh, x[1], x[2], idx)
                for name, a, b in zip(("hat_amp", "hat_phs", "tgt_amp", "tgt_phs"), again, ref):
                    if not torch.equal(a, b):
                        d = (a - b).abs()
                        found["full"].append({"rep": r, "out": name, "elements": int((d > 0).sum()), "max_abs": float(d.max()), "where": torch.nonzero(d > 0)[:4].tolist()})
                S = torch.cat((fixed.propagate_POH2Freq_forward(poh), prop.filter_AP2filteredFreq(x[1], x[2])), 0)
                if not torch.equal(torch.view_as_real(S), torch.view_as_real(S0)):
                    d = (S - S0).abs()
                    found["to_spectrum"].append({"rep": r, "elements": int((d > 0).sum()), "max_abs": float(d.max()), "where": torch.nonzero(d > 0)[:4].tolist()})
                back = prop.propagate_multiple_samples_with_random_fixed_multiple_distances_freq2amp(S0, idx)
                for name, a, b in zip(("amp", "phs"), back, back0):
                    if not torch.equal(a, b):
                        d = (a - b).abs()
                        found["from_spectrum"].append({"rep": r, "out": name, "elements": int((d > 0).sum()), "max_abs": float(d.max()), "where": torch.nonzero(d > 0)[:4].tolist()})
        return {"reps": reps, "mismatches": {k: len(v) for k, v in found.items()}, "first": {k: v[:3] for k, v in found.items()}}

    sync.enabled = False
    one_pass()                 # pass 0: local, first sight of every geometry (the GEMM launcher times its tilings here)
    sync.enabled = True
    before = hip_ops.CONTRIBUTIONS
    one_pass()                 # pass 1: reduced across the two ranks, buckets launched from inside backward
    total = hip_ops.CONTRIBUTIONS - before
    log = [(b, c - before, ff) for b, c, ff in sync.launch_log]
    sync.enabled = False
    one_pass()                 # pass 2: local again
    dist.barrier()             # both ranks enter the probe together: it measures the operator under the two-process contention of the rig
    probe = asm_probe(W.generator(x[0]).detach())
    local0, reduced, local = grabbed
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean = sum(gathered) / world
    err = ((reduced - mean).norm() / mean.norm()).item()
    both = [torch.empty_like(reduced) for _ in range(world)]
    dist.all_gather(both, reduced)

    def offenders(a, b):  # parameters whose slots differ, worst first: what to look at when a stream race shows
        flat = W._opt_G.flat
        names = {id(p): n for n, p in W.generator.named_parameters()}
        out = []
        for p_, o in zip(flat.params, flat.offsets):
            da, db = a[o:o + p_.numel()], b[o:o + p_.numel()]
            if not torch.equal(da, db):
                out.append((names.get(id(p_), "?"), float((da - db).norm() / (db.norm() + 1e-30))))
        return sorted(out, key=lambda t: -t[1])[:6]

    emit(json.dumps({"rank": rank, "err": err, "launch_log": log, "contributions": total, "buckets": len(sync.ranges),
                      "local_repeatable": bool(torch.equal(local0, local)), "ranks_agree": bool(torch.equal(both[0], both[1])),
                      "forward_repeats": fwd[0] == fwd[1] == fwd[2], "forward": fwd, "recompute_notes": notes, "asm_probe": probe,
                      "damage_vs_pass0": {f"pass{k}.{name}": dmg for k in (1, 2) for name, a, b in zip(("POH", "hat_amps", "target_amps"), kept[k], kept[0])
                                          if (dmg := damage(a, b)) is not None},
                      "tune_cache": open(os.environ["LHG_TUNE_CACHE"]).read().splitlines() if os.path.exists(os.environ.get("LHG_TUNE_CACHE", "")) else None,
                      "diff_pass0_vs_pass2": offenders(local0, local), "diff_reduced_vs_mean": offenders(reduced, mean) if err > 1e-6 else []}),
          flush=True)
    dist.barrier()
    dist.destroy_process_group()


def critic():
    """Two ranks on one GPU over gloo: the FULL step (one critic update with the gradient penalty, then the generator update) with
    different data per rank.  The critic's BatchNorm gammas receive contributions from three forward passes and from the penalty's
    double backward: all of them must be in the flat buffer before the bucket's all-reduce goes out, every large bucket of
    both models must be launched from inside backward, and the reduced gradients must equal the mean of the local ones."""
    from learned_hologram_gan_amd import distributed
    from learned_hologram_gan_amd import hip_ops
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon
    from oracle import seeded

    rank, world, _ = distributed.init_from_env("gloo")
    dev = "cuda:0"
    rows = cols = 64
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
    W = watermelon(filter_radius_coefficient=0.45, pad_size=32, distance_stack=stack, input_shape=(1, 4, rows, cols))
    W.generator.load_state_dict(seeded.generator_state_dict())
    W.discriminator.load_state_dict(seeded.critic_state_dict())
    W.generator.to(dev).train()
    W.discriminator.to(dev).train()
    W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10, grad_buckets=4)
    rgbd, tamp, tphs = seeded.smooth_batch(2, rows, cols, seed=300 + rank)
    idx = torch.tensor([5, 2])
    alphas = [torch.tensor([0.3, 0.8]).view(2, 1, 1, 1).to(dev)]
    grabbed = {"D": [], "G": []}
    logs = {"D": [], "G": []}
    for name, opt, sync in (("D", W._opt_D, W._sync_D), ("G", W._opt_G, W._sync_G)):
        def no_update(grad_scale=1.0, name=name, opt=opt, sync=sync):  # capture instead of Adam: every pass sees the same weights
            hip_ops.join_side_stream()
            torch.cuda.synchronize()
            grabbed[name].append(opt.flat.grad.detach().clone() * grad_scale)
            logs[name].append(list(sync.launch_log))
        opt.step = no_update
    x = (rgbd.to(dev), tamp.to(dev), tphs.to(dev), idx, alphas)

    def one_pass(enabled):
        W._sync_D.enabled = W._sync_G.enabled = enabled
        W.train_step(*x)

    one_pass(False)   # local (first sight of every geometry)
    one_pass(True)    # reduced
    one_pass(False)   # local again
    out = {"rank": rank}
    for name, sync in (("D", W._sync_D), ("G", W._sync_G)):
        local0, reduced, local = grabbed[name]
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        mean = sum(gathered) / world
        both = [torch.empty_like(reduced) for _ in range(world)]
        dist.all_gather(both, reduced)
        out[name] = {"err": ((reduced - mean).norm() / mean.norm()).item(), "local_repeatable": bool(torch.equal(local0, local)),
                     "ranks_agree": bool(torch.equal(both[0], both[1])), "buckets": len(sync.ranges),
                     "bucket_elems": [hi - lo for lo, hi in sync.ranges],
                     "launch_log": [(b, bool(ff)) for b, _, ff in logs[name][1]]}
    emit(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def syncbn():
    """Two ranks x B samples with synchronised batch statistics (configure(sync_batch_stats=True)) against ONE process at 2B: the full
    step — critic update with the gradient penalty (BatchNorm double backward), generator update — must produce the same losses,
    reconstructions, averaged gradients of both models and BatchNorm running statistics.  Rank 0 also runs the
    single-process reference on the concatenated batch."""
    from learned_hologram_gan_amd import distributed, hip_ops
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon
    from oracle import seeded

    rank, world, _ = distributed.init_from_env("gloo")
    dev = "cuda:0"
    rows = cols = 64
    B = 2
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
    data = [seeded.smooth_batch(B, rows, cols, seed=400 + r) for r in range(world)]
    idxs = [torch.tensor([5, 2]), torch.tensor([1, 7])]
    alphas = [torch.tensor([0.3, 0.8]), torch.tensor([0.6, 0.15])]

    def run(batch, idx, alpha, sync):
        W = watermelon(filter_radius_coefficient=0.45, pad_size=32, distance_stack=stack, input_shape=(1, 4, rows, cols))
        W.generator.load_state_dict(seeded.generator_state_dict())
        W.discriminator.load_state_dict(seeded.critic_state_dict())
        W.generator.to(dev).train()
        W.discriminator.to(dev).train()
        W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1, 10, grad_buckets=4, sync_batch_stats=sync)
        if not sync:
            W._sync_G.enabled = W._sync_D.enabled = False
        grads = {}
        for name, opt in (("D", W._opt_D), ("G", W._opt_G)):
            def no_update(grad_scale=1.0, name=name, opt=opt):
                hip_ops.join_side_stream()
                torch.cuda.synchronize()
                grads[name] = opt.flat.grad.detach().clone() * grad_scale
            opt.step = no_update
        out = W.train_step(*(t.to(dev) for t in batch), idx, [alpha.view(-1, 1, 1, 1).to(dev)])
        torch.cuda.synchronize()
        bn = torch.cat([b.detach().flatten().float() for n, b in list(W.generator.named_buffers()) + list(W.discriminator.named_buffers())
                        if n.endswith(("running_mean", "running_var"))])
        names = {"G": [(n, p_.numel()) for n, p_ in zip([k for k, _ in W.generator.named_parameters()], W._opt_G.flat.params)],
                 "D": [(n, p_.numel()) for n, p_ in zip([k for k, _ in W.discriminator.named_parameters()], W._opt_D.flat.params)]}
        return out, grads, W.train_losses_tensor.detach().clone(), bn, names

    out, grads, losses, bn, names = run(data[rank], idxs[rank], alphas[rank], True)
    res = {"rank": rank}
    gathered = [torch.empty_like(losses) for _ in range(world)]
    dist.all_gather(gathered, losses)
    hat = [torch.empty_like(out["hat_amps"]) for _ in range(world)]
    dist.all_gather(hat, out["hat_amps"].contiguous())
    hip_ops.set_sync_batch_stats(False)
    # (every rank runs the reference: configure() broadcasts rank 0's weights, a collective all ranks must enter)
    cat = tuple(torch.cat([data[r][k] for r in range(world)], 0) for k in range(3))
    ref_out, ref_grads, ref_losses, ref_bn, _ = run(cat, torch.cat(idxs), torch.cat(alphas), False)
    # Both ranks have just computed the SAME single-process step (same weights, same data, no collective inside it) concurrently on this
    # GPU: their results must agree bit for bit.  When they do not (the unexplained two-process non-repeat of DESIGN.md §5) the record
    # says which tensor differs first along the step, the shape of the damage and which rank a fresh evaluation sides with.
    agree = {}
    for key in ("POH", "hat_amps", "target_amps"):
        both = [torch.empty_like(ref_out[key]) for _ in range(world)]
        dist.all_gather(both, ref_out[key].contiguous())
        same = bool(torch.equal(both[0], both[1]))
        agree[key] = same
        if not same:
            d = (both[0] - both[1]).abs()
            nz = torch.nonzero(d > 0)
            info = {"elements": int(len(nz)), "max_abs": float(d.max()), "planes": sorted({(int(a), int(b)) for a, b in nz[:, :2].tolist()}),
                    "rows": sorted(set(nz[:, 2].tolist())), "cols": [int(nz[:, 3].min()), int(nz[:, 3].max()), len(set(nz[:, 3].tolist()))],
                    "samples": [{"at": q, "rank0": float(both[0][tuple(q)]), "rank1": float(both[1][tuple(q)])} for q in nz[:: max(1, len(nz) // 6)][:6].tolist()]}
            if key == "hat_amps":  # arbiter: the reconstruction evaluated again from rank 0's hologram
                with torch.no_grad():
                    W2 = watermelon(filter_radius_coefficient=0.45, pad_size=32, distance_stack=stack, input_shape=(1, 4, rows, cols))
                    again = W2.propagator.reconstruct_planes(W2.generator.part2.propagator, ref_out["POH"], cat[1].to(dev), cat[2].to(dev), torch.cat(idxs))[0]
                info["fresh_evaluation_equals"] = {"rank0": bool(torch.equal(again, both[0])), "rank1": bool(torch.equal(again, both[1]))}
            agree[key + "_damage"] = info
    for m in ("G", "D"):
        both = [torch.empty_like(ref_grads[m]) for _ in range(world)]
        dist.all_gather(both, ref_grads[m])
        agree["grads_" + m] = bool(torch.equal(both[0], both[1]))
    res["single_process_step_agrees_between_ranks"] = agree
    # The noise floor of the comparison, measured in the same worker: the SAME single-process step with the samples in
    # reverse order — mathematically the same batch statistics, losses and averaged gradients, another fp32 summation order.
    flip = lambda t: torch.flip(t, dims=(0,))  # noqa: E731
    perm_out, perm_grads, _, _, _ = run(tuple(flip(t) for t in cat), flip(torch.cat(idxs)), flip(torch.cat(alphas)), False)
    if rank == 0:
        l2f = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()  # noqa: E731
        res.update(floor_g=l2f(perm_grads["G"], ref_grads["G"]), floor_d=l2f(perm_grads["D"], ref_grads["D"]),
                   floor_hat=l2f(flip(perm_out["hat_amps"]), ref_out["hat_amps"]))
    if rank == 0:
        l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()  # noqa: E731
        mean_losses = sum(gathered) / world
        # the TV value every rank reports under sync is already the global one; the other terms average over the ranks
        worst = {}
        for m in ("G", "D"):  # which parameters carry the difference (a wrong normaliser shows as a whole family off by a factor)
            o, rows_ = 0, []
            for n, k in names[m]:
                a, b = grads[m][o:o + k], ref_grads[m][o:o + k]
                o += k
                if b.norm() > 0:
                    rows_.append((n, round(l2(a, b), 6), round((a.norm() / b.norm()).item(), 5)))
            worst[m] = sorted(rows_, key=lambda t: -t[1])[:6]
        res.update(worst=worst)
        res.update(g_err=l2(grads["G"], ref_grads["G"]), d_err=l2(grads["D"], ref_grads["D"]), bn_err=l2(bn, ref_bn),
                   hat_err=l2(torch.cat(hat, 0), ref_out["hat_amps"]), losses=mean_losses.tolist(), ref_losses=ref_losses.tolist())
    emit(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def syncops():
    """The synchronised operators one by one, where the comparison is well conditioned: a train-mode BatchNorm with its backward and
    its double backward (the gradient penalty's pattern) and the reconstruction losses, two ranks x half the batch against one process
    on the whole batch.  Per-rank loss = mean over the rank's samples, so input gradients must come out W times the single process's,
    parameter gradients average to it, outputs and loss values match."""
    from learned_hologram_gan_amd import distributed, hip_ops
    from learned_hologram_gan_amd.poh_ops import ReconLossFn

    rank, world, _ = distributed.init_from_env("gloo")
    dev = "cuda:0"
    g = torch.Generator().manual_seed(77)
    N, H, Wd, C = 4, 12, 10, 64
    x_all = (torch.randn((N, H, Wd, C), generator=g) * torch.linspace(0.2, 3.0, C) + torch.linspace(-2, 2, C)).to(dev)
    p1_all, p2_all = torch.randn((N, H, Wd, C), generator=g).to(dev), torch.randn((N, H, Wd, C), generator=g).to(dev)
    gamma0, beta0 = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    l2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300)).item()  # noqa: E731

    def bn_case(x, p1, p2, act, sync):
        hip_ops.set_sync_batch_stats(sync)
        x = x.clone().requires_grad_(True)
        gamma, beta = gamma0.clone().requires_grad_(True), beta0.clone().requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        y = hip_ops.BatchNormTrainFn.apply(x, gamma, beta, rm, rv, None, act, 0.2, None)
        (gx,) = torch.autograd.grad((y * p1).sum(), x, create_graph=True)       # first backward, differentiable (the penalty's inner grad)
        n_local = x.shape[0]
        loss = ((gx * gx).sum() + (y * p2).sum()) / n_local                        # second-order term + a first-order term
        loss.backward()
        return y.detach(), gx.detach(), x.grad, gamma.grad, beta.grad, rm, rv, loss.detach()

    res = {"rank": rank}
    half = slice(rank * N // world, (rank + 1) * N // world)
    for act, tag in ((hip_ops.ACT_NONE, "bn"), (hip_ops.ACT_LEAKY, "bn_leaky")):
        y, gx, dx, dg, db, rm, rv, loss = bn_case(x_all[half], p1_all[half], p2_all[half], act, True)
        for t in (dg, db):
            dist.all_reduce(t)
            t /= world
        lsum = loss.clone()
        dist.all_reduce(lsum)
        Y, GX, DX, DG, DB, RM, RV, LOSS = bn_case(x_all, p1_all, p2_all, act, False)
        res[tag] = dict(y=l2(y, Y[half]), gx=l2(gx, GX[half]), dx=l2(dx, world * DX[half]), dgamma=l2(dg, DG), dbeta=l2(db, DB), run_mean=l2(rm, RM),
                        run_var=l2(rv, RV), loss=abs(float(lsum / world) - float(LOSS)) / abs(float(LOSS)))

    # reconstruction losses
    B, Hh, Ww = 4, 24, 20
    ha, ta = torch.rand((B, 3, Hh, Ww), generator=g).to(dev), torch.rand((B, 3, Hh, Ww), generator=g).to(dev)
    hp, tp = (6.28 * torch.rand((B, 3, Hh, Ww), generator=g)).to(dev), (6.28 * torch.rand((B, 3, Hh, Ww), generator=g)).to(dev)
    wts = torch.tensor([1.0, 0.7, 0.3], device=dev)

    def loss_case(sl, sync):
        hip_ops.set_sync_batch_stats(sync)
        a, p_ = ha[sl].clone().requires_grad_(True), hp[sl].clone().requires_grad_(True)
        out = ReconLossFn.apply(a, ta[sl], p_, tp[sl])
        (out * wts).sum().backward()
        return out.detach(), a.grad, p_.grad

    hb = slice(rank * B // world, (rank + 1) * B // world)
    out, ga, gp = loss_case(hb, True)
    osum = out.clone()
    dist.all_reduce(osum)
    OUT, GA, GP = loss_case(slice(0, B), False)
    mean_out = osum / world
    res["recon"] = dict(focal=abs(float(mean_out[0] - OUT[0])) / float(OUT[0]), pixel=abs(float(mean_out[1] - OUT[1])) / float(OUT[1]),
                        tv_reported=abs(float(out[2] - OUT[2])) / float(OUT[2]),  # every rank reports the GLOBAL TV difference
                        g_amp=l2(ga, GA[hb] * world), g_phs=l2(gp, GP[hb] * world))
    hip_ops.set_sync_batch_stats(False)
    emit(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    {"rccl": rccl_world1, "overlap": overlap, "critic": critic, "syncbn": syncbn, "syncops": syncops}[sys.argv[1]]()
