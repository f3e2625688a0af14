#!/usr/bin/env python3
"""Benchmark of the RGBD -> POH hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one batch of the GAN training loop of the reference (watermelon.py:207-277): generator
forward, fused angular-spectrum reconstruction of hat/target at one random plane per sample,
`d_ratio` critic updates with the gradient penalty (double backward), generator loss + backward,
both Adam updates — on BASELINE.json configs[1]: 384x384x3, batch 4 per GPU, fp32, synthetic
random-RGBD inputs resident in HBM, reference-style random-init weights.  Weak scaling: every rank
runs the same per-GPU batch and gradients are averaged over RCCL (all-reduce overlapped with backward).

Prints ONE JSON line (rank 0).
* The TIMED REGION is un-instrumented: W warm-up steps, barrier + synchronize, K steps, barrier + synchronize, max over ranks.
* `roofline` (dominant kernel: the MFMA gather-GEMM behind every convolution / input-gradient / conv-transpose) comes from a SEPARATE
  pass over the same workload right after the timed region, with HIP events recorded on the launch stream around every launch:
  achieved = algorithmic conv FLOPs the kernel executed / its summed launch durations.  The headline figure is measured under the
  timed region's conditions (weight-gradient GEMMs running concurrently on the second stream, so durations include time sharing
  the GPU); `roofline.isolated` repeats it with the second stream off (the kernel's own duration — what rocprofv3 --kernel-trace
  shows, since the profiler serialises dispatches).
* `secondary` carries north_star's second target: 4K (3840x2160) batch-1 inference + propagation of the hologram to 8 planes.
* `reference_cli_default_step` (informational, N = 1): the same data through the step as the reference's trainingModel.py configures it
  (five critic updates per generator update, perceptual term 0.1) — not the configuration BASELINE.json's metric is quoted on.
* `cpu_baseline` times the CPU oracle (a port of the reference step, oracle/step.py) on the host cores: 1 warm-up + median of 3
  runs of a bounded sample.
"""

from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same table)


def kernel_sources():
    """Every kernel source of the conv engine (conv_engine.hip, wgrad6.hip, the .inc and wg6_*.h files they include): globs, so a new include cannot be forgotten."""
    d = os.path.join(REPO, "learned_hologram_gan_amd", "csrc")
    return sorted(glob.glob(os.path.join(d, "*.inc")) + glob.glob(os.path.join(d, "wg6_*.h")) + [os.path.join(d, "conv_engine.hip"), os.path.join(d, "wgrad6.hip"), os.path.join(d, "common.h")])


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="frames per GPU per step")
    ap.add_argument("--rows", type=int, default=384)
    ap.add_argument("--cols", type=int, default=384)
    ap.add_argument("--pad", type=int, default=320)
    ap.add_argument("--d-ratio", type=int, default=1, help="critic updates per generator update (BASELINE.md assembled step: 1)")
    ap.add_argument("--cpu-baseline", type=int, default=1, help="0 skips the CPU oracle timing")
    ap.add_argument("--cpu-rows", type=int, default=0, help="frame size of the CPU sample (0 = same as --rows)")
    ap.add_argument("--cpu-batch", type=int, default=0, help="frames per CPU step (0 = same as --batch)")
    ap.add_argument("--secondary", type=int, default=1, help="0 skips the 4K inference leg (north_star's second target)")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps of each instrumented pass behind `roofline`")
    ap.add_argument("--cli-default", type=int, default=1, help="0 skips the informational timing of the reference CLI's default step (d_ratio 5, perceptual 0.1)")
    ap.add_argument("--other-modes", type=int, default=1, help="0 skips the informational timing of the other GEMM formulations")
    ap.add_argument("--graph", type=int, default=-1,
                    help="replay a captured hipGraph instead of eager launches.  train mode: the whole step as one replay (graph.GraphedTrainStep), "
                         "single process only (collectives are not captured); default off in both modes (see DESIGN.md §6: the step is GPU-bound)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 (the benchmark metric): fp32 tensors, conv GEMMs in the library's default fp32-faithful mode.  bf16 (BASELINE configs[2]/[4], informational): bf16 conv-GEMM operands, "
                         "fp32 accumulation")
    ap.add_argument("--precision", choices=("default", "fp32", "fp32_split", "fp32_split2", "fp32_split_f16", "bf16"), default="default",
                    help="conv-GEMM arithmetic (hip_ops.set_conv_precision); default: fp32 tensors with the library's default fp32 GEMM mode, "
                         "or bf16 with --dtype bf16")
    ap.add_argument("--perceptual", type=float, default=0.0,
                    help="weight of the VGG19 perceptual term (reference CLI: 0.1; seeded random VGG weights: throughput only).  The benchmark "
                         "metric follows BASELINE.md's assembled step, which has no VGG term")
    ap.add_argument("--planes", type=int, default=0,
                    help="infer mode: also propagate the hologram to this many planes (BASELINE configs[3]: inference + multi-plane propagate)")
    ap.add_argument("--spawn-check", type=int, default=0,
                    help="1: every rank only joins the process group (RCCL on GPUs, gloo without), all-reduces its rank and rank 0 prints a short line "
                         "(n_gpus, rccl_ranks, sum of ranks) — checks the rank launch / rendezvous without running the workload (CPU test of --gpus N)")
    ap.add_argument("--mode", choices=("train", "infer"), default="train",
                    help="train: the GAN step (the benchmark metric); infer: eval-mode generator forward RGBD->POH only (informational)")
    return ap.parse_args()


def kernel_source_sha16():
    h = hashlib.sha256()
    for path in kernel_sources():
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic():
    """HBM bytes per gather-GEMM launch from the newest committed PMC passes (tools/pmc_traffic.py), with the revision they belong to."""
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic*.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                d = json.load(f)
            src = {"file": os.path.relpath(path, REPO), "git_sha": d.get("git_sha"), "kernel_src_sha16": d.get("kernel_src_sha16")}
            src["matches_this_build"] = d.get("kernel_src_sha16") == kernel_source_sha16()
            return d["kernels"]["gg"]["hbm_bytes_per_launch_corrected"], src
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def pmc_traffic_4k():
    """HBM bytes per gather-GEMM launch of the 4K leg from the newest committed PMC passes (tools/profile_4k.sh), with their revision."""
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_4k_pmc_traffic.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                d = json.load(f)
            fam = d["families"]["gather-GEMM"]
            return fam["hbm_bytes_per_launch_corrected"], {"file": os.path.relpath(path, REPO), "git_sha": d.get("git_sha"),
                                                           "launches_per_frame": fam.get("launches_per_frame"),
                                                           "hbm_bytes_per_frame_corrected": fam.get("hbm_bytes_per_frame_corrected")}
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def cpu_baseline(args):
    """The same train step through the CPU oracle (checker code timed as the reported baseline) ON THE BENCH WORKLOAD — the same frame
    size and the same batch per step as the GPU line beside it — 1 warm-up + the median of 3 runs while they fit the
    leg's ~65 s budget; a run that would overshoot it is not started and the median of what was measured is reported (the count is in
    `sample`)."""
    from oracle import seeded, step

    # the GPU box gives one GPU a 16-CPU share although it reports every core of the host
    threads = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(threads)
    rows = args.cpu_rows or args.rows
    cols = args.cpu_rows or args.cols
    B = args.cpu_batch or args.batch
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1]
    st = step.make_state(rows, cols, args.pad, 0.45, stack, seeded.generator_state_dict(), seeded.critic_state_dict())
    rgbd, amp, phs = seeded.synthetic_batch(B, rows, cols)
    w = step.LossWeights(d_ratio=args.d_ratio)
    idx = torch.tensor([(7 + 3 * b) % 20 for b in range(B)])

    def one():
        t0 = time.perf_counter()
        step.train_step(st, rgbd, amp, phs, w, idx, [torch.full((B, 1, 1, 1), 0.5) for _ in range(max(args.d_ratio, 1))])
        return time.perf_counter() - t0

    budget, t_start = 65.0, time.perf_counter()
    warm = one()
    times = []
    while len(times) < 3 and (not times or time.perf_counter() - t_start + max(times) < budget):
        times.append(one())
    dt = statistics.median(times)
    return {"value": round(B / dt, 5), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{B} frames {rows}x{cols} per run (the bench workload: batch {B}), one full train step (G fwd+bwd, {args.d_ratio} critic "
                      f"update(s) with gradient penalty, Adam x2) through oracle/step.py; 1 warm-up ({warm:.1f} s) + median of {len(times)} run(s) ("
                      + ", ".join(f"{t:.1f} s" for t in times) + f"): {dt:.2f} s per step = {dt / B:.2f} s per frame"}


def roofline_block(res, steps, peak):
    gg, wg = res
    rate = lambda r: r["algorithmic_flops"] / (r["total_ms"] * 1e-3) / 1e12 if r["total_ms"] > 0 else 0.0  # noqa: E731
    a = rate(gg)
    return {"achieved": round(a, 3), "frac": round(a / peak, 4),
            "algorithmic_flop_per_launch": round(gg["algorithmic_flops"] / max(gg["launches"], 1)),
            "launches_per_step": gg["launches"] / steps,
            "avg_launch_us": round(gg["total_ms"] * 1e3 / max(gg["launches"], 1), 2),
            "algorithmic_gflop_per_step": round(gg["algorithmic_flops"] / steps / 1e9, 2),
            "executed_gflop_per_step": round(gg["executed_flops"] / steps / 1e9, 2),
            "kernel_ms_per_step": round(gg["total_ms"] / steps, 3),
            "wgrad_kernel": {"achieved": round(rate(wg), 3), "unit": "TFLOP/s", "kernel_ms_per_step": round(wg["total_ms"] / steps, 3),
                             "algorithmic_gflop_per_step": round(wg["algorithmic_flops"] / steps / 1e9, 2)}}


def secondary_4k(native, dev, world, sync, peak, peak_note, planes=8, warmup=2, steps=5):
    """north_star's second target: RGBD -> POH for one 3840x2160 frame (batch 1, pad 72 -> 2304x4096 transforms) + the hologram
    propagated to `planes` planes (generatePOH.py --propagate).  Replicas only across GPUs (one frame per GPU)."""
    from learned_hologram_gan_amd.angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_multiple_distances as Mu
    from learned_hologram_gan_amd.watermelon_hologram.generator import Generator

    rows, cols, pad = 2160, 3840, 72
    wl = torch.tensor([638e-9, 520e-9, 450e-9])
    G = Generator(rows, cols, pad, 0.45, 3, 3.74e-6, wl, torch.tensor([1e-3])).to(dev).eval()
    d = torch.linspace(4e-4, 10e-4, planes)
    prop = Mu(sample_row_num=rows, sample_col_num=cols, distances=d, pad_size=pad, filter_radius_coefficient=0.35, pixel_pitch=3.74e-6,
              wave_length=wl, band_limit=False, cuda=True)
    x = torch.rand((1, 4, rows, cols), generator=torch.Generator().manual_seed(5)).to(dev)

    def run():
        poh = G(x)
        return prop(torch.ones_like(poh), poh, d)

    with torch.no_grad():
        for _ in range(warmup):
            run()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        sync()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
        with native.kernel_profile() as prof:
            for _ in range(steps):
                run()
            torch.cuda.synchronize()
    gg = prof.result[0]
    a = gg["algorithmic_flops"] / (gg["total_ms"] * 1e-3) / 1e12 if gg["total_ms"] > 0 else 0.0
    return {"metric": "RGBD->POH frames/sec at 3840x2160 bs=1 (eval-mode generator + propagation to 8 planes)",
            "value": round(world * steps / dt, 4), "unit": "frames/s", "ms_per_frame": round(dt / steps * 1e3, 3), "steps": steps, "warmup": warmup,
            "config": {"workload": f"{cols}x{rows}x3 bs=1/GPU generator forward (UNet + ASM back-propagation + POH encode, pad {pad} -> "
                                   f"{rows + 2 * pad}x{cols + 2 * int(pad * cols / rows)} transforms) + {planes}-plane propagate; replicas only"},
            "roofline": {"kernel": "gather-GEMM (the frame's 12.89 TFLOP of convolutions)", "bound": "mfma", "achieved": round(a, 3), "peak": peak,
                         "peak_note": peak_note, "unit": "TFLOP/s", "frac": round(a / peak, 4),
                         "frac_of_fp32_mfma_peak": round(a / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": pmc_traffic_4k()[0], "traffic_source": pmc_traffic_4k()[1],
                         "kernel_ms_per_frame": round(gg["total_ms"] / steps, 3),
                         "algorithmic_gflop_per_frame": round(gg["algorithmic_flops"] / steps / 1e9, 1)}}


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` launched plainly (no torchrun, WORLD_SIZE unset): this process touches no GPU, starts N rank
    processes of this same command line (child processes, never an exec), relays rank 0's JSON line and returns non-zero if any
    rank failed — it never reports n_gpus = 1 for a request of N."""
    from learned_hologram_gan_amd import distributed

    code, out0 = distributed.spawn_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus)
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    sys.stdout.write(out0 if not lines else "".join(ln + "\n" for ln in out0.splitlines() if not ln.startswith("{")))
    if code != 0:
        sys.stderr.write(f"bench.py: a rank of the --gpus {args.gpus} run exited with code {code}\n")
        return code
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
        return 1
    line = json.loads(lines[-1])
    if line.get("n_gpus") != args.gpus:
        sys.stderr.write(f"bench.py: asked for {args.gpus} ranks, the line reports {line.get('n_gpus')}\n")
        return 1
    print(lines[-1], flush=True)
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))  # before anything touches the GPU
    from learned_hologram_gan_amd import distributed

    if args.spawn_check:
        rank, world, _ = distributed.init_from_env()
        t = torch.tensor([float(rank)], device="cuda" if torch.cuda.is_available() else "cpu")
        if world > 1:
            torch.distributed.all_reduce(t)
        if rank == 0:
            nccl = world > 1 and torch.distributed.get_backend() == "nccl"
            print(json.dumps({"spawn_check": True, "n_gpus": world, "rccl_ranks": torch.distributed.get_world_size() if nccl else (1 if world == 1 else None),
                              "rank_sum": t.item()}), flush=True)
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        return
    from learned_hologram_gan_amd import hip_ops, native
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    rank, world, local = distributed.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a different number of ranks than asked for")
    rccl_ranks = torch.distributed.get_world_size() if (world > 1 and torch.distributed.get_backend() == "nccl") else (1 if world == 1 else None)
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(122731 + rank)

    if args.precision != "default":
        hip_ops.set_conv_precision(args.precision)
        args.dtype = "bf16" if args.precision == "bf16" else "f32"
    elif args.dtype == "bf16":
        hip_ops.set_activation_storage("bf16")  # bf16 NHWC activations + bf16 conv-GEMM operands; fp32 accumulation / BN / FFT / Adam
    bf16 = args.dtype == "bf16"
    mode = hip_ops.conv_precision()
    # the matrix roofline of the GEMM formulation in use: algorithmic (fp32 multiply-add) TFLOP/s it can reach at most
    products = {"fp32": None, "fp32_split": 6, "fp32_split2": 3, "fp32_split_f16": 3, "bf16": 1}[mode]
    mfma_peak = FP32_MFMA_PEAK_TFLOPS if products is None else round(BF16_MFMA_PEAK_TFLOPS / products, 1)
    peak_note = ("dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32)" if products is None else
                 f"dense 16-bit MFMA peak {BF16_MFMA_PEAK_TFLOPS:.0f} TFLOP/s (bf16 and fp16 alike) / {products} product(s) per multiply-add of the '{mode}' formulation"
                 + ("" if bf16 else f"; the exact-fp32 MFMA formulation peaks at {FP32_MFMA_PEAK_TFLOPS}"))
    # the arithmetic type the path computes in: fp32 tensors everywhere; the conv GEMMs' formulation is part of the label because the default
    # one is NOT the reference's arithmetic (fp32-faithful emulation on the fp16 matrix pipe; tests/test_gpu_truth.py measures it per mode)
    dtype_label = {"fp32": "f32", "fp32_split_f16": "f32 (conv GEMMs: fp16x2-split operands, 3 MFMA products, fp32 accumulate)",
                   "fp32_split": "f32 (conv GEMMs: bf16x3-split operands, 6 MFMA products, fp32 accumulate)",
                   "fp32_split2": "f32 (conv GEMMs: bf16x2-split operands, 3 MFMA products: ~2^-16)", "bf16": "bf16"}[mode]

    def build_trainer():
        stack = torch.linspace(-4e-4, 0.0, 21)[:-1]  # trainingModel.py:62
        perceptual = None
        if args.perceptual > 0:
            import warnings

            from learned_hologram_gan_amd.watermelon_hologram.perceptual import perceptualLoss

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                perceptual = perceptualLoss()
        T = watermelon(filter_radius_coefficient=0.45, pad_size=args.pad, distance_stack=stack, input_shape=(1, 4, args.rows, args.cols),
                       perceptual_loss=perceptual)
        T.generator.to(dev).train()
        T.discriminator.to(dev).train()
        T.configure(phs_gradient_loss_weight=1, perceptual_loss_weight=args.perceptual, pixel_loss_weight=1, TV_loss_weight=1e-3,
                    discriminator_loss_weight=1e-1, lr_G=1e-3, lr_D=1e-3, discriminator_train_ratio=args.d_ratio, discriminator_lambda=10)
        return T

    # One hipGraph replay per step (graph.GraphedTrainStep) is built and parity-tested, but NOT the default of the timed region: measured
    # on MI355X / ROCm 7.2 (tools/dbg/graph_probe.py) the step is GPU-bound (eager host enqueue 20.6 ms against 36.5 ms of
    # GPU time) and the two-branch graph runs its branches with less overlap than the two eager streams do (38.3 against 36.5 ms).
    train_graph = world == 1 and args.graph > 0

    W = build_trainer()
    W.use_graph = train_graph
    g = torch.Generator().manual_seed(122731 + rank)
    B = args.batch
    rgbd = torch.rand((B, 4, args.rows, args.cols), generator=g).to(dev)
    tamp = torch.rand((B, 3, args.rows, args.cols), generator=g).to(dev)
    tphs = torch.rand((B, 3, args.rows, args.cols), generator=g).to(dev)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if args.mode == "infer":
        W.generator.eval()
        run = W.generator
        if args.planes > 0:  # generatePOH.py --propagate: |crop(ifft2(fft2(pad(e^{i POH})) H(d) mask))| at `planes` distances
            from learned_hologram_gan_amd.angular_spectrum_method import bandLimitedAngularSpectrumMethod_for_multiple_distances as Mu

            dist_p = torch.linspace(4e-4, 10e-4, args.planes)
            prop = Mu(sample_row_num=args.rows, sample_col_num=args.cols, distances=dist_p, pad_size=args.pad, filter_radius_coefficient=0.35,
                      pixel_pitch=3.74e-6, wave_length=torch.tensor([638e-9, 520e-9, 450e-9]), band_limit=False, cuda=True)
            gen = W.generator

            def run(x):  # noqa: F811
                poh = gen(x)
                return prop(torch.ones_like(poh), poh, dist_p)
        if args.graph > 0:
            from learned_hologram_gan_amd.graph import GraphedGenerator

            graphed = GraphedGenerator(W.generator, rgbd)
            run = lambda x: graphed(x, clone=False)  # noqa: E731
        with torch.no_grad():
            for _ in range(args.warmup):
                run(rgbd)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run(rgbd)
            sync()
            dt = time.perf_counter() - t0
        if rank == 0:
            print(json.dumps({"metric": "RGBD->POH inference frames/sec (eval-mode generator forward, informational)",
                              "value": round(B * world * args.steps / dt, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
                              "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                              "config": {"workload": f"{args.rows}x{args.cols}x3 bs={B} generator forward (UNet + ASM back-propagation + POH encode)"
                                                     + (f" + propagation to {args.planes} planes" if args.planes > 0 else "")
                                                     + (", hipGraph replay" if args.graph > 0 else "")}}))
        return

    # LHG_MAIN_PRIORITY=high (A/B measurements): the step's main chain of kernels on a high-priority stream
    if os.environ.get("LHG_MAIN_PRIORITY") == "high":
        _least, greatest = torch.cuda.Stream.priority_range()
        torch.cuda.set_stream(torch.cuda.Stream(dev, priority=greatest))
    # ---- the timed region: no instrumentation of any kind
    for _ in range(args.warmup):
        W.train_step(rgbd, tamp, tphs)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        W.train_step(rgbd, tamp, tphs)
    host_enqueue = time.perf_counter() - t0  # the host's share: it has handed every launch / replay of the K steps to the stream
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = t.item()

    # ---- roofline passes (every rank: the steps contain collectives).  Pass 1: the timed region's conditions.  Pass 2: second stream off.
    P = max(1, min(args.steps, args.profile_steps))
    W.use_graph = False  # the per-launch HIP events of the roofline passes need eager launches
    W.train_step(rgbd, tamp, tphs)
    torch.cuda.synchronize()
    with native.kernel_profile() as prof:
        for _ in range(P):
            W.train_step(rgbd, tamp, tphs)
        torch.cuda.synchronize()
    iso = None
    if hip_ops.SIDE_WGRAD:
        hip_ops.SIDE_WGRAD = False
        W.train_step(rgbd, tamp, tphs)
        torch.cuda.synchronize()
        with native.kernel_profile() as prof_iso:
            for _ in range(P):
                W.train_step(rgbd, tamp, tphs)
            torch.cuda.synchronize()
        hip_ops.SIDE_WGRAD = True
        iso = prof_iso.result

    out = None
    if rank == 0:
        traffic, traffic_src = (None, None) if mode != hip_ops.default_precision() else pmc_traffic()
        kernel = {"fp32": "lhg::gg2_kernel / gg_kernel (exact fp32 MFMA gather-GEMM)",
                  "fp32_split": "lhg::gg3s_kernel (gather-GEMM on the bf16 matrix pipe, fp32-faithful: operands as exact sums of three bf16 terms, "
                                "six MFMA products per multiply, fp32 accumulation)",
                  "fp32_split2": "lhg::gg3s_kernel (two bf16 terms, three MFMA products per multiply)",
                  "fp32_split_f16": "lhg::gg3s_kernel (gather-GEMM on the fp16 matrix pipe, fp32-faithful: tensor-scaled operands as sums of two "
                                    "fp16 terms, three MFMA products per multiply, fp32 accumulation)",
                  "bf16": "lhg::gg2b_kernel / gg3s_kernel (bf16 MFMA gather-GEMM)"}[mode]
        r = {"kernel": kernel + ": conv forward / input-gradient / conv-transpose", "bound": "mfma", "peak": mfma_peak, "peak_note": peak_note,
             "unit": "TFLOP/s", "traffic": traffic, "traffic_source": traffic_src}
        r.update(roofline_block(prof.result, P, mfma_peak))
        r["frac_of_fp32_mfma_peak"] = round(r["achieved"] / FP32_MFMA_PEAK_TFLOPS, 4)
        r["measured"] = (f"HIP events around every launch over {P} steps of the same workload right after the (un-instrumented) timed region, "
                         "under its conditions: weight-gradient GEMMs run concurrently on a second HIP stream, so launch durations include "
                         "time sharing the GPU")
        if iso is not None:
            r["isolated"] = roofline_block(iso, P, mfma_peak)
            r["isolated"]["frac_of_fp32_mfma_peak"] = round(r["isolated"]["achieved"] / FP32_MFMA_PEAK_TFLOPS, 4)
            r["isolated"]["measured"] = (f"{P} more steps with the second stream off: the kernel's own duration (what rocprofv3 --kernel-trace "
                                         "reports, since it serialises dispatches)")
        out = {
            "metric": "RGBD->POH frames/sec at 384x384 bs=4 (GAN train step)" + (" [bf16 mode, informational]" if bf16 else ""),
            "value": round(B * world * args.steps / elapsed, 4),
            "unit": "frames/s",
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,  # dist.get_world_size() on the nccl (= RCCL) backend; None when the ranks talk over gloo (one-GPU rehearsal)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "host_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),  # host time to enqueue a step (eager: ~900 launches; graph: one replay)
            "graph": bool(train_graph),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype_label,
            "data": "synthetic random RGBD / target amplitude+phase in [0,1), reference-style random-init weights",
            "config": {"workload": f"{args.rows}x{args.cols}x3 bs={B}/GPU generator+critic train step (d_ratio={args.d_ratio}, "
                                   f"lambda_gp=10, pad {args.pad} -> {args.rows + 2 * args.pad}^2 FFTs, 20-plane stack, "
                                   + (f"VGG19 perceptual term x{args.perceptual} with random weights" if args.perceptual > 0 else "no VGG term") + "), "
                                   + ((f"bf16 conv-GEMM operands, {hip_ops.activation_storage()} activation storage, fp32 accumulation / BatchNorm / FFT / Adam "
                                      "(informational)") if bf16 else f"fp32 tensors, conv GEMMs in the '{hip_ops.conv_precision()}' mode"),
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       # wire format of the gradient buckets' all-reduce (bf16 buckets are picked silently with bf16 storage)
                       "grad_payload": getattr(W._sync_G, "payload", "fp32") if world > 1 else None},
            "roofline": r,
        }

    # ---- the same step with the other fp32-tensor GEMM formulations (informational; a few steps each, every rank: collectives inside)
    other, by_mode = {}, {}

    def mode_roofline(peak, note):
        """gather-GEMM rate of the CURRENT mode from an instrumented pass of 2 steps (second stream on: the timed conditions)."""
        with native.kernel_profile() as pm:
            for _ in range(2):
                W.train_step(rgbd, tamp, tphs)
            torch.cuda.synchronize()
        blk = roofline_block(pm.result, 2, peak)
        return {"achieved": blk["achieved"], "peak": peak, "frac": blk["frac"], "unit": "TFLOP/s", "peak_note": note,
                "kernel_ms_per_step": blk["kernel_ms_per_step"], "wgrad_kernel": blk["wgrad_kernel"]}

    if args.mode == "train" and not bf16 and args.other_modes:
        for name in ("fp32", "fp32_split", "fp32_split2"):
            if name == mode:
                continue
            hip_ops.set_conv_precision(name)
            for _ in range(2):
                W.train_step(rgbd, tamp, tphs)
            sync()
            t0 = time.perf_counter()
            for _ in range(4):
                W.train_step(rgbd, tamp, tphs)
            sync()
            other[name] = round((time.perf_counter() - t0) / 4 * 1e3, 3)
            if name == "fp32":
                by_mode["fp32"] = dict(mode_roofline(FP32_MFMA_PEAK_TFLOPS, "dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32): the reference's arithmetic"),
                                       ms_per_step=other[name])
            elif name == "fp32_split":
                by_mode["fp32_split"] = dict(mode_roofline(round(BF16_MFMA_PEAK_TFLOPS / 6, 1), "dense bf16 MFMA peak / 6 products"), ms_per_step=other[name])
        hip_ops.set_conv_precision(mode)
        # bf16 NHWC activation storage (what `--dtype bf16` measures): a fresh trainer, the mode is process-wide
        del W
        torch.cuda.empty_cache()
        hip_ops.set_conv_precision("default")
        hip_ops.set_activation_storage("bf16")
        W = build_trainer()
        W.use_graph = train_graph  # this mode is launch-bound when launched eagerly (round 3: 21.7 ms of host work for a 23 ms step)
        for _ in range(12):  # the first ~10 steps after the switch are slower (allocator pool of the new tensor sizes, autotune)
            W.train_step(rgbd, tamp, tphs)
        sync()
        t0 = time.perf_counter()
        for _ in range(8):
            W.train_step(rgbd, tamp, tphs)
        sync()
        other["bf16_storage"] = round((time.perf_counter() - t0) / 8 * 1e3, 3)
        W.use_graph = False
        W.train_step(rgbd, tamp, tphs)
        by_mode["bf16_storage"] = dict(mode_roofline(BF16_MFMA_PEAK_TFLOPS, "dense bf16 MFMA peak, one product per multiply-add (bf16 operands AND bf16 activation storage: "
                                                     "informational, not the reference's precision)"), ms_per_step=other["bf16_storage"])
        hip_ops.set_activation_storage("fp32")
        if args.precision != "default":
            hip_ops.set_conv_precision(args.precision)
    if rank == 0 and by_mode:
        r0 = out["roofline"]
        by_mode[mode] = {"achieved": r0["achieved"], "peak": r0["peak"], "frac": r0["frac"], "unit": "TFLOP/s", "peak_note": r0["peak_note"],
                         "kernel_ms_per_step": r0["kernel_ms_per_step"], "wgrad_kernel": r0["wgrad_kernel"], "ms_per_step": out["ms_per_step"], "headline": True}
        out["roofline_by_mode"] = dict(by_mode, note="gather-GEMM (dominant kernel) rate of the same step per GEMM formulation, each priced against the matrix peak of ITS "
                                       "formulation, under the timed conditions (weight gradients concurrently on the second stream); the exact-fp32 step is the "
                                       "reference's arithmetic, the headline mode is fp32-faithful (tests/test_gpu_truth.py, profiles/r03_truth_*.json)")
    if rank == 0 and other:
        out["other_modes_ms_per_step"] = dict(other, note="fp32: exact fp32 MFMA (v_mfma_f32_32x32x2_f32).  fp32_split: three bf16 terms per operand, six MFMA products "
                                              "(round 2's first fp32-faithful formulation).  fp32_split2: two bf16 terms per operand, "
                                              "three MFMA products (max-rel error ~5e-6 per op instead of ~1e-6).  bf16_storage: bf16 NHWC activations "
                                              "and bf16 GEMM operands, fp32 accumulation/BN statistics/FFT/Adam (`--dtype bf16`).  None is the headline mode")

    # ---- informational: the step the reference's training CLI runs by default (trainingModel.py:77-95: five critic updates per generator
    # update, perceptual term 0.1 on VGG19 features — random-init weights here: no download), same data shape.  Not the headline config.
    cli = None
    if args.mode == "train" and not bf16 and world == 1 and args.cli_default:
        del W
        torch.cuda.empty_cache()
        keep = (args.perceptual, args.d_ratio)
        args.perceptual, args.d_ratio = 0.1, 5
        W = build_trainer()
        args.perceptual, args.d_ratio = keep
        for _ in range(2):
            W.train_step(rgbd, tamp, tphs)
        sync()
        t0 = time.perf_counter()
        for _ in range(3):
            W.train_step(rgbd, tamp, tphs)
        sync()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        cli = {"ms_per_step": round(ms, 3), "value": round(B / (ms / 1e3), 3), "unit": "frames/s",
               "config": {"workload": f"{args.rows}x{args.cols}x3 bs={B} train step as trainingModel.py configures it: d_ratio=5, perceptual_loss_weight=0.1 "
                                      "(VGG19 features, random-init weights), lambda_gp=10; informational"}}

    # ---- free the trainer, then north_star's second target (4K bs=1 inference + 8 planes), replicas only
    sec = None
    if args.secondary and not bf16:
        del W
        torch.cuda.empty_cache()
        sec = secondary_4k(native, dev, world, sync, mfma_peak, peak_note)
    if rank == 0:
        if cli is not None:
            out["reference_cli_default_step"] = cli
        if sec is not None:
            out["secondary"] = sec
        if world == 1 and args.cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
