#!/usr/bin/env python3
"""Benchmark of the RGBD -> POH hot path on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one batch of the GAN training loop of the reference (watermelon.py:207-277): generator
forward, fused angular-spectrum reconstruction of hat/target at one random plane per sample,
`d_ratio` critic updates with the gradient penalty (double backward), generator loss + backward,
both Adam updates — on BASELINE.json configs[1]: 384x384x3, batch 4 per GPU, fp32, synthetic
random-RGBD inputs resident in HBM, reference-style random-init weights.  Weak scaling: every rank
runs the same per-GPU batch and gradients are averaged over RCCL.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the MFMA gather-GEMM behind
every convolution / input-gradient): achieved = algorithmic conv FLOPs it executed / its summed
launch duration, both measured live over the timed region with HIP events on the launch stream.
`cpu_baseline` times the CPU oracle (a port of the reference step, oracle/step.py) on the host
cores for a bounded sample (one frame).
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)


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
    ap.add_argument("--graph", type=int, default=0, help="infer mode: replay a captured hipGraph instead of eager launches")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32",
                    help="f32 (the benchmark metric): exact fp32 MFMA.  bf16 (BASELINE configs[2]/[4], informational): conv GEMM operands "
                         "rounded to bf16, fp32 accumulation, fp32 tensors")
    ap.add_argument("--perceptual", type=float, default=0.0,
                    help="weight of the VGG19 perceptual term (reference CLI: 0.1; seeded random VGG weights: throughput only).  The benchmark "
                         "metric follows BASELINE.md's assembled step, which has no VGG term")
    ap.add_argument("--planes", type=int, default=0,
                    help="infer mode: also propagate the hologram to this many planes (BASELINE configs[3]: inference + multi-plane propagate)")
    ap.add_argument("--mode", choices=("train", "infer"), default="train",
                    help="train: the GAN step (the benchmark metric); infer: eval-mode generator forward RGBD->POH only (informational)")
    return ap.parse_args()


def cpu_baseline(args):
    """One frame of the same step through the CPU oracle (checker code timed as the reported baseline)."""
    from oracle import seeded, step

    # the GPU box gives one GPU a 16-CPU share although it reports every core of the host
    threads = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    torch.set_num_threads(threads)
    rows = args.cpu_rows or args.rows
    cols = args.cpu_rows or args.cols
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1]
    st = step.make_state(rows, cols, args.pad, 0.45, stack, seeded.generator_state_dict(), seeded.critic_state_dict())
    rgbd, amp, phs = seeded.synthetic_batch(1, rows, cols)
    w = step.LossWeights(d_ratio=args.d_ratio)
    t0 = time.perf_counter()
    step.train_step(st, rgbd, amp, phs, w, torch.tensor([7]), [torch.full((1, 1, 1, 1), 0.5) for _ in range(max(args.d_ratio, 1))])
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 5), "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 frame {rows}x{cols} (batch 1), one full train step (G fwd+bwd, {args.d_ratio} critic update(s) with gradient "
                      f"penalty, Adam x2) through oracle/step.py, {dt:.1f} s"}


def main():
    args = parse()
    from learned_hologram_gan_amd import distributed, native
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    from learned_hologram_gan_amd import hip_ops

    rank, world, local = distributed.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(122731 + rank)

    if args.dtype == "bf16":
        hip_ops.set_conv_precision("bf16")
    bf16 = args.dtype == "bf16"
    mfma_peak = 2500.0 if bf16 else FP32_MFMA_PEAK_TFLOPS  # dense bf16 MFMA peak (MI355X_MICROARCH.md) / fp32 MFMA peak
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1]  # trainingModel.py:62
    perceptual = None
    if args.perceptual > 0:
        import warnings

        from learned_hologram_gan_amd.watermelon_hologram.perceptual import perceptualLoss

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            perceptual = perceptualLoss()
    W = watermelon(filter_radius_coefficient=0.45, pad_size=args.pad, distance_stack=stack, input_shape=(1, 4, args.rows, args.cols),
                   perceptual_loss=perceptual)
    W.generator.to(dev).train()
    W.discriminator.to(dev).train()
    W.configure(phs_gradient_loss_weight=1, perceptual_loss_weight=args.perceptual, pixel_loss_weight=1, TV_loss_weight=1e-3,
                discriminator_loss_weight=1e-1, lr_G=1e-3, lr_D=1e-3, discriminator_train_ratio=args.d_ratio, discriminator_lambda=10)
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
        if args.graph:
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
                                                     + (", hipGraph replay" if args.graph else "")}}))
        return

    for _ in range(args.warmup):
        W.train_step(rgbd, tamp, tphs)
    sync()
    with native.kernel_profile() as prof:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            W.train_step(rgbd, tamp, tphs)
        sync()
        elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = t.item()

    # The weight-gradient GEMMs run on a second stream, concurrently with the gather-GEMMs and the HBM-bound kernels of the main
    # stream, so the per-launch durations measured above include time spent sharing the GPU.  A few extra steps with that overlap
    # switched off (outside the timed region) give the kernels' own rates, reported as roofline["isolated"].
    iso = None
    if hip_ops.SIDE_WGRAD:  # every rank: the steps contain collectives
        hip_ops.SIDE_WGRAD = False
        W.train_step(rgbd, tamp, tphs)
        torch.cuda.synchronize()
        with native.kernel_profile() as prof_iso:
            for _ in range(min(args.steps, 5)):
                W.train_step(rgbd, tamp, tphs)
            torch.cuda.synchronize()
        hip_ops.SIDE_WGRAD = True
        iso = prof_iso.result

    if rank == 0:
        gg, wg = prof.result
        traffic = None  # HBM bytes per gather-GEMM launch from the committed PMC passes (tools/pmc_traffic.py)
        try:
            with open(os.path.join(REPO, "profiles", "r01_pmc_traffic_v3.json")) as f:
                traffic = json.load(f)["kernels"]["gg"]["hbm_bytes_per_launch_corrected"]
        except (OSError, KeyError, ValueError):
            pass
        achieved = gg["algorithmic_flops"] / (gg["total_ms"] * 1e-3) / 1e12 if gg["total_ms"] > 0 else 0.0
        out = {
            "metric": "RGBD->POH frames/sec at 384x384 bs=4 (GAN train step)",
            "value": round(B * world * args.steps / elapsed, 4),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic random RGBD / target amplitude+phase in [0,1), reference-style random-init weights",
            "config": {"workload": f"{args.rows}x{args.cols}x3 bs={B}/GPU generator+critic train step (d_ratio={args.d_ratio}, "
                                   f"lambda_gp=10, pad {args.pad} -> {args.rows + 2 * args.pad}^2 FFTs, 20-plane stack, "
                                   + (f"VGG19 perceptual term x{args.perceptual} with random weights" if args.perceptual > 0 else "no VGG term") + "), "
                                   + ("bf16 conv-GEMM operands, fp32 accumulation / tensors / FFT (informational)" if bf16 else "fp32"),
                       "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": {
                "kernel": "lhg::gg2b_kernel (MFMA bf16 gather-GEMM)" if bf16 else
                          "lhg::gg_kernel (MFMA fp32 gather-GEMM: conv forward / input-gradient / conv-transpose)",
                "bound": "mfma", "achieved": round(achieved, 3), "peak": mfma_peak, "unit": "TFLOP/s",
                "frac": round(achieved / mfma_peak, 4), "traffic": None if bf16 else traffic,
                "algorithmic_flop_per_launch": round(gg["algorithmic_flops"] / max(gg["launches"], 1)),
                "launches_per_step": gg["launches"] / args.steps,
                "avg_launch_us": round(gg["total_ms"] * 1e3 / max(gg["launches"], 1), 2),
                "algorithmic_gflop_per_step": round(gg["algorithmic_flops"] / args.steps / 1e9, 2),
                "executed_gflop_per_step": round(gg["executed_flops"] / args.steps / 1e9, 2),
                "kernel_ms_per_step": round(gg["total_ms"] / args.steps, 3),
                "wgrad_kernel": {"achieved": round(wg["algorithmic_flops"] / max(wg["total_ms"], 1e-9) / 1e9, 3), "unit": "TFLOP/s",
                                 "kernel_ms_per_step": round(wg["total_ms"] / args.steps, 3),
                                 "algorithmic_gflop_per_step": round(wg["algorithmic_flops"] / args.steps / 1e9, 2)},
            },
        }
        if iso is not None:
            # headline roofline = the kernel's own rate (second stream off: durations are not shared with a concurrent GEMM; this is also
            # what rocprofv3 --kernel-trace reports, since it serialises dispatches); the figures of the timed region go underneath
            ig, iw = iso
            k = min(args.steps, 5)
            ia = ig["algorithmic_flops"] / (ig["total_ms"] * 1e-3) / 1e12 if ig["total_ms"] > 0 else 0.0
            r = out["roofline"]
            r["timed_region"] = {"achieved": r["achieved"], "frac": r["frac"], "avg_launch_us": r["avg_launch_us"],
                                 "kernel_ms_per_step": r["kernel_ms_per_step"], "wgrad_kernel_achieved": r["wgrad_kernel"]["achieved"],
                                 "note": "weight-gradient GEMMs run on a second HIP stream concurrently with this kernel: launch durations "
                                         "include time sharing the GPU"}
            r.update({"achieved": round(ia, 3), "frac": round(ia / mfma_peak, 4),
                      "avg_launch_us": round(ig["total_ms"] * 1e3 / max(ig["launches"], 1), 2),
                      "kernel_ms_per_step": round(ig["total_ms"] / k, 3),
                      "measured": f"HIP events around every launch over {k} steps of the same workload right after the timed region, second stream "
                                  "off (kernel's own duration)"})
            r["wgrad_kernel"] = {"achieved": round(iw["algorithmic_flops"] / max(iw["total_ms"], 1e-9) / 1e9, 3), "unit": "TFLOP/s",
                                 "kernel_ms_per_step": round(iw["total_ms"] / k, 3),
                                 "algorithmic_gflop_per_step": round(iw["algorithmic_flops"] / k / 1e9, 2)}
        if bf16:
            out["metric"] += " [bf16 operand mode, informational]"
        if world == 1 and args.cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
