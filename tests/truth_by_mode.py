"""Where does the GPU's distance to the float64 truth come from?  ("Next round" item 1.)

Runs BASELINE configs[1]'s generator + reconstruction forward (384x384, batch 4, train-mode BatchNorm, pad 320) in float64 on the
CPU (truth), in fp32 on the CPU (the reference's arithmetic: oracle/) and on the GPU once per conv-GEMM mode, stage by stage:

  unet      the UNet's 6 sigmoid outputs                       (convs + 18 train-mode BatchNorms)
  tail      POH / reconstructed amplitudes FROM THE TRUTH'S unet output  (optics + encode only: no conv GEMM involved)
  e2e       POH / hat_amps / target_amps of the whole forward

and writes e_gpu / e_cpu per quantity and mode to gpurun_out/r03_truth_by_mode.json (copied to profiles/ by hand).
Test infrastructure (lives under tests/: it imports oracle/; not collected by pytest).  Usage: python tests/truth_by_mode.py [rows] [batch]
"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from oracle import nets, optics, seeded  # noqa: E402

DEV = "cuda:0"
WL = torch.tensor([638e-9, 520e-9, 450e-9])


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max()).item()


def l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm()).item()


def phase_q(a, b, q=0.999):
    e = (torch.exp(1j * a.double()) - torch.exp(1j * b.double())).abs().flatten()[::7]
    return torch.quantile(e, q).item(), e.max().item()


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 384
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    pad, coef = (320 if rows == 384 else rows // 2), 0.45
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1]
    rgbd, tamp, tphs = seeded.smooth_batch(B, rows, rows, seed=51)
    idx = torch.tensor([17, 3, 11, 6][:B])
    o32 = optics.make_optics(rows, rows, pad, coef, 3.74e-6, WL)
    Hf32 = optics.transfer_function(o32.w, torch.tensor([1e-3]))[0]
    Hs32 = optics.transfer_function(o32.w, stack)

    def oracle(dtype, y_in=None):
        cdt = torch.complex128 if dtype == torch.float64 else torch.complex64
        o = optics.Optics(o32.rows0, o32.cols0, o32.pad_r, o32.pad_c, o32.rows, o32.cols, o32.w.to(dtype), o32.mask.to(dtype))
        sd = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in seeded.generator_state_dict().items()}
        with torch.no_grad():
            y = nets.unet(sd, "part1.part1.", rgbd.to(dtype), True) if y_in is None else y_in.to(dtype)
            poh = nets.amp_phase_to_poh(sd, o, Hf32.to(cdt), 1.1 * y[:, :3], 2 * torch.pi * y[:, 3:])
            hat_freq = optics.poh_to_filtered_spectrum(o, Hf32.to(cdt), poh)
            tgt_freq = optics.target_to_filtered_spectrum(o, tamp.to(dtype), tphs.to(dtype))
            amps, _ = optics.spectrum_to_planes_indexed(o, Hs32.to(cdt), torch.cat((hat_freq, tgt_freq), 0), idx)
        return dict(unet=y, POH=poh, hat_amps=amps[:B], target_amps=amps[B:])

    t0 = time.time()
    t64 = oracle(torch.float64)
    c32 = oracle(torch.float32)
    c32_tail = oracle(torch.float32, y_in=t64["unet"].float())  # fp32 tail fed with the truth's UNet output (rounded once)
    print(f"cpu oracles: {time.time() - t0:.1f} s", flush=True)

    from learned_hologram_gan_amd import hip_ops
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    out = {"config": dict(rows=rows, batch=B, pad=pad), "cpu_fp32": {}, "modes": {}}

    def score(got, ref64, tag, dst):
        for k in ("unet", "hat_amps", "target_amps"):
            if k in got:
                dst[f"{tag}.{k}.maxrel"] = rel(got[k], ref64[k])
                dst[f"{tag}.{k}.l2"] = l2(got[k], ref64[k])
        q, m = phase_q(got["POH"], ref64["POH"])
        dst[f"{tag}.POH.q999"], dst[f"{tag}.POH.max"] = q, m

    score(c32, t64, "e2e", out["cpu_fp32"])
    score({k: v for k, v in c32_tail.items() if k != "unet"}, t64, "tail", out["cpu_fp32"])

    for mode in ("fp32", "fp32_split", "fp32_split_f16"):
        hip_ops.set_conv_precision(mode)
        W = watermelon(filter_radius_coefficient=coef, pad_size=pad, distance_stack=stack, input_shape=(1, 4, rows, rows))
        W.generator.load_state_dict(seeded.generator_state_dict())
        W.generator.to(DEV).train()
        G = W.generator
        res = {}
        with torch.no_grad():
            y = G.part1.part1(rgbd.to(DEV))
            poh, hat_a, tgt_a, _, _ = W.reconstruct(rgbd.to(DEV), tamp.to(DEV), tphs.to(DEV), idx)
            score(dict(unet=y.cpu(), POH=poh.cpu(), hat_amps=hat_a.cpu(), target_amps=tgt_a.cpu()), t64, "e2e", res)
            # tail only: the truth's UNet output through the GPU's optics (no conv GEMM)
            y64 = t64["unet"].float().to(DEV)
            poh_t = G.part2(1.1 * y64[:, :3], 2 * torch.pi * y64[:, 3:])
            hat_t, _, tgt_t, _ = W.propagator.reconstruct_planes(G.part2.propagator, poh_t, tamp.to(DEV), tphs.to(DEV), idx)
            score(dict(POH=poh_t.cpu(), hat_amps=hat_t.cpu(), target_amps=tgt_t.cpu()), t64, "tail", res)
        torch.cuda.synchronize()
        out["modes"][mode] = res
        print(mode, json.dumps(res), flush=True)
    hip_ops.set_conv_precision("default")
    print("cpu_fp32", json.dumps(out["cpu_fp32"]))
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "r03_truth_by_mode.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
