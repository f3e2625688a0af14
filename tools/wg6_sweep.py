"""Time the tap-fused weight-gradient kernel (csrc/wg6_kernel.inc) per layer of the 384^2 batch-4 train step: every tile variant that fits
x a few split counts x {separate reduce launch, in-launch reduction}, next to the library's own plan and the per-tap kernels (LHG_WG6=0 in a
second process).  Random operands (the expensive case for the matrix pipe's clock).

    python tools/wg6_sweep.py [--quick] [--json out.jsonl]
"""
import argparse
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

# (name, N, Ci, Co, H, W (of x), k, stride, launches per step) — the MFMA weight gradients of one train step
LAYERS = [
    ("G.enc1.c2 64>64@384", 4, 64, 64, 384, 384, 3, 1, 2),
    ("G.dec4.c1 128>64@384", 4, 128, 64, 384, 384, 3, 1, 1),
    ("G.enc2.c1 64>128@192", 4, 64, 128, 192, 192, 3, 1, 5),
    ("G 128>128@192", 4, 128, 128, 192, 192, 3, 1, 2),
    ("G.dec3.c1 256>128@192", 4, 256, 128, 192, 192, 3, 1, 1),
    ("G.enc3.c1 128>256@96", 4, 128, 256, 96, 96, 3, 1, 1),
    ("G 256>256@96", 4, 256, 256, 96, 96, 3, 1, 2),
    ("G.dec2.c1 512>256@96 / D.b5 256>512@96", 4, 256, 512, 96, 96, 3, 1, 5),
    ("G.enc4.c1 256>512@48", 4, 256, 512, 48, 48, 3, 1, 1),
    ("G 512>512@48", 4, 512, 512, 48, 48, 3, 1, 2),
    ("G.dec1.c1 1024>512@48", 4, 1024, 512, 48, 48, 3, 1, 1),
    ("G.bott.c1 512>1024@24", 4, 512, 1024, 24, 24, 3, 1, 1),
    ("G.bott.c2 1024>1024@24", 4, 1024, 1024, 24, 24, 3, 1, 1),
    ("D.b2 32>64 s2 @384", 4, 32, 64, 384, 384, 3, 2, 4),
    ("D.b4 128>256 s2 @192", 4, 128, 256, 192, 192, 3, 2, 4),
    ("D.b6 512>1024 s2 @96", 4, 512, 1024, 96, 96, 3, 2, 4),
    ("1x1 64>128@192", 4, 64, 128, 192, 192, 1, 1, 1),
    ("1x1 128>64@384", 4, 128, 64, 384, 384, 1, 1, 1),
    ("1x1 512>256@96", 4, 512, 256, 96, 96, 1, 1, 1),
    ("1x1 1024>512@48", 4, 1024, 512, 48, 48, 1, 1, 1),
]
CONVT = [("convT 1024>512 @24", 4, 1024, 512, 24, 24), ("convT 512>256 @48", 4, 512, 256, 48, 48), ("convT 256>128 @96", 4, 256, 128, 96, 96),
         ("convT 128>64 @192", 4, 128, 64, 192, 192)]


def time_call(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--json", default=None)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--only-convt", action="store_true")
    args = ap.parse_args()
    from learned_hologram_gan_amd import hip_ops as ops
    from learned_hologram_gan_amd import native

    lib = native.load()
    dev = "cuda:0"
    ops.set_conv_precision("fp32_split_f16")
    names = [lib.lhg_wg6_variant_name(v).decode() for v in range(lib.lhg_wg6_variants())]
    out = open(args.json, "a") if args.json else None
    legacy = os.environ.get("LHG_WG6", "1") == "0"
    total_plan = 0.0
    import ctypes

    for (name, N, Ci, Co, H, W, k, stride, count) in ([] if args.only_convt else LAYERS):
        g = torch.Generator().manual_seed(1)
        x = torch.rand((N, H, W, Ci), generator=g).to(dev) * 2 - 1
        Ho, Wo = (H + 2 * (k // 2) - k) // stride + 1, (W + 2 * (k // 2) - k) // stride + 1
        gy = torch.rand((N, Ho, Wo, Co), generator=g).to(dev) * 2 - 1
        flops = 2.0 * N * Ho * Wo * Ci * Co * k * k
        run = lambda: ops.conv2d_weight_grad_raw(x, gy, (Co, Ci, k, k), stride)  # noqa: E731
        lib.lhg_wg6_force(-1, -1, -1)
        with torch.no_grad():
            us = time_call(run, args.reps)
        total_plan += us * count
        pv, ps, pf = ctypes.c_int(-1), ctypes.c_int(0), ctypes.c_int(0)
        lib.lhg_wg6_last_plan(ctypes.byref(pv), ctypes.byref(ps), ctypes.byref(pf))
        chosen = "" if legacy else f" [v{pv.value} {names[pv.value] if pv.value >= 0 else '-'} S={ps.value} fused={pf.value}]"
        print(f"{name:42s} {'per-tap kernels' if legacy else 'plan'}: {us:8.1f} us {flops / us / 1e6:7.1f} TFLOP/s   x{count}{chosen}", flush=True)
        rec = {"layer": name, "legacy": legacy, "plan_us": us, "plan": [pv.value, ps.value, pf.value], "flops": flops, "count": count, "variants": []}
        if not legacy and not args.quick:
            pad = lambda c: (c + 63) // 64 * 64  # noqa: E731
            steps = (N * Ho * (Wo + 2) + 31) // 32
            for v, vn in enumerate(names):
                tile, cw, t, ny, s = vn.split()
                bm, bn = map(int, tile.split("x"))
                if int(t[2:]) != k or int(s[1:]) != stride or pad(Ci) % bm or pad(Co) % bn or k % int(ny[2:]):
                    continue
                tiles = (pad(Ci) // bm) * (pad(Co) // bn) * (k // int(ny[2:]))
                cands = sorted({max(1, min(steps // 8, s_)) for s_ in (1, max(1, 128 // tiles), max(1, 256 // tiles), max(1, 384 // tiles), max(1, 512 // tiles))})
                for S in cands:
                    for fused in (0, 1):
                        lib.lhg_wg6_force(v, S, fused)
                        with torch.no_grad():
                            u = time_call(run, args.reps)
                        print(f"    v{v:2d} {vn:26s} S={S:3d} fused={fused}: {u:8.1f} us {flops / u / 1e6:7.1f} TFLOP/s", flush=True)
                        rec["variants"].append({"v": v, "name": vn, "S": S, "fused": fused, "us": u})
            lib.lhg_wg6_force(-1, -1, -1)
        if out:
            out.write(json.dumps(rec) + "\n")
            out.flush()
    for (name, N, Ci, Co, H, W) in CONVT:
        g = torch.Generator().manual_seed(2)
        x = (torch.rand((N, H, W, Ci), generator=g).to(dev) * 2 - 1).requires_grad_(True)
        w = torch.zeros((Ci, Co, 2, 2), device=dev, requires_grad=True)
        gy = torch.rand((N, 2 * H, 2 * W, Co), generator=g).to(dev) * 2 - 1
        flops = 2.0 * N * H * W * 4 * Ci * Co
        ops.SIDE_WGRAD = False
        y = ops.ConvTranspose2x2Fn.apply(x.detach(), w, None, None)

        def run():
            w.grad = None
            y.backward(gy, retain_graph=True, inputs=[w])

        lib.lhg_wg6_force(-1, -1, -1)
        us = time_call(run, args.reps)
        total_plan += us
        pv, ps, pf = ctypes.c_int(-1), ctypes.c_int(0), ctypes.c_int(0)
        lib.lhg_wg6_last_plan(ctypes.byref(pv), ctypes.byref(ps), ctypes.byref(pf))
        print(f"{name:42s} backward incl. autograd: {us:8.1f} us ({flops / us / 1e6:6.1f} TFLOP/s incl. overhead) [v{pv.value} S={ps.value} fused={pf.value}]", flush=True)
        if not legacy and not args.quick:
            pad = lambda c: (c + 63) // 64 * 64  # noqa: E731
            steps = (N * H * (W + 2) + 31) // 32
            for v, vn in enumerate(names):
                tile, cw, t, ny, st_ = vn.split()
                bm, bn = map(int, tile.split("x"))
                if t != "nt2" or pad(Co) % bm or pad(Ci) % bn:  # strip operand = gy (Co channels), point operand = x (Ci)
                    continue
                tiles = (pad(Co) // bm) * (pad(Ci) // bn) * 2
                for S in sorted({max(1, min(steps // 8, s_)) for s_ in (1, max(1, 128 // tiles), max(1, 256 // tiles), max(1, 512 // tiles))}):
                    lib.lhg_wg6_force(v, S, 1 if S == 1 else 0)
                    u = time_call(run, args.reps)
                    print(f"    v{v:2d} {vn:26s} S={S:3d}: {u:8.1f} us", flush=True)
            lib.lhg_wg6_force(-1, -1, -1)
    print(f"sum over the step's launches: {total_plan / 1e3:.2f} ms")


if __name__ == "__main__":
    main()
