"""Diagnostic for the two-process non-repeat: N identical train steps per process, `procs` processes sharing this GPU at the
same time (no torch.distributed, no collectives: each process is an ordinary single-GPU run).  Every step's outputs are compared bit for
bit with the process's first step; a differing step is recorded with the SHAPE of the damage of each tensor (planes, rows, columns, a few
value triples), so that the granularity can be matched against buffers and kernels.  One JSON line per process.

    python tools/two_process_probe.py [steps=40] [procs=2] [mode=step|forward|torch]      (GPU box; seeded weights need no oracle)
"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def damage(a, b):
    import torch

    d = (a - b).abs()
    nz = torch.nonzero(d > 0)
    if len(nz) == 0:
        return None
    out = {"elements": int(len(nz)), "max_abs": float(d.max()), "shape": list(a.shape)}
    if a.dim() == 4:
        out.update(planes=sorted({(int(u), int(v)) for u, v in nz[:, :2].tolist()})[:8], rows=sorted(set(nz[:, 2].tolist())),
                   cols=[int(nz[:, 3].min()), int(nz[:, 3].max()), len(set(nz[:, 3].tolist()))])
    flat = torch.nonzero((a - b).flatten() != 0).flatten()
    out["flat_first_last"] = [int(flat[0]), int(flat[-1])]
    out["flat_runs"] = int((flat[1:] - flat[:-1] != 1).sum()) + 1  # number of contiguous runs of differing elements
    if len(flat) <= 256:
        out["flat_all"] = flat.tolist()
    out["samples"] = [{"at": int(i), "this": float(a.flatten()[i]), "first": float(b.flatten()[i])} for i in flat[:: max(1, len(flat) // 5)][:5].tolist()]
    out["byte_offset_of_first"] = int(flat[0]) * a.element_size()
    out["data_ptr_mod_4096"] = int(a.data_ptr() % 4096)
    return out


def torch_only_worker(steps):
    """Control experiment: the same two-process set-up with stock ATen / MIOpen / rocFFT kernels only (no kernel of this repository):
    a small conv -> batch-norm -> relu -> conv -> fft2 -> filter -> ifft2 -> abs chain on tensors of the rig's sizes."""
    import torch
    import torch.nn.functional as F

    dev = "cuda:0"
    torch.manual_seed(5)
    x = torch.rand((2, 4, 64, 64), device=dev)
    ws = [torch.randn((64, 4, 3, 3), device=dev) * 0.2] + [torch.randn((64, 64, 3, 3), device=dev) * 0.05 for _ in range(6)] + [torch.randn((6, 64, 1, 1), device=dev) * 0.1]
    H = torch.exp(1j * torch.rand((3, 128, 128), device=dev) * 6.28)
    first, events = None, []
    for k in range(steps):
        with torch.no_grad():
            h = x
            for w in ws[:-1]:
                h = F.relu(F.batch_norm(F.conv2d(h, w, padding=1), None, None, training=True))
            y = torch.sigmoid(F.conv2d(h, ws[-1]))
            field = F.pad(y[:, :3] * torch.exp(1j * 6.28 * y[:, 3:]), (32, 32, 32, 32))
            z = torch.fft.ifft2(torch.fft.fft2(field) * H)[:, :, 32:-32, 32:-32]
            cur = {"y": y, "amp": z.abs(), "phs": z.angle()}
        torch.cuda.synchronize()
        cur = {n: t.detach().clone() for n, t in cur.items()}
        if first is None:
            first = cur
            continue
        bad = {n: dmg for n in cur if (dmg := damage(cur[n], first[n])) is not None}
        if bad:
            events.append({"step": k, "damage": bad})
    print(json.dumps({"pid": os.getpid(), "steps": steps, "mode": "torch", "events": len(events), "first_events": events[:4], "env": {}}), flush=True)


def kernel_family_worker(steps, mode):
    """Co-tenant that runs ONE family of this repository's kernels in a loop (which one does the co-tenant have to run for the damage
    to show in the other process?): conv = gather-GEMM forward 3x3 64->64, conv1 = 1x1, bn = statistics + apply, thin = 4->64 / 64->6,
    pool = max-pool + layout."""
    import torch

    sys.path.insert(0, REPO)
    from learned_hologram_gan_amd import hip_ops as ops

    dev = "cuda:0"
    torch.manual_seed(3)
    ci, co, hw = (int(v) for v in os.environ.get("PROBE_CONV", "64,64,64").split(","))  # PROBE_CONV=Ci,Co,extent of the conv co-tenant
    x = torch.randn((2, hw, hw, ci), device=dev)
    w3, w1 = torch.randn((co, ci, 3, 3), device=dev) * 0.05, torch.randn((co, ci, 1, 1), device=dev) * 0.1
    x4 = torch.randn((2, 64, 64, 32), device=dev)
    wt_in, wt_out = torch.randn((64, 4, 3, 3), device=dev) * 0.2, torch.randn((6, 64, 1, 1), device=dev) * 0.1
    gamma, beta, rm, rv = torch.ones(64, device=dev), torch.zeros(64, device=dev), torch.zeros(64, device=dev), torch.ones(64, device=dev)
    xn = torch.randn((2, 4, 64, 64), device=dev)
    first, events = None, 0
    with torch.no_grad():
        for k in range(steps):
            for j in range(20):
                mode_j = mode
                if mode.startswith("mix"):  # "mix:conv,bn,pool": cycle through the named families
                    fams = mode.split(":")[1].split(",")
                    mode_j = fams[j % len(fams)]
                if mode_j == "conv":
                    y = ops.conv2d_forward_raw(x, w3, None, 1)
                elif mode_j == "conv1":
                    y = ops.conv2d_forward_raw(x, w1, None, 1)
                elif mode_j == "bn":
                    y = ops.BatchNormTrainFn.apply(x[..., :64], gamma, beta, rm, rv, None, ops.ACT_RELU, 0.0, None)
                elif mode_j == "thin":
                    y = ops.conv2d_forward_raw(ops.conv2d_forward_raw(x4[..., :32], wt_in, None, 1), wt_out, None, 1, act=ops.ACT_SIGMOID, planar=True)
                else:
                    y = ops.MaxPool2x2Fn.apply(ops.ToNHWC.apply(xn, 32))
            torch.cuda.synchronize()
            c = y.float().double().sum().item()
            if first is None:
                first = c
            events += c != first
    print(json.dumps({"pid": os.getpid(), "steps": steps, "mode": mode, "events": int(events), "first_events": [], "env": {}}), flush=True)


def same_process_worker(steps):
    """ONE process, two HIP streams: the angular-spectrum operator in a loop on one stream, the deep 3x3 convolution (the strongest
    co-tenant of the two-process experiments) on the other — does the damage need two PROCESSES, or only two queues?"""
    import torch

    sys.path.insert(0, REPO)
    from learned_hologram_gan_amd import hip_ops as ops
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    dev = "cuda:0"
    torch.manual_seed(1234)
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
    W = watermelon(filter_radius_coefficient=0.45, pad_size=32, distance_stack=stack, input_shape=(1, 4, 64, 64))
    W.generator.to(dev).train()
    g = torch.Generator().manual_seed(7)
    x0 = torch.rand((2, 4, 64, 64), generator=g).to(dev)
    ci, co, hw = (int(v) for v in os.environ.get("PROBE_CONV", "1024,512,8").split(","))
    xc, wc = torch.randn((2, hw, hw, ci), device=dev), torch.randn((co, ci, 3, 3), device=dev) * 0.05
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.no_grad():
        poh0 = W.generator(x0)
        fp = W.generator.part2.propagator
        first = fp.propagate_POH2AP_forward(poh0)[0].clone()
        for _ in range(3):
            ops.conv2d_forward_raw(xc, wc, None, 1)  # tuning / forced variant warm-up on the default stream
        torch.cuda.synchronize()
        events = 0
        for k in range(steps):
            with torch.cuda.stream(sb):
                for _ in range(6):
                    yc = ops.conv2d_forward_raw(xc, wc, None, 1)
            with torch.cuda.stream(sa):
                outs = [fp.propagate_POH2AP_forward(poh0)[0] for _ in range(8)]
            torch.cuda.synchronize()
            events += sum(int(not torch.equal(o, first)) for o in outs)
    print(json.dumps({"pid": os.getpid(), "steps": steps * 8, "mode": "same_process(asm1|conv)", "events": events, "first_events": [], "env": {}}), flush=True)


def worker(steps, mode):
    import torch

    if mode == "same":
        return same_process_worker(steps)
    if mode == "torch":
        return torch_only_worker(steps)
    if mode in ("conv", "conv1", "bn", "thin", "pool") or mode.startswith("mix"):
        return kernel_family_worker(steps, mode)
    sys.path.insert(0, REPO)
    from learned_hologram_gan_amd import hip_ops
    from learned_hologram_gan_amd.watermelon_hologram.watermelon import watermelon

    dev = "cuda:0"
    rows, pad = (int(v) for v in os.environ.get("PROBE_GEOM", "64,32").split(","))  # PROBE_GEOM=rows,pad: 64,32 -> 128 (LDS Stockham passes); 192,32 -> 256 (register passes)
    cols = rows
    torch.manual_seed(1234)  # both processes: the same weights and data (compared within a process only)
    stack = torch.linspace(-4e-4, 0.0, 21)[:-1][:8]
    W = watermelon(filter_radius_coefficient=0.45, pad_size=pad, distance_stack=stack, input_shape=(1, 4, rows, cols))
    W.generator.to(dev).train()
    W.discriminator.to(dev).train()
    W.configure(1, 0.0, 1, 1e-3, 0.1, 1e-3, 1e-3, 1 if mode == "step" else 0, 10)
    g = torch.Generator().manual_seed(7)
    x = [torch.rand((2, c, rows, cols), generator=g).to(dev) for c in (4, 3, 3)]
    idx = torch.tensor([5, 2])
    alphas = [torch.tensor([0.3, 0.8]).view(2, 1, 1, 1).to(dev)]
    grads = {}
    for name, opt in (("D", W._opt_D), ("G", W._opt_G)):
        if opt is None:
            continue

        def no_update(grad_scale=1.0, name=name, opt=opt):  # capture instead of Adam: every step sees the same weights
            hip_ops.join_side_stream()
            grads[name] = opt.flat.grad.detach().clone()
        opt.step = no_update
    first, events = None, []
    if mode in ("asm", "unet", "asm1", "asmto", "asmfrom"):  # one half of the forward only: the angular-spectrum operators on fixed inputs / the UNet alone
        with torch.no_grad():
            poh0 = W.generator(x[0])
        for k in range(steps):
            with torch.no_grad():
                if mode == "asm":
                    r = W.propagator.reconstruct_planes(W.generator.part2.propagator, poh0, x[1], x[2], idx)
                    cur = {"hat_amps": r[0], "target_amps": r[2]}
                elif mode in ("asmto", "asmfrom"):  # the halves of the operator: row pass + column FFT x filter / column IFFT + row pass
                    fp = W.generator.part2.propagator
                    if mode == "asmto":
                        cur = {"spectrum": torch.view_as_real(fp.propagate_POH2Freq_forward(poh0))}
                    else:
                        if first is None:
                            S_fixed = fp.propagate_POH2Freq_forward(poh0)
                        r = W.propagator.propagate_multiple_samples_with_random_fixed_multiple_distances_freq2amp(S_fixed, idx)
                        cur = {"amp": r[0], "phs": r[1]}
                elif mode == "asm1":  # the SAME operator call twice per iteration: its workspace block only ever holds one content
                    fp = W.generator.part2.propagator
                    r = fp.propagate_POH2AP_forward(poh0)
                    r2 = fp.propagate_POH2AP_forward(poh0)
                    cur = {"amp": r[0], "amp_again": r2[0]}
                else:
                    cur = {"unet": W.generator.part1.part1(x[0])}
            torch.cuda.synchronize()
            cur = {n: t.detach().clone() for n, t in cur.items()}
            if first is None:
                first = cur
                continue
            bad = {n: dmg for n in cur if (dmg := damage(cur[n], first[n])) is not None}
            if bad:
                events.append({"step": k, "damage": bad})
        print(json.dumps({"pid": os.getpid(), "steps": steps, "mode": mode, "events": len(events), "first_events": events[:4], "env": {}}), flush=True)
        return
    for k in range(steps):
        if mode == "forward":
            with torch.no_grad():
                poh, hat_a, tgt_a, hat_p, tgt_p = W.reconstruct(x[0], x[1], x[2], idx)
            cur = {"POH": poh, "hat_amps": hat_a, "target_amps": tgt_a, "hat_phs": hat_p, "tgt_phs": tgt_p}
        else:
            out = W.train_step(x[0], x[1], x[2], idx, alphas)
            cur = {"POH": out["POH"], "hat_amps": out["hat_amps"], "target_amps": out["target_amps"], "grad_G": grads["G"], "grad_D": grads.get("D", grads["G"])}
        torch.cuda.synchronize()
        cur = {n: t.detach().clone() for n, t in cur.items()}
        if first is None:
            first = cur
            continue
        bad = {n: dmg for n in cur if (dmg := damage(cur[n], first[n])) is not None}
        if bad:
            events.append({"step": k, "damage": {n: v for n, v in bad.items()}})
    print(json.dumps({"pid": os.getpid(), "steps": steps, "mode": mode, "events": len(events), "first_events": events[:4],
                      "env": {k: os.environ[k] for k in ("AMD_SERIALIZE_KERNEL", "LHG_SIDE_WGRAD", "LHG_AUTOTUNE", "HIP_LAUNCH_BLOCKING") if k in os.environ}}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), sys.argv[3])
    else:
        steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
        procs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
        mode = sys.argv[3] if len(sys.argv) > 3 else "step"
        modes = mode.split("+")  # "forward+torch": process i runs modes[i % len(modes)] (who must be the co-tenant for the damage to show?)
        mult = {"torch": 3, "asm": 8, "asm1": 8, "asmto": 8, "asmfrom": 8, "unet": 1, "conv": 2, "conv1": 2, "bn": 2, "thin": 2, "pool": 2, "mix": 2}
        ps = []
        for i in range(procs):  # "asm@HIP_LAUNCH_BLOCKING=1+unet": VAR=value pairs after '@' go into that process's environment only
            m, *envs = modes[i % len(modes)].split("@")
            env = dict(os.environ, **dict(e.split("=", 1) for e in envs))
            ps.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(steps * mult.get(m.split(":")[0], 1)), m], stdout=subprocess.PIPE, text=True, env=env))
        rc = 0
        for p_ in ps:
            out, _ = p_.communicate(timeout=900)
            rc |= p_.returncode
            for ln in out.splitlines():
                if ln.startswith("{"):
                    print(ln, flush=True)
        sys.exit(rc)
