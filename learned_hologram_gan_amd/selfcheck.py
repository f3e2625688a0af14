"""smoke_check(): one small generator forward + one-plane propagation on the GPU, compared with
the CPU oracle.  The oracle is imported here as the CHECKER only (see oracle/__init__.py)."""

from __future__ import annotations

import torch


def smoke_check(device="cuda:0", rows=64, cols=64, pad=32, batch=1):
    from oracle import nets, optics, seeded  # checker only

    from .watermelon_hologram.generator import Generator

    torch.cuda.set_device(device)
    wl = torch.tensor([638e-9, 520e-9, 450e-9])
    sd = seeded.generator_state_dict()
    rgbd, _, _ = seeded.smooth_batch(batch, rows, cols, seed=3)
    G = Generator(rows, cols, pad, 0.45, 3, 3.74e-6, wl, torch.tensor([1e-3]))
    G.load_state_dict(sd)
    G.to(device).eval()
    with torch.no_grad():
        poh = G(rgbd.to(device))
        amp, _ = G.part2.propagator.propagate_POH2AP_forward(poh)
    torch.cuda.synchronize()
    o = optics.make_optics(rows, cols, pad, 0.45, 3.74e-6, wl)
    Hf = optics.transfer_function(o.w, torch.tensor([1e-3]))[0]
    with torch.no_grad():
        poh_ref = nets.generator(nets.as_parameters(sd), o, Hf, rgbd, False)
        amp_ref, _ = optics.poh_to_amp_phase(o, Hf, poh_ref)
    e_poh = (torch.exp(1j * poh.cpu()) - torch.exp(1j * poh_ref)).abs().max().item()
    e_amp = ((amp.cpu() - amp_ref).abs().max() / amp_ref.abs().max()).item()
    assert e_poh < 5e-3 and e_amp < 1e-3, f"smoke mismatch vs oracle: POH {e_poh:.3e}, amplitude {e_amp:.3e}"
    return dict(poh_phase_err=e_poh, amp_rel_err=e_amp, shape=tuple(poh.shape))
