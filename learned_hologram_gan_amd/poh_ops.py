"""Autograd op for the AP2POH tail (C ABI: lhg_symconv_field / lhg_double_phase_encode and their backward kernels).

    POH = double_phase( normalise( |stencil(field)| ), angle(stencil(field)) )

ref: watermelon_hologram/AP2POH.py:105-116 (forward), utilities.py:53-66 (amplitude_normalizor),
neural_network_components.py:68-95 (SymmetricConv2d / ChannelWiseSymmetricConv).
"""

from __future__ import annotations

import torch
from torch.autograd import Function

from . import native
from .native import call, ptr, stream_ptr


class PohEncodeFn(Function):
    """field (B,3,H,W) complex64, taps (3,3) [colour][centre, edge, corner], bias (3,) -> POH (B,3,H,W) fp32."""

    @staticmethod
    def forward(ctx, field, taps, bias):
        if field.dtype != torch.complex64 or field.dim() != 4 or field.shape[1] != 3:
            raise ValueError(f"PohEncodeFn expects a (B,3,H,W) complex64 field, got {field.dtype} {tuple(field.shape)}")
        B, Cc, H, W = field.shape
        field = field.contiguous()
        taps = taps.detach().contiguous()
        bias = bias.detach().contiguous()
        mod = torch.empty_like(field)
        peak = torch.zeros((B * Cc,), dtype=torch.int64, device=field.device)
        poh = torch.empty((B, Cc, H, W), dtype=torch.float32, device=field.device)
        call("lhg_symconv_field", ptr(torch.view_as_real(field)), B * Cc, H, W, ptr(taps), ptr(bias), ptr(torch.view_as_real(mod)),
             ptr(peak), stream_ptr())
        call("lhg_double_phase_encode", ptr(torch.view_as_real(mod)), ptr(peak), B * Cc, H, W, ptr(poh), stream_ptr())
        ctx.save_for_backward(field, taps, mod, peak)
        return poh

    @staticmethod
    def backward(ctx, g_poh):
        field, taps, mod, peak = ctx.saved_tensors
        B, Cc, H, W = field.shape
        planes = B * Cc
        nblk = native.load().lhg_poh_partial_blocks(H, W)
        g_poh = g_poh.contiguous()  # bound to a local: a temporary's block could be reused by the allocations below
        g_mod = torch.empty_like(field)
        ws = torch.empty((planes * nblk,), dtype=torch.float32, device=field.device)
        call("lhg_double_phase_encode_backward", ptr(g_poh), ptr(torch.view_as_real(mod)), ptr(peak), planes, H, W,
             ptr(torch.view_as_real(g_mod)), ptr(ws), stream_ptr())
        g_field = torch.empty_like(field)
        partial = torch.empty((B, Cc, nblk, 4), dtype=torch.float32, device=field.device)
        call("lhg_symconv_field_backward", ptr(torch.view_as_real(g_mod)), ptr(torch.view_as_real(field)), planes, H, W, ptr(taps),
             ptr(torch.view_as_real(g_field)), ptr(partial), stream_ptr())
        sums = partial.sum(dim=(0, 2))  # (3 colours, 4): 12 numbers
        return g_field, sums[:, :3].contiguous(), sums[:, 3].contiguous()


class ReconLossFn(Function):
    """(focal phase-gradient, pixel MSE, TV difference) of (hat_amp, target_amp, hat_phase, target_phase), each (B,3,H,W).
    One fused forward kernel (+ a 9-value final reduction) and one fused backward kernel (C ABI: lhg_recon_loss_*).
    ref: loss_func.py:66-98, 135-163; watermelon.py:418-445."""

    @staticmethod
    def forward(ctx, hat_amp, tgt_amp, hat_phs, tgt_phs):
        ha, ta, hp, tp = (t.contiguous() for t in (hat_amp, tgt_amp, hat_phs, tgt_phs))
        B, Cc, H, W = ha.shape
        planes = B * Cc
        nblk = native.load().lhg_recon_loss_blocks(planes, H, W)
        sums = torch.empty((9,), dtype=torch.float32, device=ha.device)
        losses = torch.empty((3,), dtype=torch.float32, device=ha.device)
        ws = torch.empty((nblk * 9,), dtype=torch.float32, device=ha.device)
        call("lhg_recon_loss_forward", ptr(ha), ptr(ta), ptr(hp), ptr(tp), planes, H, W, ptr(sums), ptr(losses), ptr(ws), stream_ptr())
        from . import hip_ops

        world = hip_ops.sync_world()
        if world > 1:
            # Global-batch normalisers (hip_ops.set_sync_batch_stats): the focal loss divides by the WHOLE batch's max |difference| (loss_func.py:
            # 152-157) and the TV term is |TV(hat) - TV(target)| of whole-batch means (:94-98).  Per replica: local sums over the global max, and
            # sign(global TV difference) x the local TV difference — their mean over the replicas is the single-device loss, and so are the
            # gradients after the gradient all-reduce.  sums[1], sums[3] <- global max; sums[5:9] <- global means (the backward kernel takes
            # the sign from them); the reported TV value is the global one.
            mx = sums[[1, 3]].contiguous()
            hip_ops.all_reduce_(mx, "max")
            a_w, a_h = float(planes * H * (W - 1)), float(planes * (H - 1) * W)
            tvg = sums[5:9].clone()
            hip_ops.all_reduce_(tvg).div_(world)
            sums[1], sums[3] = mx[0], mx[1]
            sums[5:9] = tvg
            s64 = sums.double()
            n_w, n_h = 2.0 * a_w, 2.0 * a_h
            tv_global = (s64[5] / a_w + s64[6] / a_h) - (s64[7] / a_w + s64[8] / a_h)
            losses = torch.stack((s64[0] / (n_w * s64[1]) + s64[2] / (n_h * s64[3]), s64[4] / float(planes * H * W), tv_global.abs())).float()
        ctx.save_for_backward(ha, ta, hp, tp, sums)
        return losses

    @staticmethod
    def backward(ctx, g):
        ha, ta, hp, tp, sums = ctx.saved_tensors
        B, Cc, H, W = ha.shape
        g = g.contiguous().float()  # bound to a local before the allocations below
        g_ha, g_hp = torch.empty_like(ha), torch.empty_like(hp)
        call("lhg_recon_loss_backward", ptr(ha), ptr(ta), ptr(hp), ptr(tp), B * Cc, H, W, ptr(sums), ptr(g),
             ptr(g_ha), ptr(g_hp), stream_ptr())
        return g_ha, None, g_hp, None


def psnr_ssim(hat: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """(PSNR, SSIM) of torchmetrics' default configuration as a 2-element device tensor: three reduction launches and one tiled
    separable-window kernel (csrc/metrics.hip), no host synchronisation.  ref: watermelon.py:134-135, 447-456."""
    if not hat.is_cuda:
        raise RuntimeError("psnr_ssim runs on the GPU only")
    h, t = hat.detach().contiguous().float(), target.detach().contiguous().float()
    if h.shape != t.shape or h.dim() < 2:
        raise ValueError(f"psnr_ssim: shapes {tuple(h.shape)} vs {tuple(t.shape)}")
    H, W = h.shape[-2], h.shape[-1]
    planes = h.numel() // (H * W)
    nbytes = int(native.load().lhg_psnr_ssim_workspace(planes, H, W))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=h.device)
    out = torch.empty((2,), dtype=torch.float32, device=h.device)
    native.call("lhg_psnr_ssim", native.ptr(h), native.ptr(t), planes, H, W, native.ptr(out), native.ptr(ws), nbytes, native.stream_ptr())
    return out
