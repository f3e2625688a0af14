"""Autograd ops for the fused angular-spectrum operator (C ABI: lhg_asm_*).

    out = crop( IFFT2( F1 (.) F2 (.) FFT2( pad( in ) ) ) )

is linear in the complex field, so its adjoint is the same operator with conjugated factors
(FFT^H = N*IFFT and IFFT^H = FFT/N cancel).  The backward therefore re-uses the three forward
kernels; only the polar <-> cartesian Jacobians at both ends are extra (pointwise, 384^2 planes).
"""

from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Optional

import torch
from torch.autograd import Function

from . import native
from .native import (F_DIV, F_DIV_CONJ, F_MUL, F_MUL_CONJ, F_NONE, IN_COMPLEX, IN_PHASE, IN_POLAR, OUT_ABS, OUT_ABS_ANGLE,
                     OUT_COMPLEX, call, ptr, stream_ptr)

_ADJOINT_OP = {F_NONE: F_NONE, F_MUL: F_MUL_CONJ, F_MUL_CONJ: F_MUL, F_DIV: F_DIV_CONJ, F_DIV_CONJ: F_DIV}

_twiddle_cache: dict = {}
_FUSED_JACOBIANS = os.environ.get("LHG_FUSED_JACOBIANS", "1") != "0"  # 0: the torch expressions in every backward (A/B measurements)


def twiddles(n: int, device) -> torch.Tensor:
    key = (n, str(device))
    t = _twiddle_cache.get(key)
    if t is None:
        floats = int(native.load().lhg_fft_table_floats(n))  # 2n for a direct transform, the Bluestein table otherwise
        t = torch.empty((floats // 2, 2), dtype=torch.float32, device=device)
        call("lhg_fft_twiddles", ptr(t), n, stream_ptr())
        _twiddle_cache[key] = t
    return t


def smooth_extent(n: int) -> bool:
    """Lengths the LDS FFT transforms directly (radix 4 / 2 / 3 / 5 / 7 / 11 / 13 Stockham stages): products of those primes in
    [16, 4096] that one 256-thread workgroup holds as a row (16 inputs per thread and stage; 12 with a factor 3, 15 / 14 / 11 / 13 with
    5 / 7 / 11 / 13) — 832 = 2^6 13, 2800 = 2^4 5^2 7, the 4K geometry 2304 x 4096.  Mirrors smooth_in_range in csrc/asm_fft.hip."""
    budget = min([16] + [b for q, b in ((3, 12), (5, 15), (7, 14), (11, 11), (13, 13)) if n % q == 0])
    if not 16 <= n <= min(4096, budget * 256):
        return False
    for q in (2, 3, 5, 7, 11, 13):
        while n % q == 0:
            n //= q
    return n == 1


def supported_extent(n: int) -> bool:
    """Lengths the fused operator handles: the direct ones, and ANY length in [16, 8192] through a Bluestein convolution of
    length m >= 2n - 1 (a power of two up to 4096, the shortest 2^a 3^b length above: at most 16384) inside the same kernels; 832 = 192 + 2*320
    (the CLI's default pad on the 192^2 frames) is direct, 2800 x 4976 (a 4K frame with that pad) has direct rows and 10368-point columns."""
    return smooth_extent(n) or 16 <= n <= 8192


@dataclass
class Factor:
    """One multiplicative factor of the frequency-domain filter: slabs (S, R, C) complex64 and,
    per plane, which slab applies (None = slab 0 for every plane)."""
    slabs: torch.Tensor
    op: int = F_MUL
    index: Optional[torch.Tensor] = None  # int32 (planes,) on the device

    def adjoint(self):
        return Factor(self.slabs, _ADJOINT_OP[self.op], self.index)


@dataclass
class Geometry:
    rows0: int
    cols0: int
    pad_r: int
    pad_c: int

    @property
    def rows(self):
        return self.rows0 + 2 * self.pad_r

    @property
    def cols(self):
        return self.cols0 + 2 * self.pad_c

    def supported(self):
        """True: the fused HIP operator runs this geometry (every padded extent in [16, 8192]; round 3 dropped the LHG_ASM_ROCFFT switch
        that sent the non-direct extents to torch.fft: the operator is faster than that route on them now, tools/bench_bluestein.py)."""
        return supported_extent(self.rows) and supported_extent(self.cols)


@dataclass
class Spec:
    geom: Geometry
    in_mode: int
    out_mode: int
    phase_scale: float = 1.0
    factors: tuple = field(default_factory=tuple)  # up to two Factor


def _filter_struct(factors, planes, geom):
    keep = []
    st = native.AsmFilter()
    for i, f in enumerate(factors[:2]):
        if f.slabs.dtype != torch.complex64 or f.slabs.dim() != 3 or tuple(f.slabs.shape[-2:]) != (geom.rows, geom.cols):
            raise ValueError(f"filter slabs must be complex64 (S,{geom.rows},{geom.cols}), got {f.slabs.dtype} {tuple(f.slabs.shape)}")
        sl = f.slabs if f.slabs.is_contiguous() else f.slabs.contiguous()
        idx = f.index
        if idx is not None:
            if idx.dtype != torch.int32 or idx.numel() != planes:
                raise ValueError("filter index must be int32 with one entry per plane")
        elif sl.shape[0] != 1:
            raise ValueError("a factor with several slabs needs a per-plane index")
        keep += [sl, idx]
        if i == 0:
            st.f1, st.f1_index, st.f1_op = ptr(sl), ptr(idx), f.op
        else:
            st.f2, st.f2_index, st.f2_op = ptr(sl), ptr(idx), f.op
    if len(factors) > 2:
        raise ValueError("at most two filter factors")
    return st, keep


def _workspace(planes, geom, device, n_buffers):
    nbytes = n_buffers * planes * geom.rows0 * geom.cols * 8
    return torch.empty((nbytes // 4,), dtype=torch.float32, device=device), nbytes


def propagate_raw(a, b, spec: Spec, want_complex_copy=False, plane_src=None, out_lead=None):
    """Run the three passes.  a, b: real (..., rows0, cols0) (b None for IN_PHASE) or a complex for IN_COMPLEX.
    ``plane_src`` (int32 device tensor) + ``out_lead`` (leading shape of the outputs): several filters of ONE field — output plane q is
    input plane plane_src[q] under filter q; the first pass runs once per input plane (lhg_asm_propagate_shared)."""
    g = spec.geom
    in_planes = 1
    for d in a.shape[:-2]:
        in_planes *= d
    lead = a.shape[:-2] if out_lead is None else tuple(out_lead)
    planes = 1
    for d in lead:
        planes *= d
    if plane_src is None and planes != in_planes:
        raise ValueError("propagate_raw: another number of outputs than inputs needs plane_src")
    if plane_src is not None and (plane_src.dtype != torch.int32 or plane_src.numel() != planes):
        raise ValueError("plane_src must be int32 with one entry per output plane")
    if tuple(a.shape[-2:]) != (g.rows0, g.cols0):
        raise ValueError(f"field has extent {tuple(a.shape[-2:])}, geometry says {(g.rows0, g.cols0)}")
    dev = a.device
    # contiguous copies are bound to locals that live until after the launch: a temporary's block could be handed to the
    # workspace / output allocations below and be overwritten while the kernel still reads it
    a = a.contiguous()
    b = b.contiguous() if b is not None else None
    if spec.in_mode == IN_COMPLEX:
        if a.dtype != torch.complex64:
            raise TypeError("IN_COMPLEX needs complex64")
        pa, pb = ptr(torch.view_as_real(a)), None
    else:
        pa = ptr(a)
        pb = ptr(b) if b is not None else None
        if spec.in_mode == IN_POLAR and b is None:
            raise ValueError("IN_POLAR needs amplitude and phase")
    st, keep = _filter_struct(spec.factors, planes, g)
    nbytes = (in_planes + planes) * g.rows0 * g.cols * 8
    ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=dev)
    out_a = out_b = out_c = None
    shape = tuple(lead) + (g.rows0, g.cols0)
    if spec.out_mode == OUT_COMPLEX or want_complex_copy:
        out_c = torch.empty(shape, dtype=torch.complex64, device=dev)
    if spec.out_mode in (OUT_ABS_ANGLE, OUT_ABS):
        out_a = torch.empty(shape, dtype=torch.float32, device=dev)
    if spec.out_mode == OUT_ABS_ANGLE:
        out_b = torch.empty(shape, dtype=torch.float32, device=dev)
    call("lhg_asm_propagate_shared", pa, pb, spec.in_mode, float(spec.phase_scale), in_planes, ptr(plane_src), planes, g.rows0, g.cols0, g.pad_r,
         g.pad_c, ctypes.addressof(st), ptr(out_a), ptr(out_b), ptr(torch.view_as_real(out_c)) if out_c is not None else None,
         spec.out_mode, ptr(ws), nbytes, ptr(twiddles(g.rows, dev)), ptr(twiddles(g.cols, dev)), stream_ptr())
    del keep
    return out_a, out_b, out_c


def to_spectrum_raw(a, b, spec: Spec):
    g = spec.geom
    lead = a.shape[:-2]
    planes = 1
    for d in lead:
        planes *= d
    dev = a.device
    a = a.contiguous()
    b = b.contiguous() if b is not None else None  # kept alive until after the launch (see propagate_raw)
    if spec.in_mode == IN_COMPLEX:
        pa, pb = ptr(torch.view_as_real(a)), None
    else:
        pa, pb = ptr(a), (ptr(b) if b is not None else None)
    st, keep = _filter_struct(spec.factors, planes, g)
    ws, nbytes = _workspace(planes, g, dev, 1)
    out = torch.empty(tuple(lead) + (g.rows, g.cols), dtype=torch.complex64, device=dev)
    call("lhg_asm_to_spectrum", pa, pb, spec.in_mode, float(spec.phase_scale), planes, g.rows0, g.cols0, g.pad_r, g.pad_c,
         ctypes.addressof(st), ptr(torch.view_as_real(out)), ptr(ws), nbytes, ptr(twiddles(g.rows, dev)), ptr(twiddles(g.cols, dev)),
         stream_ptr())
    del keep
    return out


def from_spectrum_raw(S, spec: Spec, want_complex_copy=False, plane_src=None, out_lead=None):
    """``plane_src`` + ``out_lead``: several filters of one spectrum (output plane q reads spectrum plane plane_src[q])."""
    g = spec.geom
    lead = S.shape[:-2] if out_lead is None else tuple(out_lead)
    planes = 1
    for d in lead:
        planes *= d
    if plane_src is not None and (plane_src.dtype != torch.int32 or plane_src.numel() != planes):
        raise ValueError("plane_src must be int32 with one entry per output plane")
    if tuple(S.shape[-2:]) != (g.rows, g.cols) or S.dtype != torch.complex64:
        raise ValueError("spectrum must be complex64 (..., rows, cols)")
    dev = S.device
    S = S.contiguous()
    st, keep = _filter_struct(spec.factors, planes, g)
    ws, nbytes = _workspace(planes, g, dev, 1)
    shape = tuple(lead) + (g.rows0, g.cols0)
    out_a = out_b = out_c = None
    if spec.out_mode == OUT_COMPLEX or want_complex_copy:
        out_c = torch.empty(shape, dtype=torch.complex64, device=dev)
    if spec.out_mode in (OUT_ABS_ANGLE, OUT_ABS):
        out_a = torch.empty(shape, dtype=torch.float32, device=dev)
    if spec.out_mode == OUT_ABS_ANGLE:
        out_b = torch.empty(shape, dtype=torch.float32, device=dev)
    call("lhg_asm_from_spectrum_shared", ptr(torch.view_as_real(S)), ptr(plane_src), planes, g.rows0, g.cols0, g.pad_r, g.pad_c, ctypes.addressof(st),
         ptr(out_a), ptr(out_b), ptr(torch.view_as_real(out_c)) if out_c is not None else None, spec.out_mode, ptr(ws), nbytes,
         ptr(twiddles(g.rows, dev)), ptr(twiddles(g.cols, dev)), stream_ptr())
    del keep
    return out_a, out_b, out_c


# --------------------------------------------------------------------------- Jacobians at the two ends
def _fused_jacobians(*tensors) -> bool:
    """The one-launch Jacobian kernels (lhg_polar_*_cotangent) serve plain backward passes on the GPU; a backward that is itself being
    recorded (double backward) keeps the differentiable torch expressions below."""
    return _FUSED_JACOBIANS and not torch.is_grad_enabled() and all(t is None or (t.is_cuda and t.dtype in (torch.float32, torch.complex64)) for t in tensors)


def _output_cotangent(spec, z, g_a, g_b, g_c, scale=1.0):
    """``scale`` times the cotangent of the complex field z from the cotangents of the requested outputs
    (PyTorch convention: grad of a complex tensor = dL/dRe + i dL/dIm)."""
    if spec.out_mode == OUT_COMPLEX:
        return g_c if scale == 1.0 or g_c is None else g_c * scale
    if g_a is None and g_b is None:
        return None
    if _fused_jacobians(z, g_a, g_b):
        z = z.contiguous()
        g_a = g_a.contiguous() if g_a is not None else None
        g_b = g_b.contiguous() if g_b is not None else None
        gz = torch.empty_like(z)
        call("lhg_polar_output_cotangent", ptr(torch.view_as_real(z)), ptr(g_a), ptr(g_b), float(scale), ptr(torch.view_as_real(gz)), z.numel(), stream_ptr())
        return gz
    mag2 = z.real * z.real + z.imag * z.imag
    safe = mag2 > 0
    total = None
    if g_a is not None:
        inv = torch.where(safe, torch.rsqrt(mag2.clamp_min(1e-45)), torch.zeros_like(mag2))
        total = torch.complex(g_a * z.real * inv, g_a * z.imag * inv)
    if g_b is not None:
        inv2 = torch.where(safe, 1.0 / mag2.clamp_min(1e-45), torch.zeros_like(mag2))
        t = torch.complex(-g_b * z.imag * inv2, g_b * z.real * inv2)
        total = t if total is None else total + t
    return total if scale == 1.0 or total is None else total * scale


def _input_cotangent(spec, a, b, g_in, pre_scale=1.0):
    """Cotangents of (a, b) from ``pre_scale`` times the cotangent g_in of the complex input field."""
    if spec.in_mode == IN_COMPLEX:
        return (g_in if pre_scale == 1.0 else g_in * pre_scale), None
    if _fused_jacobians(g_in, a, b):
        g_in, a = g_in.contiguous(), a.contiguous()
        polar = spec.in_mode == IN_POLAR
        b = b.contiguous() if polar else None
        ga = torch.empty_like(a)
        gb = torch.empty_like(a) if polar else None
        call("lhg_polar_input_cotangent", ptr(torch.view_as_real(g_in)), ptr(a), ptr(b), float(spec.phase_scale), float(pre_scale), ptr(ga), ptr(gb),
             a.numel(), stream_ptr())
        return ga, gb
    if pre_scale != 1.0:
        g_in = g_in * pre_scale
    phs = (b if spec.in_mode == IN_POLAR else a) * spec.phase_scale
    c, s = torch.cos(phs), torch.sin(phs)
    radial = g_in.real * c + g_in.imag * s  # Re(conj(u) g)
    tangential = -g_in.real * s + g_in.imag * c  # Im(conj(u) g)
    if spec.in_mode == IN_POLAR:
        return radial, tangential * a * spec.phase_scale
    return tangential * spec.phase_scale, None


class PropagateFn(Function):
    """Differentiable fused propagation.  Returns (amp, phase, complex) with unused entries None."""

    @staticmethod
    def forward(ctx, a, b, spec):
        need_z = spec.out_mode != OUT_COMPLEX and any(ctx.needs_input_grad[:2])
        out_a, out_b, out_c = propagate_raw(a, b, spec, want_complex_copy=need_z)
        ctx.spec = spec
        ctx.save_for_backward(a, b, out_c if need_z else None)
        ctx.set_materialize_grads(False)
        if spec.out_mode != OUT_COMPLEX and out_c is not None:
            ctx.mark_non_differentiable(out_c)
        return out_a, out_b, out_c

    @staticmethod
    def backward(ctx, g_a, g_b, g_c):
        a, b, z = ctx.saved_tensors
        spec = ctx.spec
        gz = _output_cotangent(spec, z, g_a, g_b, g_c)
        if gz is None:
            return None, None, None
        adj = Spec(spec.geom, IN_COMPLEX, OUT_COMPLEX, 1.0, tuple(f.adjoint() for f in spec.factors))
        _, _, g_in = PropagateFn.apply(gz.contiguous(), None, adj) if torch.is_grad_enabled() else propagate_raw(gz.contiguous(), None, adj)
        ga, gb = _input_cotangent(spec, a, b, g_in)
        return (ga if ctx.needs_input_grad[0] else None), (gb if b is not None and ctx.needs_input_grad[1] else None), None


class ToSpectrumFn(Function):
    """S = F (.) FFT2(pad(in)), full (rows, cols) spectrum."""

    @staticmethod
    def forward(ctx, a, b, spec):
        ctx.spec = spec
        ctx.save_for_backward(a, b)
        return to_spectrum_raw(a, b, spec)

    @staticmethod
    def backward(ctx, gS):
        a, b = ctx.saved_tensors
        spec = ctx.spec
        g = spec.geom
        adj = Spec(g, IN_COMPLEX, OUT_COMPLEX, 1.0, tuple(f.adjoint() for f in spec.factors))
        _, _, g_in = FromSpectrumFn.apply(gS.contiguous(), adj)
        ga, gb = _input_cotangent(spec, a, b, g_in, float(g.rows * g.cols))  # FFT2^H = (R*C) * IFFT2: folded into the Jacobian's launch
        return ga, (gb if b is not None else None), None


class FromSpectrumFn(Function):
    """out = crop(IFFT2(F (.) S)).  Returns (amp, phase, complex)."""

    @staticmethod
    def forward(ctx, S, spec):
        need_z = spec.out_mode != OUT_COMPLEX and ctx.needs_input_grad[0]
        out_a, out_b, out_c = from_spectrum_raw(S, spec, want_complex_copy=need_z)
        ctx.spec = spec
        ctx.save_for_backward(out_c if need_z else None)
        ctx.set_materialize_grads(False)
        if spec.out_mode != OUT_COMPLEX and out_c is not None:
            ctx.mark_non_differentiable(out_c)
        return out_a, out_b, out_c

    @staticmethod
    def backward(ctx, g_a, g_b, g_c):
        (z,) = ctx.saved_tensors
        spec = ctx.spec
        g = spec.geom
        # IFFT2^H = FFT2 / (R*C): the factor rides on the cropped cotangent (linear operator) instead of a pass over the full spectrum
        gz = _output_cotangent(spec, z, g_a, g_b, g_c, 1.0 / float(g.rows * g.cols))
        if gz is None:
            return None, None
        adj = Spec(g, IN_COMPLEX, OUT_COMPLEX, 1.0, tuple(f.adjoint() for f in spec.factors))
        gS = ToSpectrumFn.apply(gz.contiguous(), None, adj)
        return gS, None
