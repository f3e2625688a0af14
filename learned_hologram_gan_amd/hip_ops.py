"""torch.autograd ops over the C ABI of liblhg_hip.so.

Activations between ops are NHWC fp32 tensors of shape (N, H, W, C) whose last dim has stride 1;
channel slices of a wider buffer are allowed (the pixel stride is passed to the kernels as `ld`),
which is how UNet skip connections avoid ``torch.cat`` copies.

Every backward is written with the same differentiable ops (conv forward <-> input-gradient <->
weight-gradient form a closed family; BN backward has an explicit double backward), so
``autograd.grad(create_graph=True)`` through the critic — the WGAN-GP penalty of
ref: watermelon_hologram/watermelon.py:458-477 — composes without any special casing.

There is no CPU path here: every op raises if its tensors are not on the GPU.
"""

from __future__ import annotations

import ctypes
import gc
import os
import weakref

import torch
from torch.autograd import Function

from . import native
from .native import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, call, ptr, stream_ptr  # noqa: F401

BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# Element type of every NHWC activation (and activation-gradient) tensor between the ops: fp32, or bf16 in the bf16 storage mode
# (set_activation_storage; BASELINE configs[2] / [4]).  NCHW tensors at the model boundary, planar outputs, parameters, statistics
# and gradients of parameters are fp32 in both.
_ACT_DTYPE = torch.float32


def act_dtype() -> torch.dtype:
    return _ACT_DTYPE


def set_activation_storage(name: str) -> None:
    """"fp32" or "bf16".  bf16 storage reads the conv-GEMM operands from HBM as bf16, so it also selects the "bf16" conv precision;
    going back to fp32 storage restores the default precision.  Tensors created under one setting must not be used under the other."""
    global _ACT_DTYPE
    if name not in ("fp32", "bf16"):
        raise ValueError(f"unknown activation storage {name!r} (fp32 | bf16)")
    if name == "bf16":
        set_conv_precision("bf16")
        call("lhg_set_activation_dtype", 1)
        _ACT_DTYPE = torch.bfloat16
    else:
        call("lhg_set_activation_dtype", 0)
        _ACT_DTYPE = torch.float32
        set_conv_precision("default")
    _THIN_MODE.clear()


def activation_storage() -> str:
    return "bf16" if _ACT_DTYPE == torch.bfloat16 else "fp32"


_DEFER_GC = os.environ.get("LHG_DEFER_GC", "1") != "0"


class deferred_gc:
    """Keep CPython's cyclic garbage collector out of a region whose host thread feeds the GPU (``with deferred_gc(): step``).

    A train step records ~10^4 autograd nodes, contexts and argument tuples; while they are alive the allocation counters trip the
    collector over and over, and a pass over the old generations costs 5-8 ms during which no kernel is launched (measured on the
    MI355X box with the step shrunk to 96x96 so that the host is the limit: 17.8 -> 16.3 ms per step, and no 5 ms stall wherever a
    burst of allocations happens to fall).  Everything the step allocates is released by reference counting when the step ends; the
    collector is re-enabled on exit, so whatever is cyclic is collected between steps, when the GPU has a queue of work to hide it.
    LHG_DEFER_GC=0 leaves the collector alone."""

    def __enter__(self):
        self.was = _DEFER_GC and gc.isenabled()
        if self.was:
            gc.disable()
        return self

    def __exit__(self, *exc):
        if self.was:
            gc.enable()
        return False


def pad_to(c: int, m: int) -> int:
    return (c + m - 1) // m * m


# --------------------------------------------------------------------------- descriptors
def nhwc(t: torch.Tensor):
    """(ptr, N, H, W, C, ld) of an NHWC view; validates the layout the kernels assume."""
    if t.dim() != 4:
        raise ValueError(f"expected a 4-D NHWC tensor, got shape {tuple(t.shape)}")
    if t.dtype != _ACT_DTYPE:
        raise TypeError(f"NHWC activation tensors are {_ACT_DTYPE} in the current storage mode, got {t.dtype}")
    N, H, W, Cc = t.shape
    s = t.stride()
    ld = s[2] if W > 1 else (s[1] if H > 1 else max(Cc, s[2]))
    ok = (Cc == 1 or s[3] == 1) and (W == 1 or s[2] == ld) and (H == 1 or s[1] == W * ld) and (N == 1 or s[0] == H * W * ld)
    if not ok or ld < Cc:
        raise ValueError(f"tensor is not an NHWC (sliced) view: shape {tuple(t.shape)} stride {s}")
    return ptr(t), N, H, W, Cc, ld


def new_nhwc(N, H, W, Cc, device):
    return torch.empty((N, H, W, Cc), dtype=_ACT_DTYPE, device=device)


class OutSlot:
    """Destination view for an op's output (e.g. a channel slice of a skip-concat buffer).
    Deliberately NOT a Tensor so that Function.apply does not treat it as an autograd input:
    the op writes through the raw pointer and returns a detached alias of the view, which
    autograd then owns like any freshly allocated output."""

    __slots__ = ("t", "share")

    def __init__(self, t, share=None):
        self.t = t
        self.share = share  # AmaxShare of the buffer this view belongs to, or None


class AmaxShare:
    """One max|.| slot for a buffer that several producers fill (the two halves of a skip-concatenation buffer): every producer that
    measures its output max-accumulates into ``slot`` and counts itself in; the buffer may be tagged with the slot only when all of
    its ``parts`` producers did (a producer that cannot measure — thin convolution, another GEMM mode — simply does not count).
    ``buf`` (the whole buffer): producers that measure per-channel maxima of what they write (the BatchNorm calls) do so into their
    channel range of ONE source for the buffer (ChanMaxSource); whatever range nobody covered is measured by a pass over that slice
    only when a weight-gradient GEMM asks (operand_chanmax)."""

    __slots__ = ("slot", "parts", "writers", "buf", "csrc")

    def __init__(self, device, parts=2, buf=None):
        self.slot = fused_absmax_slot(device)
        self.parts, self.writers = parts, 0
        self.buf, self.csrc = buf, None

    def writer(self):
        """The slot for a producer about to measure into it (None when the mode has no use for it)."""
        if self.slot is not None:
            self.writers += 1
        return self.slot

    def chanmax_for(self, view, pixels):
        """ChanMaxDest for the producer of ``view`` (a channel slice of the buffer), or None (no buffer known, or nobody will ask)."""
        if self.buf is None or not chanmax_wanted(view.shape[-1]):
            return None
        off = (view.data_ptr() - self.buf.data_ptr()) // self.buf.element_size()
        if off < 0 or off + view.shape[-1] > self.buf.shape[-1] or off % 4:
            return None
        if self.csrc is None:
            self.csrc = ChanMaxSource(self.buf.shape[-1], self.buf.device)
        return ChanMaxDest(self.csrc, off, view.shape[-1], pixels, shared=True)

    def tag(self, buf):
        if self.slot is not None and self.writers >= self.parts:
            tag_absmax(buf, self.slot)
        if self.csrc is not None:  # the channels no producer covered are measured by a pass over that slice only (operand_chanmax)
            tag_chanmax_source(buf, self.csrc)
        return buf


def _out_amax(out, device, fresh_ok=True):
    """(slot, tag_target_is_fresh): where a producer with destination ``out`` (None, or an OutSlot) accumulates max|output|."""
    if isinstance(out, OutSlot):
        return out.share.writer() if out.share is not None else None
    return fused_absmax_slot(device) if fresh_ok else None


def _resolve_out(out, shape, device, dtype=None):
    if out is None:
        return torch.empty(shape, dtype=dtype or _ACT_DTYPE, device=device)
    t = out.t
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"output slot has shape {tuple(t.shape)}, op produces {tuple(shape)}")
    return t.detach()


class CatViewsFn(Function):
    """torch.cat((a, b), channel) where a and b were already written into the two channel
    slices of `buf` by their producers: forward is free, backward hands out grad slices.
    ref: torch.cat call sites neural_network_components.py:310-313 (skip first)."""

    @staticmethod
    def forward(ctx, a, b, buf_slot):
        ctx.ca = a.shape[-1]
        out = buf_slot.t.detach()
        return buf_slot.share.tag(out) if buf_slot.share is not None else out

    @staticmethod
    def backward(ctx, g):
        # the slices' elements are a subset of g's: its measured max|.| bounds theirs (the input-gradient GEMMs that read them need a scale)
        return _inherit_absmax(g[..., : ctx.ca], g), _inherit_absmax(g[..., ctx.ca :], g), None


def _as_nhwc_view(t):
    """`t` itself when it already is an NHWC (sliced) view, else a dense copy (autograd hands out expanded gradients,
    e.g. the all-ones gradient of ``.sum()``, with zero strides)."""
    N, H, W, Cc = t.shape
    s = t.stride()
    ld = s[2] if W > 1 else (s[1] if H > 1 else max(Cc, s[2]))
    ok = (Cc == 1 or s[3] == 1) and (W == 1 or s[2] == ld) and (H == 1 or s[1] == W * ld) and (N == 1 or s[0] == H * W * ld) and ld >= Cc
    return t if ok else t.contiguous()


def _dense(t):
    """Materialise a strided NHWC view as a dense tensor when a kernel needs ld == C."""
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------- weight gradients on a side stream
# The weight-gradient GEMMs are MFMA-bound and nothing in backward depends on them before the optimiser step, while the chain they
# branch off (BN backward -> input-gradient -> BN backward ...) alternates MFMA-bound and HBM-bound kernels.  For parameters whose
# gradient lives in a flat buffer (optim.FlatParams registers the slot), the weight gradient is therefore launched on a second HIP
# stream and ACCUMULATED straight into the slot; autograd gets None for it.  The main stream's HBM-bound kernels then run under the
# side stream's GEMMs.  A callback queued on the autograd engine joins the two streams when the backward pass ends, so reading
# .grad after backward() is as safe as without the second stream (join_side_stream() is that fence).  LHG_SIDE_WGRAD=0 disables.
#
# Single host thread by contract (the reference is single-threaded): the registries below are plain module globals.
_GRAD_SLOTS: dict = {}
_JOIN_QUEUED = False
_SIDE_STREAMS: dict = {}
SIDE_WGRAD = os.environ.get("LHG_SIDE_WGRAD", "1") != "0"
_SIDE_BIAS = os.environ.get("LHG_SIDE_BIAS", "1") != "0"  # 0: bias gradients are summed on the main stream (A/B measurements)
SLOT_ACCUMULATE = True  # False: weight gradients go back through autograd's own accumulation (A/B tests of the slot path)

# Gradients that never pass through autograd's AccumulateGrad (slot-accumulated weight gradients, analytically-zero biases) are
# invisible to post-accumulate hooks.  distributed.GradSynchronizer registers a listener per parameter; every recorded use of such a
# parameter is counted at forward time (note_use) and every contribution at backward time (note_contribution): when the count
# returns to zero the parameter's gradient is complete and the listener fires, so a bucket's all-reduce can start while backward is
# still running.
_ARRIVAL_LISTENERS: dict = {}  # data_ptr -> callable()
_PENDING_USES: dict = {}       # data_ptr -> recorded uses whose backward has not run yet
CONTRIBUTIONS = 0              # contributions seen so far (host-side sequence number, read by the overlap tests)
_PARAM_GRADS_OFF = 0


class only_input_gradients:
    """Context: backward passes inside it skip every parameter gradient.  ``autograd.grad(outputs, inputs=x)`` still runs each node's
    full backward and then drops what it did not ask for — for the WGAN-GP penalty (watermelon.py:466-473: d D(x^)/d x^ with
    create_graph=True) that is one dead weight-gradient GEMM per critic layer.  Values that the caller receives are unchanged."""

    def __enter__(self):
        global _PARAM_GRADS_OFF
        _PARAM_GRADS_OFF += 1

    def __exit__(self, *exc):
        global _PARAM_GRADS_OFF
        _PARAM_GRADS_OFF -= 1
        return False


def param_grads_wanted() -> bool:
    return _PARAM_GRADS_OFF == 0


def set_arrival_listener(param: torch.Tensor, fn) -> None:
    if fn is None:
        _ARRIVAL_LISTENERS.pop(param.data_ptr(), None)
        _PENDING_USES.pop(param.data_ptr(), None)
    else:
        _ARRIVAL_LISTENERS[param.data_ptr()] = fn
        _PENDING_USES[param.data_ptr()] = 0


_RECORDING = False


class TrackedFunction(Function):
    """autograd.Function whose ``apply`` remembers whether the call is being recorded: grad mode is always off INSIDE ``forward`` and
    ``ctx.needs_input_grad`` ignores it, so ``note_use`` could not tell an inference call from a recorded one."""

    @classmethod
    def apply(cls, *args):
        global _RECORDING
        prev, _RECORDING = _RECORDING, torch.is_grad_enabled()
        try:
            return super().apply(*args)
        finally:
            _RECORDING = prev


def note_use(param, needed: bool = True) -> None:
    """Forward of an op whose backward will contribute to `param` outside autograd; counted only while the graph is being recorded
    (`needed` = ctx.needs_input_grad[i])."""
    if param is not None and needed and _RECORDING and _ARRIVAL_LISTENERS:
        k = param.data_ptr()
        if k in _PENDING_USES:
            _PENDING_USES[k] += 1


def note_contribution(param) -> None:
    global CONTRIBUTIONS
    CONTRIBUTIONS += 1
    if param is None or not _ARRIVAL_LISTENERS:
        return
    k = param.data_ptr()
    left = _PENDING_USES.get(k)
    if left is None:
        return
    _PENDING_USES[k] = left - 1
    if left - 1 <= 0:
        _ARRIVAL_LISTENERS[k]()


def clear_pending_uses(params) -> None:
    """End of a backward pass over `params`: every recorded use has contributed (or belonged to a graph that was dropped)."""
    for p_ in params:
        if p_.data_ptr() in _PENDING_USES:
            _PENDING_USES[p_.data_ptr()] = 0


def reset_backward_state() -> None:
    """Start of a top-level backward: if an earlier backward raised after queueing the end-of-backward join, its callback never ran
    and the flag would suppress every later join."""
    global _JOIN_QUEUED
    _JOIN_QUEUED = False


def register_grad_slot(param: torch.Tensor, grad_view: torch.Tensor) -> None:
    """Keyed by address for the lookup in backward, but validated against a weak reference to the parameter itself: an entry whose
    parameter has died is dropped, so a later tensor that happens to reuse the address can never inherit the slot."""
    _GRAD_SLOTS[param.data_ptr()] = (weakref.ref(param), grad_view)


def unregister_grad_slots(params) -> None:
    for p_ in params:
        _GRAD_SLOTS.pop(p_.data_ptr(), None)


def _grad_slot(w):
    entry = _GRAD_SLOTS.get(w.data_ptr())
    if entry is None:
        return None
    owner = entry[0]()
    if owner is None:
        del _GRAD_SLOTS[w.data_ptr()]
        return None
    same = owner is w or (owner.data_ptr() == w.data_ptr() and owner.shape == w.shape and owner._version == w._version)
    return entry[1] if same and entry[1].shape == w.shape else None


def _side_stream(device):
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        # LHG_SIDE_PRIORITY (A/B measurements): "low" = the least priority the device offers — the weight-gradient stream has slack and
        # should yield compute units to the chain of kernels the step waits for; default: the stream's default priority
        prio = os.environ.get("LHG_SIDE_PRIORITY", "")
        if prio == "low":
            least, _greatest = torch.cuda.Stream.priority_range()
            _SIDE_STREAMS[key] = torch.cuda.Stream(key, priority=least)
        else:
            _SIDE_STREAMS[key] = torch.cuda.Stream(key)
    return _SIDE_STREAMS[key]


def side_stream(device):
    """The stream weight gradients are accumulated on (None when the second stream is switched off or `device` is not a GPU)."""
    if not SIDE_WGRAD or torch.device(device).type != "cuda":
        return None
    return _side_stream(device)


def join_side_stream(device=None) -> None:
    """Make the current stream wait for every weight gradient launched on the side stream so far."""
    for key, st in _SIDE_STREAMS.items():
        if device is None or torch.device(device).index in (None, key):
            torch.cuda.current_stream(key).wait_stream(st)


def _weight_grad(w, inputs, fn):
    """fn(slot) -> weight gradient from `inputs`: with ``slot`` None it returns a new tensor, otherwise it ACCUMULATES into ``slot``.
    Plain backward into a registered slot: run on the side stream, accumulate, return None (autograd's zero)."""
    slot = _grad_slot(w) if (SLOT_ACCUMULATE and not torch.is_grad_enabled()) else None
    if slot is None:
        return fn(None)
    if not (SIDE_WGRAD and w.is_cuda):  # same accumulate-into-the-slot contract on the caller's stream
        fn(slot)
        note_contribution(w)
        return None
    main, side = torch.cuda.current_stream(w.device), _side_stream(w.device)
    global _JOIN_QUEUED
    if not _JOIN_QUEUED:  # when this backward pass ends, its stream waits for the side stream: .grad is then safe to read as usual
        _JOIN_QUEUED = True

        def _join_at_end(dev=w.device):
            global _JOIN_QUEUED
            _JOIN_QUEUED = False
            join_side_stream(dev)

        torch.autograd.Variable._execution_engine.queue_callback(_join_at_end)
    side.wait_stream(main)  # inputs (and the zeroed slot) are ready on the main stream
    for t in inputs:
        t.record_stream(side)  # keep their memory out of main-stream reuse until the side stream is done with it
    with torch.cuda.stream(side):
        fn(slot)
    note_contribution(w)
    return None


def _small_grad_slot(p):
    """Slot of a bias / BatchNorm affine parameter for direct accumulation from the main stream, or None (autograd path)."""
    if p is None or torch.is_grad_enabled() or not SLOT_ACCUMULATE:
        return None
    return _grad_slot(p)


# --------------------------------------------------------------------------- operand precision of the conv GEMMs
_PRECISIONS = {"fp32": 0, "f32": 0, "bf16": 1, "fp32_split": 2, "fp32_split2": 3, "fp32_split_f16": 4}
_PRECISION_NAMES = {0: "fp32", 1: "bf16", 2: "fp32_split", 3: "fp32_split2", 4: "fp32_split_f16"}
_F16_SPLIT = 4
_precision = None  # read from the library on first use (it starts in lhg_default_conv_precision())


def _mode() -> int:
    global _precision
    if _precision is None:
        _precision = int(native.load().lhg_get_conv_precision())
    return _precision


def default_precision() -> str:
    """The mode the library starts in: "fp32_split_f16" unless LHG_CONV_PRECISION (fp32 | fp32_split | fp32_split2 | fp32_split_f16 | bf16)
    says otherwise."""
    return _PRECISION_NAMES[int(native.load().lhg_default_conv_precision())]


def set_conv_precision(name: str) -> None:
    """Arithmetic of the conv GEMMs (tensors are fp32 in every mode), see lhg_set_conv_precision:
    "fp32_split_f16" fp32-faithful on the fp16 matrix pipe: tensor-scaled operands as sums of two fp16 terms, three MFMA products,
                  fp32 accumulate (the GEMM operands' max|x| is measured with lhg_absmax, see operand_absmax);
    "fp32_split"  fp32-faithful on the bf16 matrix pipe: operands as exact sums of three bf16 terms, six MFMA products, fp32 accumulate;
    "fp32"        exact fp32 MFMA (v_mfma_f32_32x32x2_f32);
    "fp32_split2" two bf16 terms, three products (~2^-16 per product), measurements only;
    "bf16"        operands rounded to bf16, fp32 accumulation (BASELINE configs[2] / [4]);
    "default"     back to default_precision()."""
    global _precision
    if name == "default":
        name = default_precision()
    if name not in _PRECISIONS:
        raise ValueError(f"unknown precision {name!r} ({' | '.join(_PRECISIONS)} | default)")
    call("lhg_set_conv_precision", _PRECISIONS[name])
    _precision = _PRECISIONS[name]


def conv_precision() -> str:
    return _PRECISION_NAMES[_mode()]


class precision:
    """``with hip_ops.precision("fp32"): ...`` — the conv-GEMM arithmetic (and, with ``storage="bf16"``, the activation storage) for a
    region, restored on exit whatever happens inside.  The switches are process-global (one host thread per process, as in the
    reference); this is the form tests and tools use, so that one forgotten ``finally`` cannot change every later call's arithmetic."""

    def __init__(self, name: str = "default", storage: str | None = None):
        self.name, self.storage = name, storage

    def __enter__(self):
        self.prev_mode, self.prev_storage = conv_precision(), activation_storage()
        if self.storage is not None and self.storage != self.prev_storage:
            set_activation_storage(self.storage)
        if not (self.storage == "bf16"):
            set_conv_precision(self.name)
        return self

    def __exit__(self, *exc):
        if activation_storage() != self.prev_storage:
            set_activation_storage(self.prev_storage)
        set_conv_precision(self.prev_mode)
        return False


def operand_absmax(t):
    """max|t| of an NHWC fp32 GEMM operand as a one-element device tensor (lhg_absmax on the current stream) — the tensor scale of
    the "fp32_split_f16" mode; None in every other mode.  Measured once per operand and handed to each GEMM that reads it."""
    if _mode() != _F16_SPLIT:
        return None
    known = t.__dict__.get("_lhg_amax")
    if known is not None and known[0] == t._version and _slot_alive(known[1]):
        return known[1]
    p, N, H, W, Cc, ld = nhwc(t)
    out = _amax_slot(t.device)
    call("lhg_absmax", p, N * H * W, Cc, ld, ptr(out), stream_ptr())
    tag_absmax(t, out)
    return out


_TIMING_FAKE_CHANMAX = False  # set by _timing_switch("LHG_TIMING_FAKE_CHANMAX") once it is defined (below)
_FAKE_CMAX = {}


def operand_chanmax(t):
    """Per-channel max|t| (or an upper bound) of an NHWC fp32 operand of a WEIGHT-GRADIENT GEMM as a (C,) device tensor — the
    per-channel scales of the "fp32_split_f16" mode (the contraction runs over pixels, so a scale per channel factors out of the sum
    exactly); None in every other mode.  Sources, cheapest first: the vector the kernel that WROTE the tensor finished on its way out
    (tag_chanmax_source: the fused BatchNorm calls — nothing to launch here); lhg_channel_absmax_fused, one pass over the tensor, for
    whatever channels no producer covered."""
    if _mode() != _F16_SPLIT:
        return None
    if _TIMING_FAKE_CHANMAX:  # timing experiments only: what the step costs without the per-channel maxima passes
        key = (t.shape[-1], t.device)
        if key not in _FAKE_CMAX:
            _FAKE_CMAX[key] = torch.full((t.shape[-1],), float(os.environ.get("LHG_TIMING_FAKE_CHANMAX_VALUE", "64")), dtype=torch.float32, device=t.device)
        return _FAKE_CMAX[key]
    root = t.__dict__.get("_lhg_root", t)  # the tensor t is an alias of (Conv2dSharedInputFn's second output): one cache for both
    if root is not t and not (root.data_ptr() == t.data_ptr() and root.shape == t.shape and root.stride() == t.stride() and root._version == t._version):
        root = t
    st = stream_ptr()
    known = root.__dict__.get("_lhg_cmax")
    if known is not None and known[0] == t._version:
        if known[2] != st:  # another stream reads the vector now: keep its memory out of reuse until that stream is done with it
            known[1].record_stream(torch.cuda.current_stream(t.device))
            root.__dict__["_lhg_cmax"] = (known[0], known[1], st)
        return known[1]
    p, N, H, W, Cc, ld = nhwc(t)
    pixels = N * H * W
    src = root.__dict__.get("_lhg_cmax_src")
    if src is not None and src[0] == t._version and _FUSED_CHANMAX and src[1].vec.shape[0] == Cc:
        out = src[1].vec
        for off, cc, partial, grid_pixels, rows in src[1].pending:
            # the kernel that wrote these channels left per-workgroup partial maxima: one small launch (on the caller's stream — the
            # weight-gradient stream, beside the main chain; the caller is ordered behind that kernel, it reads t) instead of a pass
            partial.record_stream(torch.cuda.current_stream(t.device))
            if rows is not None:
                call("lhg_channel_absmax_finish_rows", ptr(partial), int(rows), cc, out.data_ptr() + 4 * off, st)
            else:
                call("lhg_channel_absmax_finish", ptr(partial), grid_pixels, cc, out.data_ptr() + 4 * off, st)
            src[1].covered.append((off, cc))
        src[1].pending.clear()
        covered = sorted(src[1].covered)
    else:
        out, covered = torch.empty((Cc,), dtype=torch.float32, device=t.device), []
    ws, tickets = fused_scratch(t.device)

    def measure(c0, c1):  # a pass over channels [c0, c1) of t
        CHANMAX_STATS["pass"] += 1
        CHANMAX_STATS["pass_bytes"] += pixels * min(ld, 32 * ((c1 - c0 + 31) // 32 + 1)) * 4
        if CHANMAX_PASS_LOG is not None:  # diagnostics (tools/chanmax_sites.py): who still pays a pass
            import traceback
            CHANMAX_PASS_LOG.append(((N, H, W, c1 - c0, ld), [f"{f.name}:{f.lineno}" for f in traceback.extract_stack(limit=8)[:-2]], type(t.grad_fn).__name__))
        call("lhg_channel_absmax_fused", p + 4 * c0, pixels, c1 - c0, ld, out.data_ptr() + 4 * c0, ws, tickets, st)

    at = 0
    for off, cc in covered:
        if off < at or off + cc > Cc:
            continue
        if off > at:
            measure(at, off)
        CHANMAX_STATS["fused"] += 1  # finished by the kernel that wrote these channels; the caller's stream is ordered behind it (it reads t)
        at = off + cc
    if at < Cc:
        measure(at, Cc)
    out.record_stream(torch.cuda.current_stream(t.device))  # (written on the producer's stream, read by a GEMM of this one)
    root.__dict__["_lhg_cmax"] = (t._version, out, st)
    return out


CHANMAX_PASS_LOG = None
CHANMAX_STATS = {"fused": 0, "pass": 0, "pass_bytes": 0}  # parts finished from a producer's partials / measured by a pass of their own
_FUSED_CHANMAX = os.environ.get("LHG_FUSED_CHANMAX", "1") != "0"


def chanmax_wanted(Cc) -> bool:
    """Will a weight-gradient GEMM ask for per-channel maxima of a tensor with ``Cc`` channels written now?  (fp16-split mode, fp32 storage.)"""
    return _mode() == _F16_SPLIT and _ACT_DTYPE == torch.float32 and _FUSED_CHANMAX and Cc % 4 == 0


# LHG_CHANMAX_IN_LAUNCH=1: the BatchNorm apply launches fold their own partial maxima (chanmax_finish of the fused calls) instead of
# leaving the rows to a finish launch on the stream that asks.  Measured (round 5): 84 launches fewer per step, but ~11 us more in every
# apply launch ON THE MAIN CHAIN (27 in the backward apply with its two outputs), where the finish launches ran on the weight-gradient
# stream beside it: the step does not get shorter.  Off by default.
_CHANMAX_IN_LAUNCH = os.environ.get("LHG_CHANMAX_IN_LAUNCH", "0") == "1"
# LHG_FUSED_BN=1: the BatchNorm calls of the MAIN chain finish their reductions inside the launch that writes the partial rows (ABI 9:
# lhg_bn_forward_train / lhg_bn_backward_fused / ..._backward_backward_fused) instead of the two-launch form (partial rows, then a reduce
# launch).  Measured, interleaved on one box (round 5, tools/ab_env3.sh, tools/ab_fused_bn.sh): ~100 launches fewer per step, and
#   384^2 batch 4 (GPU-bound):  31.63 ms fused against 31.40 two-launch — the finish costs 12 - 20 us at the end of a launch on the chain the
#                               step waits for (six dependent memory round trips), the reduce launch it replaces 8 - 10 us;
#   bf16 storage (GPU-bound):   21.2 against 20.85;
#   96^2 batch 4 (host-bound):  15.4 - 16.1 ms fused against 17.7 - 18.0 — there every launch is host time.
# BASELINE's configurations are GPU-bound, so the default is the two-launch form; reductions that run on the weight-gradient stream
# (bias gradients, per-channel-maxima passes) are always the one-launch form: beside the main chain their tail costs nothing.
_FUSED_BN = os.environ.get("LHG_FUSED_BN", "0") == "1"


class ChanMaxSource:
    """Per-channel bounds of one tensor (or of one shared buffer): ``vec`` (C floats) with the channel ranges ``covered`` holding finished
    maxima and ``pending`` = [(first channel, channels, partial rows, pixels of the producer's grid, explicit row count | None)] still
    to be finished into vec by lhg_channel_absmax_finish when a weight-gradient GEMM asks (operand_chanmax)."""

    __slots__ = ("vec", "covered", "pending")

    def __init__(self, Cc, device):
        self.vec = torch.empty((Cc,), dtype=torch.float32, device=device)
        self.covered, self.pending = [], []


class ChanMaxDest:
    """What a producer kernel is handed for the per-channel maxima of the ``cc`` channels it writes: ``ptr`` / ``finish`` are the
    y_chanmax / chanmax_finish arguments of the fused BatchNorm calls; ``commit(t)`` records the result once the launch is queued."""

    __slots__ = ("src", "off", "cc", "pixels", "shared", "partial", "ptr", "finish")

    def __init__(self, src, off, cc, pixels, shared=False):
        self.src, self.off, self.cc, self.pixels, self.shared = src, off, cc, pixels, shared
        self.finish = int(_CHANMAX_IN_LAUNCH)
        if self.finish:
            self.partial, self.ptr = None, src.vec.data_ptr() + 4 * off
        else:
            rows = int(native.load().lhg_chanmax_partial_rows(pixels, cc))
            self.partial = torch.empty((rows * cc,), dtype=torch.float32, device=src.vec.device)
            self.ptr = self.partial.data_ptr()

    def commit(self, t):
        if self.finish:
            self.src.covered.append((self.off, self.cc))
        else:
            self.src.pending.append((self.off, self.cc, self.partial, self.pixels, None))
        if not self.shared:
            tag_chanmax_source(t, self.src)
        return t


def chanmax_dest(Cc, pixels, device, out=None):
    """ChanMaxDest for a kernel about to write a tensor with ``Cc`` channels over ``pixels`` pixels, or None when nobody will ask.
    ``out``: the op's destination — an OutSlot of a shared buffer gets its channel range of the buffer's source."""
    if isinstance(out, OutSlot) and out.share is not None:
        return out.share.chanmax_for(out.t, pixels)
    if not chanmax_wanted(Cc):
        return None
    return ChanMaxDest(ChanMaxSource(Cc, device), 0, Cc, pixels)


def chanmax_partial_for(pixels, Cc, device):
    """Buffer for the per-workgroup partial maxima of the UNFUSED calls (lhg_bn_apply_chanmax / lhg_bn_backward_chanmax: kept for the
    C ABI's users and the operator tests; the op layer uses the fused calls)."""
    if not chanmax_wanted(Cc):
        return None
    rows = int(native.load().lhg_chanmax_partial_rows(pixels, Cc))
    return torch.empty((rows * Cc,), dtype=torch.float32, device=device)


def tag_chanmax_source(t, src):
    """Remember where per-channel bounds of t come from (a ChanMaxSource, or None)."""
    if src is not None:
        t.__dict__["_lhg_cmax_src"] = (t._version, src)
    return t


def _inherit_chanmax(dst, src, alias=False):
    """dst's elements are a subset of src's per channel (max-pooling, an alias): src's per-channel bounds hold for dst.  ``alias``: dst IS
    src (same memory): they also share the finished values."""
    known = src.__dict__.get("_lhg_cmax_src")
    if known is not None and known[0] == src._version:
        dst.__dict__["_lhg_cmax_src"] = (dst._version, known[1])
    if alias:
        dst.__dict__["_lhg_root"] = src.__dict__.get("_lhg_root", src)
    return dst


_FUSED_SCRATCH = {}


def fused_scratch(device):
    """(ws, tickets) data pointers of the fused calls' scratch for the current stream (include/lhg_hip.h "fused calls"): one persistent
    pair per (device, stream, capture state) — launches of one stream never overlap, and every fused call consumes what it put there
    before it returns the stream to the next.  The tickets are zero-filled once; every call leaves them at zero."""
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), stream_ptr(),
           torch.cuda.is_current_stream_capturing())
    hit = _FUSED_SCRATCH.get(key)
    if hit is None:
        lib = native.load()
        dev = torch.device("cuda", key[0])
        ws = torch.empty((int(lib.lhg_fused_workspace_floats()),), dtype=torch.float32, device=dev)
        tickets = torch.zeros((int(lib.lhg_fused_ticket_count()),), dtype=torch.int32, device=dev)
        hit = _FUSED_SCRATCH[key] = (ws, tickets, ws.data_ptr(), tickets.data_ptr())
    return hit[2], hit[3]


def fused_absmax_slot(device):
    """A zeroed slot for a producer kernel that measures max|output| on its way out (lhg_bn_apply, lhg_bn_backward), or None when the
    current modes have no use for it."""
    if _mode() != _F16_SPLIT or _ACT_DTYPE != torch.float32 or not _FUSED_ABSMAX:
        return None
    return _amax_slot(device)


def _inherit_absmax(dst, src):
    """dst's elements are a subset of src's (or zeros): src's known max|.| bounds dst's."""
    known = src.__dict__.get("_lhg_amax")
    if known is not None and known[0] == src._version and _slot_alive(known[1]):
        tag_absmax(dst, known[1])
    return dst


def tag_absmax(t, slot):
    """Remember max|t| on the tensor object (with its version counter: an in-place update — autograd's gradient accumulation included —
    invalidates it).  Only for tensors nobody writes through raw pointers after this point."""
    if slot is not None:
        t.__dict__["_lhg_amax"] = (t._version, slot)
    return t


_EPILOGUE_GX_AMAX = os.environ.get("LHG_EPILOGUE_GX_AMAX", "1") != "0"  # 0: input-gradient GEMMs do not measure max|gx| (A/B measurements)
_FUSED_ABSMAX = os.environ.get("LHG_FUSED_ABSMAX", "1") != "0"  # 0: every operand is measured by lhg_absmax (A/B measurements)
ABSMAX_WORDS = 1  # LHG_ABSMAX_WORDS: an operand's max|x| is the maximum of this many floats (include/lhg_hip.h)
_AMAX_POOL = {}  # (device, stream, capturing) -> _SlotRing
_RING_POOLS, _POOL_SLOTS = 64, 1024


class _SlotRing:
    """Zero-filled one-float slots for lhg_absmax (it max-accumulates), handed out once each.  One fill launch prepares 1024 slots.
    The pools form a ring of 64 that is never freed: a slot's memory is reused only 65536 measurements later (re-zeroed on the
    stream that owns the ring), so a kernel of ANOTHER stream that still reads a slot — the weight-gradient GEMMs run on the side
    stream — can never see its pool handed to a different tensor by the caching allocator.  Every pool carries a generation counter;
    a tag on a long-lived tensor (tag_absmax) is honoured only while its pool has not been recycled."""

    __slots__ = ("pools", "gens", "cur", "nxt", "device")

    def __init__(self, device):
        self.device, self.pools, self.gens, self.cur, self.nxt = device, [], [], -1, _POOL_SLOTS

    def take(self):
        if self.nxt >= _POOL_SLOTS:
            self.cur = (self.cur + 1) % _RING_POOLS
            if self.cur == len(self.pools):
                self.pools.append(torch.zeros((_POOL_SLOTS * ABSMAX_WORDS,), dtype=torch.float32, device=self.device))
                self.gens.append([0])
            else:
                self.pools[self.cur].zero_()
                self.gens[self.cur][0] += 1
            self.nxt = 0
        i = self.nxt
        self.nxt = i + 1
        slot = self.pools[self.cur][i * ABSMAX_WORDS:(i + 1) * ABSMAX_WORDS]
        slot.__dict__["_lhg_gen"] = (self.gens[self.cur], self.gens[self.cur][0])
        return slot


def _slot_alive(slot) -> bool:
    g = slot.__dict__.get("_lhg_gen")
    return g is None or g[0][0] == g[1]


def _amax_slot(device):
    """A zeroed slot on the ring of the current stream — fill and measurement are ordered by the stream they both run on — and of the
    current capture state (a pool filled outside a graph capture is not used inside one: the fill has to be part of the graph for
    the replays to start from zero)."""
    key = (device, stream_ptr(), torch.cuda.is_current_stream_capturing())
    ring = _AMAX_POOL.get(key)
    if ring is None:
        ring = _AMAX_POOL[key] = _SlotRing(device)
    return ring.take()


def begin_graph_capture() -> None:
    """Call right before a hipGraph capture: the zero-filled slot rings of EARLIER captures are dropped, so that this capture fills its
    own (a slot handed out from a ring whose fill launch belongs to another graph would never be re-zeroed by this graph's replays:
    lhg_absmax max-accumulates, and the scales — though still valid upper bounds — would depend on the replay history)."""
    for key in [k for k in _AMAX_POOL if k[2]]:
        del _AMAX_POOL[key]


def apply_env_precision() -> None:
    """Kept for the entry points (trainingModel.py, generatePOH.py): LHG_CONV_PRECISION is read by the library itself at load time."""
    _mode()


# --------------------------------------------------------------------------- weight packing
def pack_weight(w: torch.Tensor, rows_from_d0: bool, k_pad_to: int = 32) -> torch.Tensor:
    """OIHW / IOHW -> [KH*KW][rows_pad][k_pad] panels (lhg_pack_weight).

    The packed copy is cached ON the weight tensor object (so it dies with it — an id()/data_ptr
    keyed table would alias a freed temporary) and is reused while (data_ptr, _version) are
    unchanged; raw-pointer updates must call bump_version()."""
    D0, D1, KH, KW = w.shape
    rows, K = (D0, D1) if rows_from_d0 else (D1, D0)
    rows_pad, k_pad = pad_to(rows, 64), pad_to(K, k_pad_to)
    stamp = (w.data_ptr(), w._version, tuple(w.shape))
    try:
        cache = w.__dict__.setdefault("_lhg_packed", {})
    except AttributeError:  # pragma: no cover
        cache = {}
    hit = cache.get((rows_from_d0, k_pad, _mode()))
    if hit is not None and hit[0] == stamp:
        return hit[1]
    wd = w.detach()
    if not wd.is_contiguous():
        wd = wd.contiguous()
    out = _packed_buffer(KH, KW, rows_pad, k_pad, w.device)
    call("lhg_pack_weight", ptr(wd), D0, D1, KH, KW, int(rows_from_d0), ptr(out), rows_pad, k_pad, stream_ptr())
    cache[(rows_from_d0, k_pad, _mode())] = (stamp, out)
    return out


def _packed_buffer(KH, KW, rows_pad, k_pad, device):
    floats = int(native.load().lhg_packed_weight_floats(KH * KW, rows_pad, k_pad))
    panel_rows = KH * KW * rows_pad
    buf = torch.empty((floats,), dtype=torch.float32, device=device)  # panels (+ 16 bytes behind them in the fp16-split mode: max|w|)
    return buf[: panel_rows * (floats // panel_rows)].view(KH * KW, rows_pad, floats // panel_rows)


def repack_weights(params) -> int:
    """Refresh, in one batched call (lhg_pack_weights), every packed form the ops hold of weights that changed since they were
    packed — what an optimiser step leaves behind (ref: watermelon.py:137-138).  Without it each conv re-packs its weight on its
    next use: the same values from ~200 small launches per train step.  The panels are overwritten IN PLACE (no allocation; a backward
    pass that still wants the old version of a weight is an error in autograd's terms anyway).  Returns the number of panels written."""
    mode, items, fresh = _mode(), [], []
    for w in params:
        cache = w.__dict__.get("_lhg_packed")
        if not cache or w.dim() != 4 or not w.is_cuda or not w.is_contiguous():
            continue
        stamp = (w.data_ptr(), w._version, tuple(w.shape))
        D0, D1, KH, KW = w.shape
        for key, (old, out) in list(cache.items()):
            rows_from_d0, k_pad, m = key
            if m != mode or old == stamp or old[0] != stamp[0] or old[2] != stamp[2]:
                continue
            rows_pad = pad_to(D0 if rows_from_d0 else D1, 64)
            # written in place: the panels' readers (gather-GEMMs of the stale version) were enqueued on this stream before this call
            items.append(native.PackItem(w.data_ptr(), out.data_ptr(), D0, D1, KH, KW, int(rows_from_d0), rows_pad, k_pad))
            fresh.append((cache, key, stamp, out))
    if items:
        arr = (native.PackItem * len(items))(*items)
        call("lhg_pack_weights", arr, len(items), stream_ptr())
        for cache, key, stamp, out in fresh:
            cache[key] = (stamp, out)
    return len(items)


def bump_version(t: torch.Tensor):
    """Raw-pointer writes do not touch autograd's version counter; bump it so cached packed
    weights are invalidated after an optimiser step."""
    try:
        torch.autograd.graph.increment_version(t)
    except AttributeError:  # pragma: no cover
        t.add_(0)


# --------------------------------------------------------------------------- layout
class ToNHWC(Function):
    """NCHW -> NHWC with zero-padded channels."""

    @staticmethod
    def forward(ctx, x, ld):
        x = x.contiguous()
        N, Cc, H, W = x.shape
        ctx.C = Cc
        out = new_nhwc(N, H, W, ld, x.device)
        call("lhg_nchw_to_nhwc", ptr(x), ptr(out), N, Cc, H, W, ld, stream_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        return ToNCHW.apply(g, ctx.C), None


class ToNCHW(Function):
    """First C channels of an NHWC tensor -> NCHW."""

    @staticmethod
    def forward(ctx, x, Cc):
        p, N, H, W, Cx, ld = nhwc(x)
        ctx.Cx = Cx
        out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
        call("lhg_nhwc_to_nchw", p, ld, ptr(out), N, Cc, H, W, stream_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        return ToNHWC.apply(g, ctx.Cx), None


# --------------------------------------------------------------------------- convolution family
def _conv_out_hw(H, W, k, stride):
    p = k // 2
    return (H + 2 * p - k) // stride + 1, (W + 2 * p - k) // stride + 1


_THIN_MODE = {}


def thin_mode(Ci, Co, k, stride):
    """0: MFMA gather-GEMM; 1 / 2: direct kernels of csrc/thin_conv.hip (thin input / thin output).  LHG_THIN=0 disables."""
    key = (Ci, Co, k, stride)
    if key not in _THIN_MODE:
        on = os.environ.get("LHG_THIN", "1") != "0" and _ACT_DTYPE == torch.float32  # bf16 storage: every layer on the MFMA path
        _THIN_MODE[key] = int(native.load().lhg_conv2d_thin_supported(Ci, Co, k, stride)) if on else 0
    return _THIN_MODE[key]


def _raw_weight(w):
    wd = w.detach()
    return wd if wd.is_contiguous() else wd.contiguous()


_SPLITK_FLOATS = {}


_STATS_ROWS_BOUND = {}


def _stats_rows_bound(N, Ho, Wo):
    key = (N, Ho, Wo)
    n = _STATS_ROWS_BOUND.get(key)
    if n is None:
        n = _STATS_ROWS_BOUND[key] = int(native.load().lhg_conv2d_stats_rows_bound(N, Ho, Wo))
    return n


def _splitk_release(ws):
    """After the launch the workspace was meant for — or after the call failed before reaching it: the library must not keep the pointer
    (a later launch would write through it into memory the allocator has handed to somebody else)."""
    if ws is not None:
        call("lhg_gather_gemm_workspace", None, 0)


def _splitk_workspace(M, M_padded, rows_pad, K, taps, device):
    """The split-K slabs of the gather-GEMM launch that follows (lhg_gather_gemm_splitk_floats: 0 for all but the launches that cannot
    fill the chip), allocated HERE — through torch's allocator, so that a hipGraph capture gets them from the graph's pool — and handed to
    the library for its next launch.  Returns the tensor (keep it referenced until the launch is enqueued) or None."""
    key = (M, M_padded, rows_pad, K, taps, _mode())
    n = _SPLITK_FLOATS.get(key)
    if n is None:
        n = _SPLITK_FLOATS[key] = int(native.load().lhg_gather_gemm_splitk_floats(M, M_padded, rows_pad, K, taps))
    if not n:
        return None
    ws = torch.empty((n,), dtype=torch.float32, device=device)
    call("lhg_gather_gemm_workspace", ptr(ws), n)
    return ws


# LHG_EPILOGUE_BN_STATS=0: a conv that feeds a train-mode BatchNorm does not leave the statistics' partial rows behind (lhg_bn_stats runs its
# pass over the tensor, as before ABI 10) — for A/B measurements.
_EPILOGUE_BN_STATS = os.environ.get("LHG_EPILOGUE_BN_STATS", "1") != "0"


def conv2d_forward_raw(x, w, bias, stride, act=ACT_NONE, slope=0.0, scale=None, shift=None, res=None, out=None, planar=False, x_amax=None,
                       measure_out=False, bn_stats=False):
    """y = act((conv(x, w) + bias) * scale + shift + res); no autograd.  ``bn_stats`` (bias-only NHWC calls): the output goes straight into a
    train-mode BatchNorm — the GEMM's epilogue leaves the first stage of the batch statistics behind (lhg_conv2d_forward_stats) and the
    tensor is tagged with the rows for BatchNormTrainFn (lhg_bn_stats_finish instead of a statistics pass over y)."""
    px, N, H, W, Ci, ldx = nhwc(x)
    Co, Ciw, KH, KW = w.shape
    if pad_to(Ciw, 32) != Ci and not (Ciw <= Ci and thin_mode(Ciw, Co, KH, stride)):
        raise ValueError(f"conv2d: input has {Ci} channels, weight expects {Ciw} (padded {pad_to(Ciw, 32)})")
    mode = thin_mode(Ciw, Co, KH, stride) if KH == KW and res is None else 0
    if mode and not (mode == 2 and (scale is not None or shift is not None)) and not (mode == 1 and planar):
        if planar:
            y = _resolve_out(out, (N, Co, H, W), x.device, torch.float32)
            py, ldy = ptr(y), Co
        else:
            y = _resolve_out(out, (N, H, W, Co), x.device)
            py, _, _, _, _, ldy = nhwc(y)
        # measure_out (thin input, NHWC output that feeds a GEMM directly: eval-mode chains): the kernel measures max|y| as the GEMM epilogues do
        y_amax = _out_amax(out, x.device) if (measure_out and mode == 1 and not planar) else None
        call("lhg_conv2d_thin_forward_amax", px, N, H, W, Ciw, ldx, ptr(_raw_weight(w)), Co, KH, py, ldy, ptr(bias), ptr(scale), ptr(shift),
             act, float(slope), int(planar), ptr(y_amax), stream_ptr())
        if y_amax is not None:
            tag_absmax(y, y_amax)
        return y
    wp = pack_weight(w, True)
    Ho, Wo = _conv_out_hw(H, W, KH, stride)
    if planar:
        y = _resolve_out(out, (N, Co, Ho, Wo), x.device, torch.float32)  # NCHW outputs are fp32 in every storage mode
        py, ldy = ptr(y), Co
    else:
        y = _resolve_out(out, (N, Ho, Wo, Co), x.device)
        py, _, _, _, _, ldy = nhwc(y)
    pres, ldres = (None, 0)
    if res is not None:
        pres, _, _, _, _, ldres = nhwc(res)
    native.count_flops(0, 2.0 * N * Ho * Wo * Co * (w.shape[1] * w.shape[2] * w.shape[3]))
    if x_amax is None:
        x_amax = operand_absmax(x)
    # measure_out: the output feeds another GEMM directly (eval-mode chains, conv + activation blocks): the epilogue measures max|y|
    y_amax = _out_amax(out, x.device) if measure_out and not planar else None
    ws = None if planar else _splitk_workspace(N * Ho * Wo, N * Ho * (Wo + 2) if (KH == 3 and KW == 3 and stride == 1) else 0, wp.shape[1], Ci, KH * KW, x.device)
    try:
        if (bn_stats and _EPILOGUE_BN_STATS and not planar and act == ACT_NONE and scale is None and shift is None and res is None
                and ldy == Co and sync_world() == 1):
            bound = _stats_rows_bound(N, Ho, Wo)
            partial = torch.empty((bound * 2 * Co,), dtype=torch.float32, device=x.device)
            rows = ctypes.c_int(0)
            call("lhg_conv2d_forward_stats", px, N, H, W, Ci, ldx, ptr(wp), wp.shape[1], KH, KW, stride, py, Co, ldy, ptr(bias), ptr(x_amax), ptr(y_amax),
                 ptr(partial), ctypes.byref(rows), stream_ptr())
            if rows.value > 0:
                y.__dict__["_lhg_bn_partial"] = (y._version, partial, rows.value, bias.detach() if bias is not None else None)
        else:
            call("lhg_conv2d_forward", px, N, H, W, Ci, ldx, ptr(wp), wp.shape[1], KH, KW, stride, py, Co, ldy,
                 ptr(bias), ptr(scale), ptr(shift), pres, ldres, act, float(slope), int(planar), ptr(x_amax), ptr(y_amax), stream_ptr())
    finally:
        _splitk_release(ws)
    if y_amax is not None:
        tag_absmax(y, y_amax)  # (a view into a shared buffer: the slot bounds the whole buffer, hence the view)
    return y


def conv2d_thin_forward_nchw(X, w, bias, act=ACT_NONE, slope=0.0, scale=None, shift=None, out=None, measure_out=False):
    """y (NHWC) = act((conv(X, w) + bias) * scale + shift) for a thin-INPUT convolution (<= 8 input channels, 3x3 / 1x1, stride 1) that reads
    the reference's NCHW tensor X directly (lhg_conv2d_thin_forward_nchw): no NCHW -> NHWC(32) conversion of the network input.  No autograd
    (eval-mode generator: generatePOH.py:41-70)."""
    N, Ciw, H, W = X.shape
    Co, Ciw2, KH, KW = w.shape
    if Ciw2 != Ciw or KH != KW or thin_mode(Ciw, Co, KH, 1) != 1:
        raise ValueError(f"conv2d_thin_forward_nchw: {Ciw} -> {Co} channels, {KH}x{KW} is not a thin-input convolution of this tensor")
    Xc = X.detach()
    if Xc.dtype != torch.float32 or not Xc.is_contiguous():
        Xc = Xc.float().contiguous()
    y = _resolve_out(out, (N, H, W, Co), X.device)
    py, _, _, _, _, ldy = nhwc(y)
    y_amax = _out_amax(out, X.device) if measure_out else None
    call("lhg_conv2d_thin_forward_nchw", ptr(Xc), N, H, W, Ciw, ptr(_raw_weight(w)), Co, KH, py, ldy, ptr(bias), ptr(scale), ptr(shift),
         act, float(slope), ptr(y_amax), stream_ptr())
    if y_amax is not None:
        tag_absmax(y, y_amax)
    return y


def conv2d_forward_thin_res(x, w, bias, X_nchw, w3, b3, act=ACT_NONE, slope=0.0, scale=None, shift=None, out=None, measure_out=False):
    """y = act((conv3x3(x, w) + bias) * scale + shift + conv1x1(X_nchw, w3) + b3) with the 1x1 shortcut of a thin NCHW tensor (<= 4
    channels) evaluated inside the GEMM's epilogue (lhg_conv2d_forward_thin_res) — the tail of an eval-mode first ResidualBlock
    (ref: neural_network_components.py:22-31) without the shortcut tensor.  Returns None when the launch does not qualify (the caller
    takes the two-kernel route: same bits)."""
    px, N, H, W, Ci, ldx = nhwc(x)
    Co, Ciw, KH, KW = w.shape
    Xc = X_nchw.detach()
    if (_mode() != _F16_SPLIT or _ACT_DTYPE != torch.float32 or KH != 3 or KW != 3 or Co > 64 or pad_to(Ciw, 32) != Ci or H * W < 512
            or Xc.dtype != torch.float32 or not Xc.is_contiguous() or Xc.shape[1] > 4 or tuple(Xc.shape[2:]) != (H, W) or tuple(w3.shape[2:]) != (1, 1)):
        return None
    wp = pack_weight(w, True)
    if wp.shape[1] != 64:
        return None
    y = _resolve_out(out, (N, H, W, Co), x.device)
    py, _, _, _, _, ldy = nhwc(y)
    native.count_flops(0, 2.0 * N * H * W * Co * (Ciw * 9))
    x_amax = operand_absmax(x)
    y_amax = _out_amax(out, x.device) if measure_out else None
    call("lhg_conv2d_forward_thin_res", px, N, H, W, Ci, ldx, ptr(wp), wp.shape[1], 3, 3, 1, py, Co, ldy, ptr(bias), ptr(scale), ptr(shift),
         ptr(Xc), int(Xc.shape[1]), ptr(_raw_weight(w3)), ptr(b3), act, float(slope), ptr(x_amax), ptr(y_amax), stream_ptr())
    if y_amax is not None:
        tag_absmax(y, y_amax)
    return y


def _padded_gy(gy, k_multiple=32):
    """The GEMM K axis must be a multiple of 32 channels: zero-pad narrow gradients (head: 6, critic head: 1)."""
    Cc = gy.shape[-1]
    if Cc % k_multiple == 0:
        return gy
    out = torch.zeros(gy.shape[:-1] + (pad_to(Cc, k_multiple),), dtype=gy.dtype, device=gy.device)
    out[..., :Cc] = gy
    return out


class Conv2dFn(TrackedFunction):
    """y = conv2d(x, w, stride, padding=k//2) + bias.  x NHWC with C padded to 32; w OIHW.

    ``out`` is None, an OutSlot, or the string "feeds_bn": the conv output goes straight into a train-mode
    BatchNorm, whose input-gradient has exactly zero mean per channel, so d loss / d bias == 0 analytically.
    The reference (and the generic path here) would compute rounding noise of ~1e-7 for it; it is returned as
    exact zero (``None`` in a plain backward pass) instead, which saves one full read of gy per such conv."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, out):
        ctx.bias_grad_is_zero = isinstance(out, str) and out in ("feeds_bn", "feeds_bn_pair")
        bn_stats = ctx.bias_grad_is_zero and out == "feeds_bn"  # ("feeds_bn_pair": two stacked batches with separate statistics, BatchNormPairTrainFn)
        if ctx.bias_grad_is_zero:
            out = None
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.has_bias = stride, bias is not None
        note_use(w, ctx.needs_input_grad[1])
        ctx.bias = bias if (bias is not None and ctx.needs_input_grad[2]) else None  # its gradient never passes through autograd
        note_use(ctx.bias)
        gemm = not (w.shape[2] == w.shape[3] and thin_mode(w.shape[1], w.shape[0], w.shape[2], stride))
        return conv2d_forward_raw(x, w, bias, stride, out=out, x_amax=operand_absmax(x) if gemm else None, bn_stats=bn_stats)

    @staticmethod
    def backward(ctx, gy):
        return Conv2dFn._backward(ctx, gy, None)

    @staticmethod
    def _backward(ctx, gy, g_shared):
        """``g_shared``: the gradient that reached x through its other consumer (Conv2dSharedInputFn), added in the GEMM epilogue."""
        x, w = ctx.saved_tensors
        gx = gw = gb = None
        if gy is None:  # only the alias of x was used downstream
            return g_shared, None, None, None, None
        gy = _as_nhwc_view(gy)  # autograd may hand out an expanded (zero-stride) gradient, e.g. from .sum()
        gemm = not (w.shape[2] == w.shape[3] and thin_mode(w.shape[1], w.shape[0], w.shape[2], ctx.stride))
        if ctx.needs_input_grad[0]:
            # the tensor scale of gy for the input-gradient GEMM (the weight gradient scales per channel: operand_chanmax)
            gy_amax = operand_absmax(gy) if gemm and gy.shape[-1] % 32 == 0 else None
            gx = Conv2dInputGradFn.apply(gy, w, ctx.stride, x.shape[1], x.shape[2], x.shape[3], gy_amax, g_shared, _feeds_gemm(x))
        if not param_grads_wanted():  # inside only_input_gradients(): the caller differentiates with respect to activations only
            return gx, None, None, None, None
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(w, (x, gy), lambda slot: conv2d_weight_grad(x, gy, w.shape, ctx.stride, slot))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _bias_grad(ctx.bias, gy, ctx.bias_grad_is_zero, w.shape[0])
        return gx, gw, gb, None, None


class Conv2dSharedInputFn(TrackedFunction):
    """(conv2d(x, w) + bias, x): Conv2dFn for an input that has a second consumer — ``X`` of the reference's ResidualBlock feeds
    convolution_layer_1 and the skip path (ref: neural_network_components.py:22-31).  The second output is x itself; whatever
    gradient arrives through it is added to this conv's input gradient inside the GEMM epilogue (lhg_conv2d_backward_input_add)
    instead of by autograd's accumulation pass over the two full-size gradients.  LHG_FUSE_SKIP_GRAD=0: callers use Conv2dFn."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, out):
        ctx.set_materialize_grads(False)  # an unused output must not cost a zero-filled gradient
        return Conv2dFn.forward(ctx, x, w, bias, stride, out), x

    @staticmethod
    def backward(ctx, gy, g_shared):
        return Conv2dFn._backward(ctx, gy, g_shared)


FUSE_SKIP_GRAD = os.environ.get("LHG_FUSE_SKIP_GRAD", "1") != "0"


def conv2d_shared_input(x, w, bias, stride, out):
    """(y, x') with x' an alias of x for its other consumer — see Conv2dSharedInputFn."""
    if not FUSE_SKIP_GRAD or not (torch.is_grad_enabled() and x.requires_grad):
        return Conv2dFn.apply(x, w, bias, stride, out), x
    y, xs = Conv2dSharedInputFn.apply(x, w, bias, stride, out)
    return y, _inherit_chanmax(_inherit_absmax(xs, x), x, alias=True)


class Conv2dInputGradFn(TrackedFunction):
    """gx = d conv2d / d x contracted with gy.  Returns (N, H, W, pad32(Ci))."""

    @staticmethod
    def forward(ctx, gy, w, stride, H, W, Cx, gy_amax=None, res=None, measure_out=False):
        """``res``: a second gradient of the same tensor (N, H, W, >= Ci channels), added to the result in the GEMM epilogue.
        ``measure_out``: the result is the operand of another backward GEMM (see _feeds_gemm): the epilogue measures max|gx|."""
        ctx.save_for_backward(gy, w)
        ctx.stride = stride
        note_use(w, ctx.needs_input_grad[1])
        Co, Ci, KH, KW = w.shape
        if res is not None:
            res = _as_nhwc_view(res)
            if tuple(res.shape[:3]) != (gy.shape[0], H, W) or res.shape[3] < Ci:
                raise ValueError(f"conv2d input-grad: added gradient of shape {tuple(res.shape)} does not cover ({gy.shape[0]}, {H}, {W}, {Ci})")
        if KH == KW and gy.shape[-1] >= Co and thin_mode(Ci, Co, KH, stride):
            gyv = _as_nhwc_view(gy)  # keep the (possibly temporary) dense copy alive until the launch
            pg, N, _, _, _, ldg = nhwc(gyv)
            gx = new_nhwc(N, H, W, Cx, gy.device)
            if Cx > Ci:
                gx[..., Ci:].zero_()
            call("lhg_conv2d_thin_backward_input", pg, N, H, W, Co, ldg, ptr(_raw_weight(w)), Ci, KH, ptr(gx), Cx, stream_ptr())
            if res is not None:
                gx[..., :Ci] += res[..., :Ci]
            return gx
        gyp = _padded_gy(gy)
        pg, N, Ho, Wo, Cg, ldg = nhwc(gyp)
        Co, Ci, KH, KW = w.shape
        wp = pack_weight(w, False, 32)
        if pad_to(Co, 32) != Cg:
            raise ValueError(f"conv2d input-grad: gy has {Cg} channels, packed K is {pad_to(Co, 32)}")
        gx = new_nhwc(N, H, W, Cx, gy.device)
        if Cx > Ci:
            gx[..., Ci:].zero_()
        pgx, _, _, _, _, ldgx = nhwc(gx)
        native.count_flops(0, 2.0 * N * Ho * Wo * Co * (w.shape[1] * w.shape[2] * w.shape[3]))
        if gy_amax is None or gyp is not gy:
            gy_amax = operand_absmax(gyp)
        pres, ldres = (nhwc(res)[0], nhwc(res)[5]) if res is not None else (None, 0)
        # max|gx| from the epilogue (the added gradient included): gx may be the operand of the next backward GEMM — the gradient of a
        # skip-concatenation buffer feeds the transposed conv's backward, a critic block's the previous block's — without a pass of its own
        gx_amax = fused_absmax_slot(gy.device) if (measure_out and _EPILOGUE_GX_AMAX) else None
        ws = _splitk_workspace(N * H * W, N * H * (W + 2) if (KH == 3 and KW == 3 and stride == 1) else 0, wp.shape[1], Cg, KH * KW, gy.device)
        try:
            call("lhg_conv2d_backward_input_add_amax", pg, N, H, W, Cg, ldg, ptr(wp), wp.shape[1], KH, KW, stride, pgx, Ci, ldgx, pres, ldres,
                 ptr(gy_amax), ptr(gx_amax), stream_ptr())
        finally:
            _splitk_release(ws)
        return tag_absmax(gx, gx_amax)

    @staticmethod
    def backward(ctx, ggx):
        gy, w = ctx.saved_tensors
        g_gy = g_w = None
        if ctx.needs_input_grad[0]:
            g_gy = Conv2dFn.apply(ggx, w, None, ctx.stride, None)
        if ctx.needs_input_grad[1] and param_grads_wanted():
            g_w = _weight_grad(w, (ggx, gy), lambda slot: conv2d_weight_grad(ggx, gy, w.shape, ctx.stride, slot))
        n = len(ctx.needs_input_grad)  # 6 to 9 arguments were passed
        grads = (g_gy, g_w, None, None, None, None, None, ggx if n > 7 and ctx.needs_input_grad[7] else None, None)
        return grads[:n]


def _feeds_gemm(x) -> bool:
    """Is the gradient of ``x`` handed to another backward GEMM as it leaves the input-gradient GEMM?  True for a skip-concatenation
    buffer (its gradient is sliced by CatViewsFn.backward: one slice is the transposed conv's gy operand) — measuring max|gx| in the
    epilogue of EVERY input-gradient launch was tried and cost more than the four lhg_absmax passes it saves (+0.13 ms per step)."""
    fn = getattr(x, "grad_fn", None)
    return fn is not None and fn.name() == "CatViewsFnBackward"


def _timing_switch(name: str) -> bool:
    """LHG_TIMING_* switches make the step compute WRONG gradients on purpose (what would it cost without ...): honoured only when the
    process also sets LHG_ALLOW_WRONG_RESULTS=1 (tools/cpu_slack.py, tools/time_wgrad.py do), refused loudly otherwise."""
    if os.environ.get(name, "0") != "1":
        return False
    if os.environ.get("LHG_ALLOW_WRONG_RESULTS", "0") != "1":
        raise RuntimeError(f"{name}=1 produces wrong gradients (a timing experiment): set LHG_ALLOW_WRONG_RESULTS=1 as well, or unset it")
    import warnings

    warnings.warn(f"{name}=1: gradients of this process are WRONG on purpose (timing experiment)")
    return True


_TIMING_SKIP_WGRAD = _timing_switch("LHG_TIMING_SKIP_WGRAD")
_TIMING_FAKE_CHANMAX = _timing_switch("LHG_TIMING_FAKE_CHANMAX")


def conv2d_weight_grad_raw(x, gy, wshape, stride, slot=None):
    """gw (OIHW) = sum over pixels of x (gathered) outer gy; no autograd.  With ``slot`` the slab reduction accumulates straight into
    it (the parameter's view of the flat gradient buffer) and nothing is returned."""
    Co, Ci, KH, KW = wshape
    if KH == KW and gy.shape[-1] >= Co and x.shape[-1] >= Ci and thin_mode(Ci, Co, KH, stride):
        px, N, H, W, _, ldx = nhwc(x)
        gyv = _as_nhwc_view(gy)
        pg, _, _, _, _, ldg = nhwc(gyv)
        nbytes = int(native.load().lhg_conv2d_thin_wgrad_workspace(N, H, W, Ci, Co, KH))
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device=x.device)
        gw = torch.empty(tuple(wshape), dtype=torch.float32, device=x.device)
        call("lhg_conv2d_thin_backward_weight", px, N, H, W, Ci, ldx, pg, Co, ldg, KH, ptr(gw), ptr(ws), nbytes, stream_ptr())
        if slot is None:
            return gw
        slot.add_(gw)
        return None
    if _TIMING_SKIP_WGRAD:  # timing experiments only (tools/cpu_slack.py): what the step costs without the MFMA weight gradients
        return None if slot is not None else torch.zeros(tuple(wshape), dtype=torch.float32, device=x.device)
    gyp = _padded_gy(gy, 4)
    px, N, H, W, Cx, ldx = nhwc(x)
    pg, _, _, _, Cg, ldg = nhwc(gyp)
    lib = native.load()
    native.count_flops(1, 2.0 * N * gy.shape[1] * gy.shape[2] * Co * Ci * KH * KW)
    # fp16-split mode: per-CHANNEL max|.| of both operands (not the scalar tensor maxima the forward / input-gradient GEMMs scale by)
    x_cmax, gy_cmax = operand_chanmax(x), operand_chanmax(gyp)
    gw = slot if slot is not None else torch.empty(tuple(wshape), dtype=torch.float32, device=x.device)
    if not gw.is_contiguous():
        raise ValueError("weight-gradient slot must be contiguous")
    if Ci % 4 == 0 and Co % 4 == 0:
        # GEMM + reduction of the K splits behind one call (in one launch where the tap-fused kernel reduces its slabs itself); the
        # operands may carry padding channels behind the weight's (their pixel strides say so)
        nbytes = int(lib.lhg_conv2d_backward_weight_workspace(N, H, W, Ci, Co, KH, KW, stride))
        ws = _wgrad_workspace(nbytes, x.device)
        call("lhg_conv2d_backward_weight_into", px, N, H, W, Ci, ldx, pg, Co, ldg, KH, KW, stride, ptr(gw), int(slot is not None), ptr(ws), nbytes,
             ptr(x_cmax), ptr(gy_cmax), stream_ptr())
        return None if slot is not None else gw
    # channel counts that are not multiples of four: the operands' padded counts through the per-tap GEMM, the true ones in the reduction
    S = lib.lhg_conv2d_wgrad_splits(N, H, W, Cx, Cg, KH, KW, stride)
    ci_pad, co_pad = pad_to(Cx, 64), pad_to(Cg, 64)
    slabs = torch.empty((S, KH * KW, ci_pad, co_pad), dtype=torch.float32, device=x.device)
    call("lhg_conv2d_backward_weight", px, N, H, W, Cx, ldx, pg, Cg, ldg, KH, KW, stride, ptr(slabs), S, ci_pad, co_pad,
         ptr(x_cmax), ptr(gy_cmax), stream_ptr())
    call("lhg_wgrad_reduce", ptr(slabs), S, KH * KW, ci_pad, co_pad, ptr(gw), Co, Ci, 1, int(slot is not None), stream_ptr())
    return None if slot is not None else gw


def _wgrad_workspace(nbytes, device):
    """Scratch of a weight-gradient call (partial slabs + tickets): a fresh allocation per call — the caching allocator hands the block
    back to the stream that used it (the weight-gradient stream), 256-byte aligned like every block it returns."""
    return torch.empty((max(nbytes, 256) // 4 + 1,), dtype=torch.float32, device=device)


def conv2d_weight_grad(x, gy, wshape, stride, slot):
    """Differentiable op without a slot, raw accumulation with one."""
    if slot is None:
        return Conv2dWeightGradFn.apply(x, gy, wshape, stride)
    return conv2d_weight_grad_raw(x, gy, wshape, stride, slot)


def _bias_grad(bias, gy, is_zero, Co):
    """d loss / d bias = sum of gy over pixels.  In a plain backward pass into a registered slot the sum is accumulated straight into
    the slot (None goes back to autograd: no AccumulateGrad add launch); an analytically-zero gradient (conv in front of a train-mode
    BatchNorm) costs nothing at all."""
    if is_zero:
        if not torch.is_grad_enabled():
            note_contribution(bias)
            return None
        return torch.zeros(Co, dtype=torch.float32, device=gy.device)
    slot = _small_grad_slot(bias)
    if slot is None:
        return channel_sum(gy)  # autograd accumulates it (and its post-accumulate hook reports the arrival)
    if gy.shape[-1] % 4 == 0 and not _SIDE_BIAS:
        channel_sum_into(gy, slot)
        note_contribution(bias)
        return None
    if gy.shape[-1] % 4 == 0:
        # nothing in backward waits for a bias gradient before the optimiser step: like the weight gradients it is accumulated on the
        # second stream, beside the main chain (a pass over gy: 0.5 ms per step off the chain the step waits for)
        return _weight_grad(bias, (gy,), lambda s_: channel_sum_into(gy, s_) if s_ is not None else channel_sum(gy))
    slot.add_(gy.sum(dim=(0, 1, 2), dtype=torch.float32))  # narrow heads (6 / 1 channels): a few hundred KB
    note_contribution(bias)
    return None


class Conv2dWeightGradFn(Function):
    """gw (OIHW) = sum over pixels of x (gathered) outer gy."""

    @staticmethod
    def forward(ctx, x, gy, wshape, stride):
        ctx.save_for_backward(x, gy)
        ctx.stride, ctx.wshape = stride, tuple(wshape)
        return conv2d_weight_grad_raw(x, gy, ctx.wshape, stride)

    @staticmethod
    def backward(ctx, ggw):
        x, gy = ctx.saved_tensors
        g_x = g_gy = None
        if ctx.needs_input_grad[0]:
            g_x = Conv2dInputGradFn.apply(gy, ggw, ctx.stride, x.shape[1], x.shape[2], x.shape[3])
        if ctx.needs_input_grad[1]:
            g_gy = Conv2dFn.apply(x, ggw, None, ctx.stride, None)
        return g_x, g_gy, None, None


class ConvTranspose2x2Fn(TrackedFunction):
    """y = conv_transpose2d(x, w, stride=2) + bias with kernel 2.  w IOHW (Cin, Cout, 2, 2)."""

    @staticmethod
    def forward(ctx, x, w, bias, out):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        note_use(w, ctx.needs_input_grad[1])
        ctx.bias = bias if (bias is not None and ctx.needs_input_grad[2]) else None
        note_use(ctx.bias)
        px, N, H, W, Ci, ldx = nhwc(x)
        Ciw, Co, KH, KW = w.shape
        assert (KH, KW) == (2, 2) and Ciw == Ci, "conv_transpose2x2: weight/input mismatch"
        wp = pack_weight(w, False)  # rows = Cout, K = Cin
        y = _resolve_out(out, (N, 2 * H, 2 * W, Co), x.device)
        py, _, _, _, _, ldy = nhwc(y)
        native.count_flops(0, 2.0 * N * H * W * 4 * Ci * Co)
        ctx.x_amax = operand_absmax(x)
        y_amax = _out_amax(out, x.device)  # the output is a conv input (through the concatenation buffer)
        call("lhg_conv_transpose2x2_forward", px, N, H, W, Ci, ldx, ptr(wp), wp.shape[1], py, Co, ldy, ptr(bias), ptr(ctx.x_amax), ptr(y_amax),
             stream_ptr())
        if y_amax is not None:
            tag_absmax(y, y_amax)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = _as_nhwc_view(gy)
        gx = gw = gb = None
        Ci, Co = w.shape[0], w.shape[1]
        gy_amax = operand_absmax(gy)
        if ctx.needs_input_grad[0]:
            pg, N, H2, W2, Cg, ldg = nhwc(gy)
            wp = pack_weight(w, True)  # rows = Cin, K = Cout
            gx = new_nhwc(N, H2 // 2, W2 // 2, Ci, gy.device)
            pgx, _, _, _, _, ldgx = nhwc(gx)
            native.count_flops(0, 2.0 * N * (H2 // 2) * (W2 // 2) * 4 * Ci * Co)
            gx_amax = None  # (gx goes to a BatchNorm backward, which measures what it writes: nothing to measure here)
            ws = _splitk_workspace(N * (H2 // 2) * (W2 // 2), 0, wp.shape[1], Cg, 4, gy.device)  # (a four-tap stride-2 gather over N H W pixels)
            try:
                call("lhg_conv_transpose2x2_backward_input_amax", pg, N, H2 // 2, W2 // 2, Cg, ldg, ptr(wp), wp.shape[1], pgx, Ci, ldgx, ptr(gy_amax),
                     ptr(gx_amax), stream_ptr())
            finally:
                _splitk_release(ws)
            tag_absmax(gx, gx_amax)
        if not param_grads_wanted():
            return gx, None, None, None
        if ctx.needs_input_grad[1]:
            def wgrad(slot):
                px, N, H, W, Cx, ldx = nhwc(x)
                pg, _, _, _, Cg, ldg = nhwc(gy)
                nbytes = int(native.load().lhg_conv_transpose2x2_backward_weight_workspace(N, H, W, Cx, Cg))
                ws = _wgrad_workspace(nbytes, x.device)
                native.count_flops(1, 2.0 * N * H * W * 4 * Ci * Co)
                out = slot if slot is not None else torch.empty(w.shape, dtype=torch.float32, device=x.device)
                call("lhg_conv_transpose2x2_backward_weight_into", px, N, H, W, Cx, ldx, pg, Cg, ldg, ptr(out), int(slot is not None), ptr(ws), nbytes,
                     ptr(operand_chanmax(x)), ptr(operand_chanmax(gy)), stream_ptr())
                return None if slot is not None else out

            gw = _weight_grad(w, (x, gy), wgrad)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _bias_grad(ctx.bias, gy, False, Co)
        return gx, gw, gb, None


def channel_sum(t):
    """sum over (N,H,W) of an NHWC tensor -> (C,).  Used for bias gradients (linear in t)."""
    if t.shape[-1] % 4:  # heads with 6 / 1 channels: a few hundred KB
        return t.sum(dim=(0, 1, 2), dtype=torch.float32)
    return ChannelSumFn.apply(t)


def channel_sum_into(t, slot):
    """slot += sum over (N,H,W) of t (C % 4 == 0): the bias gradient accumulated by the reduction kernel itself."""
    p, N, H, W, Cc, ld = nhwc(t)
    if not (_FUSED_BN or _SIDE_BIAS):  # on the main chain and not asked for: the two-launch form
        call("lhg_channel_sum", p, N * H * W, Cc, ld, ptr(slot), 1, ptr(torch.empty((2048 * Cc,), dtype=torch.float32, device=t.device)), stream_ptr())
        return
    ws, tickets = fused_scratch(t.device)
    call("lhg_channel_sum_fused", p, N * H * W, Cc, ld, ptr(slot), 1, ws, tickets, stream_ptr())


class ChannelSumFn(Function):
    @staticmethod
    def forward(ctx, t):
        p, N, H, W, Cc, ld = nhwc(t)
        ctx.shape = t.shape
        out = torch.empty((Cc,), dtype=torch.float32, device=t.device)
        ws, tickets = fused_scratch(t.device)
        call("lhg_channel_sum_fused", p, N * H * W, Cc, ld, ptr(out), 0, ws, tickets, stream_ptr())
        return out

    @staticmethod
    def backward(ctx, g):
        return g.to(_ACT_DTYPE).view(1, 1, 1, -1).expand(ctx.shape)


# --------------------------------------------------------------------------- activations
class ActGradFn(Function):
    """g * act'(y) with the mask read from the forward output (linear in g)."""

    @staticmethod
    def forward(ctx, g, y, act, slope):
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        pg, N, H, W, Cc, ldg = nhwc(g)
        py, _, _, _, _, ldy = nhwc(y)
        out = new_nhwc(N, H, W, Cc, g.device)
        call("lhg_act_backward", pg, ldg, py, ldy, N * H * W, Cc, act, float(slope), ptr(out), Cc, stream_ptr())
        return out

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        return ActGradFn.apply(gg, y, ctx.act, ctx.slope), None, None, None


class ConvBiasActFn(TrackedFunction):
    """y = act(conv2d(x, w) + bias) with the activation fused in the GEMM epilogue
    (critic block1, ref: discriminator.py:16-19)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, act, slope):
        y = conv2d_forward_raw(x, w, bias, stride, act=act, slope=slope, measure_out=True)
        ctx.save_for_backward(x, w, y)
        ctx.stride, ctx.act, ctx.slope, ctx.has_bias = stride, act, slope, bias is not None
        note_use(w, ctx.needs_input_grad[1])
        ctx.bias = bias if (bias is not None and ctx.needs_input_grad[2]) else None
        note_use(ctx.bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        g = ActGradFn.apply(_as_nhwc_view(gy), y, ctx.act, ctx.slope)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = Conv2dInputGradFn.apply(g, w, ctx.stride, x.shape[1], x.shape[2], x.shape[3])
        if not param_grads_wanted():
            return gx, None, None, None, None, None
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(w, (x, g), lambda slot: conv2d_weight_grad(x, g, w.shape, ctx.stride, slot))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = _bias_grad(ctx.bias, g, False, w.shape[0])
        return gx, gw, gb, None, None, None


# --------------------------------------------------------------------------- synchronised batch statistics (data-parallel replicas)
# The reference is single-device: at a global batch of W x B it normalises every train-mode BatchNorm over all W x B samples, takes the
# focal loss's two max normalisers over the whole batch and the TV difference of whole-batch means (neural_network_components.py:23-24,
# loss_func.py:94-98, 152-157).  Replicas of B samples each reproduce that only if those statistics are all-reduced:
# ``set_sync_batch_stats(True)`` (watermelon.configure(sync_batch_stats=True)) switches every BatchNorm of the op layer and
# poh_ops.ReconLossFn to global-batch statistics when torch.distributed is initialised with more than one rank.  Per BatchNorm layer that
# is one small all-reduce in forward (mean, E[x^2]), one in backward (sum g, sum g xhat) and one in the double backward (five sums);
# parameter gradients stay LOCAL sums (the gradient all-reduce averages them), as in torch.nn.SyncBatchNorm.
_SYNC_STATS = False


_SYNC_GROUP = None  # process group the synchronised statistics are reduced over (None: the default group)


def set_sync_batch_stats(on: bool, group=None) -> None:
    """``group``: the process group whose ranks share their batch statistics — pass the group the GradSynchronizer averages over when it
    is not the default one (a sub-group synchronizer with default-group statistics would mix replicas that do not train together)."""
    global _SYNC_STATS, _SYNC_GROUP
    _SYNC_STATS = bool(on)
    _SYNC_GROUP = group if on else None


def sync_world() -> int:
    """Ranks whose batches are normalised together (1: per-replica statistics)."""
    if not _SYNC_STATS:
        return 1
    import torch.distributed as dist

    return dist.get_world_size(_SYNC_GROUP) if dist.is_available() and dist.is_initialized() else 1


def all_reduce_(t, op="sum"):
    import torch.distributed as dist

    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM, group=_SYNC_GROUP)
    return t


def assert_equal_batches(samples: int, device) -> None:
    """The synchronised statistics weight every replica equally (mean of means, 1 / (pixels * world) in the backward sums): that is
    the global-batch result only when every rank holds the SAME number of samples.  One small MAX all-reduce and one host read per
    step (the option is a parity mode, off by default) — a ragged last batch raises instead of training on wrong statistics
    (torch.nn.SyncBatchNorm all-gathers the counts for the same reason)."""
    if sync_world() == 1:
        return
    t = torch.tensor([float(samples), -float(samples)], device=device)
    all_reduce_(t, "max")
    hi, lo = t.tolist()
    if hi != -lo:
        raise RuntimeError(f"sync_batch_stats needs equal per-rank batches: this rank holds {samples} samples, the ranks hold between {int(-lo)} and {int(hi)}")


def _global_batch_stats(stats, Cc, pixels, world, running_mean, running_var):
    """Local (mean, invstd) -> global ones in place: the replicas hold equal sample counts, so mean = avg(mean_r) and
    E[x^2] = avg(var_r + mean_r^2); combined in float64 (var = E[x^2] - mean^2 cancels).  Running statistics as nn.BatchNorm2d keeps them."""
    mean_l = stats[:Cc].double()
    var_l = stats[Cc:].double().pow(-2) - BN_EPS
    buf = torch.stack((mean_l, var_l + mean_l * mean_l))
    all_reduce_(buf).div_(world)
    mean, var = buf[0], (buf[1] - buf[0] * buf[0]).clamp_min_(0)
    stats[:Cc] = mean.float()
    stats[Cc:] = (var + BN_EPS).rsqrt().float()
    n = float(pixels * world)
    if running_mean is not None:
        running_mean.mul_(1 - BN_MOMENTUM).add_(mean.float(), alpha=BN_MOMENTUM)
    if running_var is not None:
        running_var.mul_(1 - BN_MOMENTUM).add_((var * (n / max(n - 1, 1))).float(), alpha=BN_MOMENTUM)


# --------------------------------------------------------------------------- batch norm
def _bn_ws(Cc, device, mult=8192):
    return torch.empty((mult * Cc,), dtype=torch.float32, device=device)


class BatchNormTrainFn(TrackedFunction):
    """y = act(BN_batchstats(x) * gamma + beta [+ res]); updates running stats in place.
    ref: F.batch_norm(training=True) via nn.LazyBatchNorm2d, neural_network_components.py:27-31;
    nn.BatchNorm2d discriminator.py:39."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, res, act, slope, out):
        px, N, H, W, Cc, ldx = nhwc(x)
        pixels = N * H * W
        stats = torch.empty((2 * Cc,), dtype=torch.float32, device=x.device)
        world = sync_world()
        ctx.world = world
        y = _resolve_out(out, (N, H, W, Cc), x.device)
        py, _, _, _, _, ldy = nhwc(y)
        pres, ldres = (None, 0)
        if res is not None:
            pres, _, _, _, _, ldres = nhwc(res)
        y_amax = _out_amax(out, x.device)  # max|y| measured by the kernel that writes y: the next conv's GEMMs need no pass of their own
        if world > 1:  # statistics over the global batch: the sums pass through the host's all-reduce between the two kernels
            call("lhg_bn_stats", px, pixels, Cc, ldx, ptr(stats), None, None, BN_MOMENTUM, BN_EPS, ptr(_bn_ws(Cc, x.device, 4104)), stream_ptr())
            _global_batch_stats(stats, Cc, pixels, world, running_mean, running_var)
            call("lhg_bn_apply", px, ldx, pixels, Cc, ptr(stats), ptr(gamma), ptr(beta), pres, ldres, act, float(slope), py, ldy, ptr(y_amax),
                 stream_ptr())
        else:
            # statistics (finished in the launch that sums them), running statistics and the apply pass behind ONE call, two launches; the
            # weight gradient of the conv that reads y wants per-channel maxima of it: finished by the apply launch (training passes only)
            cm = chanmax_dest(Cc, pixels, x.device, out) if any(ctx.needs_input_grad) else None
            if _FUSED_BN:
                ws, tickets = fused_scratch(x.device)
                call("lhg_bn_forward_train", px, ldx, pixels, Cc, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), BN_MOMENTUM, BN_EPS,
                     pres, ldres, act, float(slope), py, ldy, ptr(stats), ptr(y_amax), cm.ptr if cm else None, cm.finish if cm else 0, ws, tickets,
                     stream_ptr())
            else:
                rows = x.__dict__.get("_lhg_bn_partial")  # left by the conv GEMM that wrote x (conv2d_forward_raw, bn_stats)
                if rows is not None and rows[0] == x._version and ldx == Cc:
                    call("lhg_bn_stats_finish", ptr(rows[1]), rows[2], ptr(rows[3]), pixels, Cc, ptr(stats), ptr(running_mean), ptr(running_var),
                         BN_MOMENTUM, BN_EPS, stream_ptr())
                    del x.__dict__["_lhg_bn_partial"]  # folded: the rows (up to 19 MB per 384^2 layer) go back to the allocator now, not after backward
                else:
                    call("lhg_bn_stats", px, pixels, Cc, ldx, ptr(stats), ptr(running_mean), ptr(running_var), BN_MOMENTUM, BN_EPS,
                         ptr(_bn_ws(Cc, x.device, 4104)), stream_ptr())
                if cm is not None:
                    call("lhg_bn_apply_chanmax", px, ldx, pixels, Cc, ptr(stats), ptr(gamma), ptr(beta), pres, ldres, act, float(slope), py, ldy,
                         ptr(y_amax), cm.ptr, stream_ptr())
                else:
                    call("lhg_bn_apply", px, ldx, pixels, Cc, ptr(stats), ptr(gamma), ptr(beta), pres, ldres, act, float(slope), py, ldy, ptr(y_amax),
                         stream_ptr())
            if cm is not None:
                cm.commit(y)
        tag_absmax(y, y_amax)
        ctx.save_for_backward(x, y, gamma, stats)
        ctx.act, ctx.slope, ctx.has_res = act, slope, res is not None
        ctx.beta = beta if ctx.needs_input_grad[2] else None
        ctx.beta_value = beta  # the backward kernels recompute the activation mask from x (no residual: one tensor read less per pass)
        note_use(gamma, ctx.needs_input_grad[1])
        note_use(ctx.beta)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, gamma, stats = ctx.saved_tensors
        gy = _as_nhwc_view(gy)
        s_gamma = _small_grad_slot(gamma) if ctx.needs_input_grad[1] else None
        s_beta = _small_grad_slot(ctx.beta)
        if s_gamma is not None and s_beta is not None and param_grads_wanted():
            # plain backward into the flat gradient buffer: the kernel adds d gamma / d beta to their slots itself
            gx, gres = bn_backward_raw(gy, x, y, gamma, stats, ctx.act, ctx.slope, ctx.has_res, s_gamma, s_beta, True, ctx.beta_value, ctx.world)
            note_contribution(gamma)
            note_contribution(ctx.beta)
            return gx, None, None, None, None, (gres if ctx.has_res else None), None, None, None
        gx, gres, ggamma, gbeta = BatchNormGradFn.apply(gy, x, y, gamma, stats, ctx.act, ctx.slope, ctx.has_res, ctx.world)
        return gx, ggamma, gbeta, None, None, (gres if ctx.has_res else None), None, None, None


class BatchNormPairTrainFn(TrackedFunction):
    """y = act(BN(x) * gamma + beta) for a batch that is TWO batches of the reference stacked on dim 0 — D(real) and D(fake) of the critic
    update (ref: watermelon.py:243-244, two forward calls of the same module) run as one pass: the convolutions see both halves in one
    GEMM, and this op normalises every half with ITS OWN batch statistics, first half first (statistics, running-statistics update,
    apply — then the same for the second half), exactly what the two calls compute.  One tensor out, one node in the graph; the backward
    does the same per half and adds both halves' d gamma / d beta.  Plain backward only (the gradient penalty's pass is a call of its
    own through BatchNormTrainFn); per-replica statistics only."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, act, slope):
        px, N, H, W, Cc, ldx = nhwc(x)
        if N % 2:
            raise ValueError("BatchNormPairTrainFn: the batch must be two stacked batches of equal size")
        if sync_world() > 1:
            raise NotImplementedError("BatchNormPairTrainFn: per-replica statistics only")
        half = (N // 2) * H * W
        es = x.element_size()
        stats = torch.empty((2, 2 * Cc), dtype=torch.float32, device=x.device)
        y = new_nhwc(N, H, W, Cc, x.device)
        py, _, _, _, _, ldy = nhwc(y)
        y_amax = fused_absmax_slot(x.device)  # both halves max-accumulate into the one slot
        # per-channel maxima: both halves' partial rows one behind the other in ONE buffer, finished together when a weight gradient asks
        rows = int(native.load().lhg_chanmax_partial_rows(half, Cc)) if (any(ctx.needs_input_grad) and chanmax_wanted(Cc)) else 0
        part = torch.empty((2 * rows * Cc,), dtype=torch.float32, device=x.device) if rows else None
        ws, tickets = fused_scratch(x.device) if _FUSED_BN else (None, None)
        for h in range(2):
            xo, yo = px + h * half * ldx * es, py + h * half * ldy * es
            rows_h = None if part is None else part.data_ptr() + 4 * h * rows * Cc
            if _FUSED_BN:
                call("lhg_bn_forward_train", xo, ldx, half, Cc, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), BN_MOMENTUM, BN_EPS,
                     None, 0, act, float(slope), yo, ldy, stats[h].data_ptr(), ptr(y_amax), rows_h, 0, ws, tickets, stream_ptr())
                continue
            call("lhg_bn_stats", xo, half, Cc, ldx, stats[h].data_ptr(), ptr(running_mean), ptr(running_var), BN_MOMENTUM, BN_EPS,
                 ptr(_bn_ws(Cc, x.device, 4104)), stream_ptr())
            if rows_h is not None:
                call("lhg_bn_apply_chanmax", xo, ldx, half, Cc, stats[h].data_ptr(), ptr(gamma), ptr(beta), None, 0, act, float(slope), yo, ldy,
                     ptr(y_amax), rows_h, stream_ptr())
            else:
                call("lhg_bn_apply", xo, ldx, half, Cc, stats[h].data_ptr(), ptr(gamma), ptr(beta), None, 0, act, float(slope), yo, ldy, ptr(y_amax),
                     stream_ptr())
        tag_absmax(y, y_amax)
        if part is not None:
            src = ChanMaxSource(Cc, x.device)
            src.pending.append((0, Cc, part, 2 * half, 2 * rows))
            tag_chanmax_source(y, src)
        ctx.save_for_backward(x, y, gamma, stats)
        ctx.act, ctx.slope = act, slope
        ctx.beta = beta if ctx.needs_input_grad[2] else None
        ctx.beta_value = beta
        note_use(gamma, ctx.needs_input_grad[1])
        note_use(ctx.beta)
        return y

    @staticmethod
    def backward(ctx, gy):
        if torch.is_grad_enabled():
            raise NotImplementedError("BatchNormPairTrainFn is not twice differentiable (the gradient penalty uses BatchNormTrainFn)")
        x, y, gamma, stats = ctx.saved_tensors
        gy = _as_nhwc_view(gy)
        pg, N, H, W, Cc, ldg = nhwc(gy)
        px, _, _, _, _, ldx = nhwc(x)
        py, _, _, _, _, ldy = nhwc(y)
        half, es = (N // 2) * H * W, x.element_size()
        mask_from_x = ctx.beta_value is not None and ctx.act in (ACT_RELU, ACT_LEAKY) and _ACT_DTYPE == torch.float32
        s_gamma = _small_grad_slot(gamma) if ctx.needs_input_grad[1] else None
        s_beta = _small_grad_slot(ctx.beta)
        into_slots = s_gamma is not None and s_beta is not None and param_grads_wanted()
        ggamma = s_gamma if into_slots else torch.empty((Cc,), dtype=torch.float32, device=gy.device)
        gbeta = s_beta if into_slots else torch.empty((Cc,), dtype=torch.float32, device=gy.device)
        gx = new_nhwc(N, H, W, Cc, gy.device)
        pgx = gx.data_ptr()
        gx_amax = fused_absmax_slot(gy.device)
        rows = int(native.load().lhg_chanmax_partial_rows(half, Cc)) if chanmax_wanted(Cc) else 0
        part = torch.empty((2 * rows * Cc,), dtype=torch.float32, device=gy.device) if rows else None
        ws, tickets = fused_scratch(gy.device) if _FUSED_BN else (None, None)
        for h in range(2):
            off = h * half
            acc = 1 if (into_slots or h == 1) else 0  # the second half adds to the first's sums
            rows_h = None if part is None else part.data_ptr() + 4 * h * rows * Cc
            gyo, xo, yo = pg + off * ldg * es, px + off * ldx * es, None if mask_from_x else py + off * ldy * es
            beta_p = ptr(ctx.beta_value) if mask_from_x else None
            if _FUSED_BN:
                call("lhg_bn_backward_fused", gyo, ldg, xo, ldx, yo, ldy, half, Cc, stats[h].data_ptr(), ptr(gamma), beta_p, ctx.act, float(ctx.slope),
                     pgx + off * Cc * es, Cc, None, Cc, ptr(ggamma), ptr(gbeta), acc, ptr(gx_amax), None, rows_h, None, 0, ws, tickets, stream_ptr())
                continue
            args = (gyo, ldg, xo, ldx, yo, ldy, half, Cc, stats[h].data_ptr(), ptr(gamma), ctx.act, float(ctx.slope), pgx + off * Cc * es, Cc, None, Cc,
                    ptr(ggamma), ptr(gbeta), acc, ptr(_bn_ws(Cc, gy.device)), ptr(gx_amax), beta_p)
            if rows_h is not None:
                call("lhg_bn_backward_chanmax", *args, rows_h, None, None, stream_ptr())
            else:
                call("lhg_bn_backward", *args, stream_ptr())
        tag_absmax(gx, gx_amax)
        if part is not None:
            src = ChanMaxSource(Cc, gy.device)
            src.pending.append((0, Cc, part, 2 * half, 2 * rows))
            tag_chanmax_source(gx, src)
        if into_slots:
            note_contribution(gamma)
            note_contribution(ctx.beta)
            return gx, None, None, None, None, None, None
        return gx, ggamma, gbeta, None, None, None, None


def bn_backward_raw(gy, x, y, gamma, stats, act, slope, want_res, ggamma, gbeta, accumulate, beta=None, world=1):
    """lhg_bn_backward; ggamma / gbeta are written (or, with ``accumulate``, added to).  Returns (gx, gres).  With ``beta`` (and no
    residual, ReLU / LeakyReLU) the kernels recompute the activation mask from x instead of reading y.  ``world`` > 1: `stats` are
    global-batch statistics — the two per-channel sums are all-reduced between the kernel's halves, ggamma / gbeta take the LOCAL sums."""
    pg, N, H, W, Cc, ldg = nhwc(gy)
    px, _, _, _, _, ldx = nhwc(x)
    py, _, _, _, _, ldy = nhwc(y)
    if beta is not None and not want_res and act in (ACT_RELU, ACT_LEAKY) and _ACT_DTYPE == torch.float32:
        py = None
    gx = new_nhwc(N, H, W, Cc, gy.device)
    gres = new_nhwc(N, H, W, Cc, gy.device) if want_res else None
    gx_amax = fused_absmax_slot(gy.device)  # gx is the gy of the preceding conv's two backward GEMMs
    if world > 1:
        sums = torch.empty((2 * Cc,), dtype=torch.float32, device=gy.device)
        call("lhg_bn_backward_sums", pg, ldg, px, ldx, py, ldy, N * H * W, Cc, ptr(stats), ptr(gamma), act, float(slope), ptr(sums),
             ptr(_bn_ws(Cc, gy.device)), ptr(beta), stream_ptr())
        for dst, local in ((gbeta, sums[:Cc]), (ggamma, sums[Cc:])):  # parameter gradients: this replica's samples only
            if dst is not None:
                dst.add_(local) if accumulate else dst.copy_(local)
        all_reduce_(sums)
        call("lhg_bn_backward_apply", pg, ldg, px, ldx, py, ldy, N * H * W, Cc, ptr(stats), ptr(gamma), ptr(sums), 1.0 / float(N * H * W * world),
             act, float(slope), ptr(gx), Cc, ptr(gres), Cc, ptr(gx_amax), ptr(beta), stream_ptr())
        tag_absmax(gx, gx_amax)
        return gx, gres
    # gx is the gy operand of the preceding conv's two backward GEMMs, gres of the shortcut conv's: the apply launch finishes max|.| and the
    # per-channel maxima of both on its way out (one call, two launches: sums finished inside the first)
    cm = chanmax_dest(Cc, N * H * W, gy.device)
    cm_res = chanmax_dest(Cc, N * H * W, gy.device) if (want_res and cm is not None) else None
    gres_amax = fused_absmax_slot(gy.device) if (want_res and cm is not None) else None
    if _FUSED_BN:
        ws, tickets = fused_scratch(gy.device)
        call("lhg_bn_backward_fused", pg, ldg, px, ldx, py, ldy, N * H * W, Cc, ptr(stats), ptr(gamma), ptr(beta), act, float(slope),
             ptr(gx), Cc, ptr(gres), Cc, ptr(ggamma), ptr(gbeta), int(accumulate), ptr(gx_amax), ptr(gres_amax), cm.ptr if cm else None,
             cm_res.ptr if cm_res else None, cm.finish if cm else 0, ws, tickets, stream_ptr())
    elif cm is not None:
        call("lhg_bn_backward_chanmax", pg, ldg, px, ldx, py, ldy, N * H * W, Cc, ptr(stats), ptr(gamma), act, float(slope),
             ptr(gx), Cc, ptr(gres), Cc, ptr(ggamma), ptr(gbeta), int(accumulate), ptr(_bn_ws(Cc, gy.device)), ptr(gx_amax), ptr(beta),
             cm.ptr, cm_res.ptr if cm_res else None, ptr(gres_amax), stream_ptr())
    else:
        call("lhg_bn_backward", pg, ldg, px, ldx, py, ldy, N * H * W, Cc, ptr(stats), ptr(gamma), act, float(slope),
             ptr(gx), Cc, ptr(gres), Cc, ptr(ggamma), ptr(gbeta), int(accumulate), ptr(_bn_ws(Cc, gy.device)), ptr(gx_amax), ptr(beta),
             stream_ptr())
    tag_absmax(gx, gx_amax)
    if gres_amax is not None:
        tag_absmax(gres, gres_amax)
    if cm is not None:
        cm.commit(gx)
    if cm_res is not None:
        cm_res.commit(gres)
    return gx, gres


class BatchNormGradFn(TrackedFunction):
    """First backward of BatchNormTrainFn as a differentiable op (needed by the gradient penalty).

    Its own backward (the double backward) contributes d/d gamma: in a plain backward pass into a registered slot that contribution is
    ADDED TO THE SLOT here and counted (note_use / note_contribution) like every other contribution of the hot path — a gamma that
    received some contributions through the slot and this one through autograd's AccumulateGrad could have its bucket's all-reduce
    launched (distributed.GradSynchronizer) between the two."""

    @staticmethod
    def forward(ctx, gy, x, y, gamma, stats, act, slope, want_res, world=1):
        Cc = gy.shape[-1]
        ctx.gamma = gamma if ctx.needs_input_grad[3] else None
        note_use(ctx.gamma)
        ggamma = torch.empty((Cc,), dtype=torch.float32, device=gy.device)
        gbeta = torch.empty((Cc,), dtype=torch.float32, device=gy.device)
        gx, gres = bn_backward_raw(gy, x, y, gamma, stats, act, slope, want_res, ggamma, gbeta, False, None, world)
        ctx.save_for_backward(gy, x, y, gamma, stats)
        ctx.act, ctx.slope, ctx.want_res, ctx.world = act, slope, want_res, world
        ctx.set_materialize_grads(False)
        if gres is None:
            gres = torch.empty(0, device=gy.device)
            ctx.mark_non_differentiable(gres)
        return gx, gres, ggamma, gbeta

    @staticmethod
    def backward(ctx, ggx, ggres, gggamma, ggbeta):
        gy, x, y, gamma, stats = ctx.saved_tensors
        if ggres is not None or gggamma is not None or ggbeta is not None:
            raise NotImplementedError("double backward through the residual / gamma / beta gradients of batch norm "
                                      "is not part of the hot path (the gradient penalty differentiates d/dx only)")
        nothing = (None,) * 9
        if ggx is None:
            if param_grads_wanted() and _small_grad_slot(ctx.gamma) is not None:
                note_contribution(ctx.gamma)  # a zero contribution still answers the recorded use
            return nothing
        N, H, W, Cc = x.shape
        pixels = N * H * W
        ggx_d, gy_d, x_d, y_d = _dense(ggx), _dense(gy), _dense(x), _dense(y)
        ggy = new_nhwc(N, H, W, Cc, x.device)
        gx2 = new_nhwc(N, H, W, Cc, x.device)
        ggamma2 = torch.empty((Cc,), dtype=torch.float32, device=x.device)
        if ctx.world > 1:  # global-batch statistics: the five sums are all-reduced between the kernel's halves
            ws = _bn_ws(Cc, x.device, 5 * 4096 + 8)
            sums = torch.empty((5 * Cc,), dtype=torch.float32, device=x.device)
            call("lhg_bn_backward_backward_sums", ptr(ggx_d), ptr(gy_d), ptr(x_d), ptr(y_d), pixels, Cc, ptr(stats), ptr(gamma), ctx.act,
                 float(ctx.slope), ptr(sums), ptr(ws), stream_ptr())
            all_reduce_(sums)
            call("lhg_bn_backward_backward_apply", ptr(ggx_d), ptr(gy_d), ptr(x_d), ptr(y_d), pixels, Cc, ptr(stats), ptr(gamma), ptr(sums),
                 1.0 / float(pixels * ctx.world), ctx.act, float(ctx.slope), ptr(ggy), ptr(gx2), ptr(ggamma2), stream_ptr())
            # every term of d/d gamma is linear in ggx, whose scale is this replica's loss (W times the global-batch loss's): the
            # formula over the GLOBAL sums is W times the global-batch gradient on every rank; its share here is 1/W of it
            ggamma2.div_(ctx.world)
        else:
            if _FUSED_BN:
                fws, tickets = fused_scratch(x.device)
                call("lhg_bn_backward_backward_fused", ptr(ggx_d), ptr(gy_d), ptr(x_d), ptr(y_d), pixels, Cc, ptr(stats), ptr(gamma),
                     ctx.act, float(ctx.slope), ptr(ggy), ptr(gx2), ptr(ggamma2), fws, tickets, stream_ptr())
            else:
                call("lhg_bn_backward_backward", ptr(ggx_d), ptr(gy_d), ptr(x_d), ptr(y_d), pixels, Cc, ptr(stats), ptr(gamma),
                     ctx.act, float(ctx.slope), ptr(ggy), ptr(gx2), ptr(ggamma2), ptr(_bn_ws(Cc, x.device, 5 * 4096 + 8)), stream_ptr())
        if not param_grads_wanted():
            return (ggy, gx2) + nothing[2:]
        slot = _small_grad_slot(ctx.gamma)
        if slot is not None:  # plain backward into the flat buffer: same delivery as BatchNormTrainFn.backward's d gamma
            slot.add_(ggamma2)
            note_contribution(ctx.gamma)
            return (ggy, gx2) + nothing[2:]
        return (ggy, gx2, None, ggamma2) + nothing[4:]


def batch_norm_eval_affine(gamma, beta, running_mean, running_var):
    """Per-channel (scale, shift) of an eval-mode BN, folded into the conv epilogue."""
    scale = gamma / torch.sqrt(running_var + BN_EPS)
    return scale, beta - running_mean * scale


# --------------------------------------------------------------------------- pooling
class MaxPool2x2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        px, N, H, W, Cc, ldx = nhwc(x)
        y = new_nhwc(N, H // 2, W // 2, Cc, x.device)
        call("lhg_maxpool2x2_forward", px, N, H, W, Cc, ldx, ptr(y), Cc, stream_ptr())
        ctx.save_for_backward(x)
        return _inherit_chanmax(_inherit_absmax(y, x), x)  # max|pool(x)| <= max|x|, per channel too: any upper bound is a valid scale

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        px, N, H, W, Cc, ldx = nhwc(x)
        pg, _, _, _, _, ldg = nhwc(gy)
        gx = new_nhwc(N, H, W, Cc, x.device)
        call("lhg_maxpool2x2_backward", px, ldx, pg, ldg, N, H, W, Cc, ptr(gx), Cc, stream_ptr())
        return _inherit_absmax(gx, gy)  # gx holds gy's values and zeros


class MaxPool2x2SkipFn(Function):
    """(max_pool2d(x, 2), x): MaxPool2x2Fn for an input with a second consumer — an encoder block's output feeds the next block through the
    pool AND the decoder through the skip concatenation (ref: neural_network_components.py:299-313).  The second output is x itself;
    the gradient that arrives through it (a channel slice of the concatenation's gradient: strided) is added to the pool's gradient by
    the pool's own backward kernel (lhg_maxpool2x2_backward_add) instead of by autograd's accumulation — a pass of its own over the two
    full-size gradients, 2 - 10 x slower than a dense add because one of them is a strided slice (round 4: 0.4 ms per step)."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)  # an unused output must not cost a zero-filled gradient
        px, N, H, W, Cc, ldx = nhwc(x)
        y = new_nhwc(N, H // 2, W // 2, Cc, x.device)
        call("lhg_maxpool2x2_forward", px, N, H, W, Cc, ldx, ptr(y), Cc, stream_ptr())
        ctx.save_for_backward(x)
        return _inherit_chanmax(_inherit_absmax(y, x), x), x

    @staticmethod
    def backward(ctx, gy, g_skip):
        (x,) = ctx.saved_tensors
        if gy is None:
            return g_skip
        px, N, H, W, Cc, ldx = nhwc(x)
        pg, _, _, _, _, ldg = nhwc(gy)
        gx = new_nhwc(N, H, W, Cc, x.device)
        if g_skip is None:
            call("lhg_maxpool2x2_backward", px, ldx, pg, ldg, N, H, W, Cc, ptr(gx), Cc, stream_ptr())
            return _inherit_absmax(gx, gy)
        g_skip = _as_nhwc_view(g_skip)
        ps, _, _, _, _, lds = nhwc(g_skip)
        call("lhg_maxpool2x2_backward_add", px, ldx, pg, ldg, N, H, W, Cc, ps, lds, ptr(gx), Cc, stream_ptr())
        return gx


def maxpool2x2_with_skip(x):
    """(pool(x), x') with x' an alias of x for its other consumer — see MaxPool2x2SkipFn (LHG_FUSE_SKIP_GRAD=0: plain MaxPool2x2Fn)."""
    if not FUSE_SKIP_GRAD or not (torch.is_grad_enabled() and x.requires_grad):
        return MaxPool2x2Fn.apply(x), x
    y, xs = MaxPool2x2SkipFn.apply(x)
    return y, _inherit_chanmax(_inherit_absmax(xs, x), x, alias=True)


# --------------------------------------------------------------------------- sigmoid head (planar output)
class SigmoidHeadFn(TrackedFunction):
    """y (N, Co, H, W) = sigmoid(conv1x1(x) + bias), written planar by the GEMM epilogue.
    ref: neural_network_components.py:292-295."""

    @staticmethod
    def forward(ctx, x, w, bias):
        y = conv2d_forward_raw(x, w, bias, 1, act=ACT_SIGMOID, planar=True)
        ctx.save_for_backward(x, w, y)
        note_use(w, ctx.needs_input_grad[1])
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        g_pre = gy.float() * y * (1 - y)  # (N, Co, H, W) fp32: 6 planes, negligible
        g_nhwc = ToNHWC.apply(g_pre, 8 if thin_mode(w.shape[1], w.shape[0], 1, 1) else 32)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = Conv2dInputGradFn.apply(g_nhwc, w, 1, x.shape[1], x.shape[2], x.shape[3])
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(w, (x, g_nhwc), lambda slot: conv2d_weight_grad(x, g_nhwc, w.shape, 1, slot))
        if ctx.needs_input_grad[2]:
            gb = g_pre.sum(dim=(0, 2, 3))
        return gx, gw, gb


# --------------------------------------------------------------------------- optimiser
def adam_step_(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, consts=None):
    """torch.optim.Adam on the flat buffers; ``grad_scale`` multiplies g on the fly (1 / world after a summing all-reduce); ``consts``:
    a 4-float DEVICE tensor {1 - beta1^t, sqrt(1 - beta2^t), grad_scale, lr} the kernel reads instead of step / grad_scale / lr (a
    captured launch then follows the step count: graph.GraphedTrainStep)."""
    call("lhg_adam_step_scaled", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps), int(step),
         float(grad_scale), ptr(consts), stream_ptr())
    bump_version(p)


_ = ctypes  # keep the import explicit: AsmFilter users import ctypes through native
