"""Oracle: angular-spectrum optics.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Pure torch, CPU, fp32/complex64.

The reference keeps these as three classes with the constants as attributes
(ref: learnedMethodForHologram/angular_spectrum_method.py:5-552).  Here the same
arithmetic is written as free functions over an immutable ``Optics`` record so the
op order of each constant is visible in one place.
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import torch

DEFAULT_PITCH = 3.74e-6
DEFAULT_WAVELENGTHS = (638e-9, 520e-9, 450e-9)


# --------------------------------------------------------------------------- A1
def padded_shape(rows0: int, cols0: int, pad: int):
    """ref: angular_spectrum_method.py:45-49 — column pad is scaled by the aspect ratio."""
    pad_c = int(pad * (cols0 / rows0))
    return rows0 + 2 * pad, cols0 + 2 * pad_c, pad, pad_c


def w_grid(rows: int, cols: int, pitch: float, wave_length: torch.Tensor) -> torch.Tensor:
    """sqrt(clamp(1/lambda^2 - (fx^2 + fy^2), 0)), shape (3, rows, cols), fp32.

    ref: angular_spectrum_method.py:56-57 (fftfreq grids), :155-171 (w grid).
    The fp32 op order matters: 1/lambda^2 ~ 2.4e12 has an ulp of 2.6e5, so the
    subtraction must be done on fp32 operands exactly as the reference does.
    """
    fx = torch.fft.fftfreq(rows, pitch)
    fy = torch.fft.fftfreq(cols, pitch)
    rho2 = fx.unsqueeze(1) ** 2 + fy.unsqueeze(0) ** 2
    inv_l2 = 1 / wave_length**2
    return torch.sqrt(torch.clamp(inv_l2.view(-1, 1, 1) - rho2.unsqueeze(0), min=0))


def lowpass_mask(rows: int, cols: int, coefficient: float) -> torch.Tensor:
    """Circular 0/1 low-pass mask in fftfreq order, fp32 (rows, cols).

    ref: utilities.py:206-243 (generate_circular_frequency_mask) called from
    angular_spectrum_method.py:141-153 with radius = min(rows, cols) * coefficient.
    Raises ValueError when the radius exceeds half the shorter edge (utilities.py:226-229).
    """
    shorter = min(rows, cols)
    radius = shorter * coefficient
    if radius > shorter / 2:
        raise ValueError(
            f"The radius {radius} is larger than the half of the sample size {shorter/2}"
        )
    u = torch.fft.fftfreq(rows).unsqueeze(-1)
    v = torch.fft.fftfreq(cols).unsqueeze(0)
    dist = torch.sqrt(u**2 + v**2) * shorter
    mask = torch.ones_like(dist)
    mask[dist > radius] = 0.0
    return mask


def transfer_function(w: torch.Tensor, distances: torch.Tensor) -> torch.Tensor:
    """H = exp(-2j*pi*d*w): (D,3,R,C) complex64 for a (D,) distance vector.

    ref: angular_spectrum_method.py:206-211 (generic / multi-distance) and :464-466
    (fixed distance; there ``distance`` has shape (1,) and the result is (3,R,C) —
    take ``[0]`` of this function's output).  Both evaluate, in fp32,
    theta = fl(fl(-2pi * d) * w) and then exp(i*theta).
    """
    return torch.exp(-2j * torch.pi * distances.view(-1, 1, 1, 1) * w)


@dataclass(frozen=True)
class Optics:
    rows0: int
    cols0: int
    pad_r: int
    pad_c: int
    rows: int
    cols: int
    w: torch.Tensor  # (3,R,C) fp32
    mask: torch.Tensor  # (R,C) fp32 0/1


def make_optics(
    rows0: int,
    cols0: int,
    pad: int,
    coefficient: float,
    pitch: float = DEFAULT_PITCH,
    wave_length=None,
) -> Optics:
    """ref: angular_spectrum_method.py:30-66 (constructor of the base class)."""
    if wave_length is None:
        wave_length = torch.tensor(DEFAULT_WAVELENGTHS)
    rows, cols, pad_r, pad_c = padded_shape(rows0, cols0, pad)
    return Optics(
        rows0, cols0, pad_r, pad_c, rows, cols,
        w_grid(rows, cols, pitch, wave_length),
        lowpass_mask(rows, cols, coefficient),
    )


# --------------------------------------------------------------------------- A2
def pad_field(o: Optics, x: torch.Tensor) -> torch.Tensor:
    """Zero-pad the last two dims. ref: angular_spectrum_method.py:215-239."""
    if o.pad_r == 0:
        return x
    return torch.nn.functional.pad(x, (o.pad_c, o.pad_c, o.pad_r, o.pad_r))


def crop_field(o: Optics, x: torch.Tensor) -> torch.Tensor:
    """Centre crop, inverse of pad_field. ref: angular_spectrum_method.py:241-260.

    (The reference slices with ``pad:-pad`` on both axes, so pad_c == 0 with
    pad_r != 0 would give an empty tensor there; not reachable for cols0 >= rows0.)
    """
    if o.pad_r == 0:
        return x
    return x[..., o.pad_r : -o.pad_r, o.pad_c : -o.pad_c]


def polar(amp: torch.Tensor, phs: torch.Tensor) -> torch.Tensor:
    return amp * torch.exp(1j * phs)


# --------------------------------------------------------------------------- A5
def backpropagate_to_slm(o: Optics, H_fixed: torch.Tensor, amp_z, phs_z) -> torch.Tensor:
    """g0 = crop(ifft2(fft2(pad(amp*e^{i phs})) / H)) — no low-pass mask.

    ref: angular_spectrum_method.py:374-384 (propagate_AP2C_backward).
    """
    G = torch.fft.fft2(pad_field(o, polar(amp_z, phs_z)))
    return crop_field(o, torch.fft.ifft2(G / H_fixed))


# --------------------------------------------------------------------------- A8
def poh_to_filtered_spectrum(o: Optics, H_fixed: torch.Tensor, poh: torch.Tensor) -> torch.Tensor:
    """Gz = fft2(pad(e^{i POH})) * H * mask. ref: angular_spectrum_method.py:386-392."""
    return torch.fft.fft2(pad_field(o, torch.exp(1j * poh))) * H_fixed * o.mask


def poh_to_amp_phase(o: Optics, H_fixed: torch.Tensor, poh: torch.Tensor):
    """ref: angular_spectrum_method.py:414-424 (propagate_POH2AP_forward)."""
    g = crop_field(o, torch.fft.ifft2(poh_to_filtered_spectrum(o, H_fixed, poh)))
    return torch.abs(g), torch.angle(g)


# --------------------------------------------------------------------------- A9
def target_to_filtered_spectrum(o: Optics, amp: torch.Tensor, phs01: torch.Tensor) -> torch.Tensor:
    """fft2(pad(amp*e^{i 2pi phs})) * mask. ref: angular_spectrum_method.py:548-552."""
    return torch.fft.fft2(pad_field(o, polar(amp, 2 * torch.pi * phs01))) * o.mask


def spectrum_to_planes_indexed(o: Optics, H_stack: torch.Tensor, G: torch.Tensor, indices: torch.Tensor):
    """Training path: G holds (hat; target) stacked on dim 0, sample b of both halves
    is propagated to plane ``indices[b]``.

    ref: angular_spectrum_method.py:533-546.  The reference draws
    ``indices = torch.randperm(D)[:B]`` inside; the draw is lifted out here so the
    same indices can be handed to the implementation under test
    (``draw_plane_indices`` reproduces the draw).
    """
    H = H_stack[indices]
    Gz = G.view(2, -1, 3, o.rows, o.cols) * H * o.mask
    g = crop_field(o, torch.fft.ifft2(Gz.view(-1, 3, o.rows, o.cols)))
    return torch.abs(g), torch.angle(g)


def draw_plane_indices(num_planes: int, batch: int) -> torch.Tensor:
    """ref: angular_spectrum_method.py:536 — CPU RNG, global generator."""
    return torch.randperm(num_planes)[0:batch]


def spectrum_to_planes_all(o: Optics, H_stack: torch.Tensor, G: torch.Tensor):
    """Validation path: every sample to every plane; output order is sample-major
    (out[b*D + d]). ref: angular_spectrum_method.py:524-531."""
    Gz = G.unsqueeze(1) * H_stack * o.mask
    g = crop_field(o, torch.fft.ifft2(Gz.view(-1, 3, o.rows, o.cols)))
    return torch.abs(g), torch.angle(g)


def propagate_amplitudes(o: Optics, amp, phs, distances: torch.Tensor, H=None) -> torch.Tensor:
    """``__call__`` of the multi-distance class: |crop(ifft2(fft2(pad(a e^{i phs}))[:,None] * H(d) * mask))|,
    shape (B*D,3,h,w). ref: angular_spectrum_method.py:503-522 (used by generatePOH.py:51-70)."""
    G = torch.fft.fft2(pad_field(o, polar(amp, phs)))
    H = (transfer_function(o.w, distances) if H is None else H) * o.mask
    Gz = (G.unsqueeze(1) * H).view(-1, 3, o.rows, o.cols)
    return torch.abs(crop_field(o, torch.fft.ifft2(Gz)))


def normalize_planes_01(x: torch.Tensor) -> torch.Tensor:
    """Per-(b,c) min/max normalisation. ref: utilities.py:69-84 (tensor_normalizor_2D)."""
    hi = x.amax(dim=(-2, -1), keepdim=True)
    lo = x.amin(dim=(-2, -1), keepdim=True)
    return (x - lo) / (hi - lo)


def algorithmic_plane_transform_bytes(rows: int, cols: int) -> int:
    """one 2-D C2C of rows x cols complex64 = two 1-D passes, each reading
    and writing the plane once: 32*R*C bytes."""
    return 32 * rows * cols


def fft2_flops(rows: int, cols: int) -> float:
    return 5.0 * rows * cols * math.log2(rows * cols)
