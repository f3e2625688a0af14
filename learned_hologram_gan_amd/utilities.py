"""Host-side helpers of the hot path: frequency masks, checkerboards, normalisers, seeding and
device selection.  Mirrors the names of ref: learnedMethodForHologram/utilities.py that the
propagators, models and entry points use (plotting / dataset-prep helpers are out of scope,
SURVEY §2.1).

The constant builders run on the HOST in fp32 with the reference's operation order on purpose:
the transfer-function phases are ~1e4 rad, so an "exact" recomputation changes outputs by more
than the 1e-4 parity budget.
"""

from __future__ import annotations

import random

import numpy as np
import torch


# --------------------------------------------------------------------------- devices / seeding
def num_gpus() -> int:
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def try_gpu(i: int = 0) -> torch.device:
    """cuda:i when present, else CPU with a notice (ref: utilities.py:410-415).  Constants may
    live on the CPU; the compute ops themselves refuse CPU tensors."""
    if num_gpus() > i:
        # i == 0 means "this process's GPU": under torch.distributed every rank has called set_device(LOCAL_RANK)
        return torch.device("cuda", torch.cuda.current_device() if i == 0 else i)
    print(f"gpu with index '{i}' is not available")
    return torch.device("cpu")


def try_all_gpus():
    return [torch.device(f"cuda:{i}") for i in range(num_gpus())]


def set_seed(seed: int) -> None:
    """ref: utilities.py:385-400."""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


# --------------------------------------------------------------------------- masks
def _radial_frequency(rows: int, cols: int) -> torch.Tensor:
    """sqrt(u^2 + v^2) * min(rows, cols) on the fftfreq grid (fp32, host)."""
    u = torch.fft.fftfreq(rows).unsqueeze(-1)
    v = torch.fft.fftfreq(cols).unsqueeze(0)
    return torch.sqrt(u**2 + v**2) * min(rows, cols)


def generate_circular_frequency_mask(sample_row_num=192, sample_col_num=192, radius=60, decay_rate=None):
    """0/1 low-pass disc in FFT order (optionally with an exponential skirt).
    ref: utilities.py:206-243; raises ValueError when radius > shorter_edge / 2."""
    shorter = min(sample_row_num, sample_col_num)
    if radius > shorter / 2:
        raise ValueError(f"The radius {radius} is larger than the half of the sample size {shorter/2}")
    dist = _radial_frequency(sample_row_num, sample_col_num)
    outside = dist > radius
    mask = torch.ones_like(dist)
    mask[outside] = torch.exp(-decay_rate * (dist[outside] - radius)) if decay_rate is not None else 0.0
    return mask


def generate_circular_frequency_mask_modified(sample_row_num=192, sample_col_num=192, filter_radius_coefficient=0.5):
    """Same disc without the range check. ref: utilities.py:246-274."""
    dist = _radial_frequency(sample_row_num, sample_col_num)
    mask = torch.ones_like(dist)
    mask[dist > min(sample_row_num, sample_col_num) * filter_radius_coefficient] = 0.0
    return mask


def prepare_circular_frequency_mask_grid(samplingRowNum, samplingColNum):
    """ref: utilities.py:277-297."""
    return _radial_frequency(samplingRowNum, samplingColNum)


def generate_checkerboard_mask(height=192, width=192, cell_size=4, reserve=False):
    """((x//cell + y//cell) % 2) as fp32; ``reserve`` flips it. ref: utilities.py:354-382."""
    x = torch.arange(width).view(1, -1) // cell_size
    y = torch.arange(height).view(-1, 1) // cell_size
    board = ((x + y) % 2).to(torch.float32)
    return 1 - board if reserve else board


# --------------------------------------------------------------------------- normalisers
def amplitude_normalizor(amp):
    """amp / (1.01 * max over the last two dims). ref: utilities.py:53-66."""
    peak = torch.amax(amp, dim=(-2, -1), keepdim=True)
    return amp / (peak * 1.01)


def tensor_normalizor_2D(tensor_to_normalize):
    """(x - min) / (max - min) per plane. ref: utilities.py:69-84."""
    hi = torch.amax(tensor_to_normalize, dim=(-2, -1), keepdim=True)
    lo = torch.amin(tensor_to_normalize, dim=(-2, -1), keepdim=True)
    return (tensor_to_normalize - lo) / (hi - lo)


def complex_plain(amplitude_tensor, phase_tensor):
    """ref: utilities.py:15-27."""
    return amplitude_tensor * torch.exp(1j * phase_tensor)


def save_planes_as_png(planes01: torch.Tensor, save_dir: str, rgb_img: bool = True, titles=None) -> None:
    """Write (K,3,H,W) tensors in [0,1] as <title>.png, titles defaulting to 0..K-1 (8-bit, truncated like the
    reference's plotter output).  Minimal stand-in for multi_sample_plotter (utilities.py:160-203: one
    ``plt.imsave(save_dir/<title>.png)`` per sample) so that generatePOH.py --propagate and the trainer's
    visualisation dumps keep their documented artefacts."""
    import os

    from PIL import Image

    os.makedirs(save_dir, exist_ok=True)
    arr = (planes01.detach().clamp(0, 1) * 255.0).to("cpu").permute(0, 2, 3, 1).numpy().astype(np.uint8)
    for k in range(arr.shape[0]):
        name = k if titles is None else titles[k]
        Image.fromarray(arr[k] if rgb_img else arr[k, ..., 0]).save(os.path.join(save_dir, f"{name}.png"))
