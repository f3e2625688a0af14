"""Angular-spectrum propagators with the reference's class / method names
(ref: learnedMethodForHologram/angular_spectrum_method.py:5-552), running on the fused
pad -> FFT2 -> transfer-function -> IFFT2 -> crop HIP operator (asm_ops / csrc/asm_fft.hip).

Constants (frequency grids, w grid, low-pass mask, transfer functions) are built on the HOST in
fp32 in the reference's operation order and uploaded once; the mask is folded
into the uploaded transfer functions (it is 0/1, so H*mask is exact).

Every padded extent in [16, 8192] runs on that operator: products of the primes up to 13 directly (1024, 2304 x 4096, 832 =
192 + 2*320), anything else as a Bluestein convolution (2800 x 4976: a 4K frame with the CLI's pad 320).  Only larger non-smooth
extents take the ``torch.fft`` route on the same device (rocFFT): still GPU-only, never a CPU fallback.
"""

from __future__ import annotations

import os

import torch

from . import utilities
from .asm_ops import (Factor, FromSpectrumFn, Geometry, PropagateFn, Spec, ToSpectrumFn, from_spectrum_raw, propagate_raw, to_spectrum_raw)
from .native import F_DIV, F_MUL, IN_PHASE, IN_POLAR, OUT_ABS, OUT_ABS_ANGLE, OUT_COMPLEX

_DEFAULT_WL = (639e-9, 515e-9, 473e-9)


# multi-distance __call__: from this many distances per field on, the field's full spectrum is computed once and every distance is a filtered
# inverse pass of it (LHG_ASM_SHARE_SPECTRUM: 0 = never; the spectrum costs R x C instead of rows0 x C per plane in HBM)
_SHARE_SPECTRUM_FROM = int(os.environ.get("LHG_ASM_SHARE_SPECTRUM", "3")) or (1 << 30)


class bandLimitedAngularSpectrumMethod:
    """Generic propagator: distances are given per call.  Tensors are (batch|distances, 3, rows, cols)."""

    def __init__(self, sample_row_num=192, sample_col_num=192, pad_size=0, filter_radius_coefficient=0.5,
                 pixel_pitch=3.74e-6, wave_length=torch.tensor(_DEFAULT_WL), band_limit=False, cuda=False):
        self.originalRowNum, self.originalColNum = sample_row_num, sample_col_num
        self.pad_size_row = pad_size
        self.pad_size_col = int(pad_size * (sample_col_num / sample_row_num))
        self.samplingRowNum = sample_row_num + 2 * self.pad_size_row
        self.samplingColNum = sample_col_num + 2 * self.pad_size_col
        self.pixel_pitch, self.wave_length, self.band_limit = pixel_pitch, wave_length, band_limit
        self.device = utilities.try_gpu() if cuda else torch.device("cpu")

        self.freq_x = torch.fft.fftfreq(self.samplingRowNum, self.pixel_pitch)
        self.freq_y = torch.fft.fftfreq(self.samplingColNum, self.pixel_pitch)
        self._mask_host = self.generate_diffraction_limited_mask(filter_radius_coefficient).cpu()
        self._w_host = self._w_grid_host()
        self.diffraction_limited_mask = self._mask_host.to(self.device)
        self.w_grid = self._w_host.to(self.device)

        self._geom = Geometry(sample_row_num, sample_col_num, self.pad_size_row, self.pad_size_col)
        self._mask_slab = (self._mask_host + 0j).to(torch.complex64).unsqueeze(0).to(self.device)
        self._index_cache = {}
        self._H_call_cache = (None, None)

    # ------------------------------------------------------------------ constants (host, fp32)
    def _w_grid_host(self):
        rho2 = self.freq_x.unsqueeze(1) ** 2 + self.freq_y.unsqueeze(0) ** 2
        inv_l2 = (1 / self.wave_length.cpu() ** 2).view(-1, 1, 1)
        return torch.sqrt(torch.clamp(inv_l2 - rho2.unsqueeze(0), min=0))

    def generate_w_grid(self):
        return self._w_grid_host()

    def generate_diffraction_limited_mask(self, radius_coefficient):
        return utilities.generate_circular_frequency_mask(
            sample_row_num=self.samplingRowNum, sample_col_num=self.samplingColNum,
            radius=min(self.samplingRowNum, self.samplingColNum) * radius_coefficient).to(self.device)

    def _H_host(self, distances):
        d = torch.as_tensor(distances, dtype=torch.float32).detach().cpu().reshape(-1)
        return torch.exp(-2j * torch.pi * d.view(-1, 1, 1, 1) * self._w_host)

    def generate_transfer_function(self, distances):
        """H = exp(-2j*pi*d*w), (D,3,R,C) complex64, evaluated on the host in fp32."""
        return self._H_host(distances).to(self.device)

    def generate_band_limited_mask(self, distances):
        """Matsushima band limit — computed for API parity; the reference never applies it
        (angular_spectrum_method.py:65-66, 332-333)."""
        d = torch.as_tensor(distances, dtype=torch.float32).cpu().reshape(-1, 1)
        wl = self.wave_length.cpu().unsqueeze(0)
        lim = lambda n: 1 / (torch.sqrt((2 * (1 / (n * self.pixel_pitch)) * d) ** 2 + 1) * wl)  # noqa: E731
        mu = self.freq_x.abs().view(1, 1, -1, 1) < lim(self.samplingRowNum).unsqueeze(2).unsqueeze(3)
        mv = self.freq_y.abs().view(1, 1, 1, -1) < lim(self.samplingColNum).unsqueeze(2).unsqueeze(3)
        return mu & mv

    # ------------------------------------------------------------------ pad / crop (API parity)
    def padding(self, tensor):
        if self.pad_size_row == 0:
            return tensor
        return torch.nn.functional.pad(tensor, (self.pad_size_col, self.pad_size_col, self.pad_size_row, self.pad_size_row))

    def cropping(self, tensor):
        if self.pad_size_row == 0:
            return tensor
        return tensor[:, :, self.pad_size_row:-self.pad_size_row, self.pad_size_col:-self.pad_size_col]

    # ------------------------------------------------------------------ plumbing for the fused op
    def _colour_index(self, planes, offsets=None):
        """int32 (planes,) slab selector: plane p of colour p%3, optionally offset per (sample) by 3*distance index."""
        if offsets is None:
            key = ("c", planes)
            if key not in self._index_cache:
                self._index_cache[key] = (torch.arange(planes, dtype=torch.int32) % 3).to(self.device)
            return self._index_cache[key]
        if offsets.is_cuda:  # device-resident plane indices (a captured train step reads them from a static buffer): no host round trip
            return (offsets.to(torch.int32).reshape(-1, 1) * 3 + self._colour_index(3)).reshape(-1).contiguous()
        col = torch.arange(3, dtype=torch.int32)
        idx = (offsets.to(torch.int32).cpu().reshape(-1, 1) * 3 + col).reshape(-1)
        # a pageable host-to-device copy makes the host wait for everything queued on the stream: stage through pinned memory
        return idx.pin_memory().to(self.device, non_blocking=True) if torch.device(self.device).type == "cuda" else idx.to(self.device)

    def _run(self, a, b, in_mode, out_mode, factors, phase_scale=1.0):
        if not self._geom.supported():
            return self._run_rocfft(a, b, in_mode, out_mode, factors, phase_scale)
        return PropagateFn.apply(a, b, Spec(self._geom, in_mode, out_mode, phase_scale, tuple(factors)))

    def _field(self, a, b, in_mode, phase_scale):
        if in_mode == IN_PHASE:
            return torch.exp(1j * (a * phase_scale))
        return a * torch.exp(1j * (b * phase_scale))

    def _apply_factors(self, G, factors):
        lead = G.shape[:-2]
        for f in factors:
            sl = f.slabs if f.index is None else f.slabs[f.index.long()]
            sl = sl.reshape(lead + sl.shape[-2:]) if f.index is not None else sl[0]
            G = G / sl if f.op == F_DIV else G * sl
        return G

    def _run_rocfft(self, a, b, in_mode, out_mode, factors, phase_scale):
        """Same maths through torch.fft on the GPU for extents the LDS FFT does not cover."""
        if not a.is_cuda:
            raise RuntimeError("angular-spectrum ops run on the GPU only (no CPU fallback)")
        g = self.cropping_any(torch.fft.ifft2(self._apply_factors(torch.fft.fft2(self.padding_any(self._field(a, b, in_mode, phase_scale))), factors)))
        if out_mode == OUT_COMPLEX:
            return None, None, g
        return torch.abs(g), (torch.angle(g) if out_mode == OUT_ABS_ANGLE else None), None

    def padding_any(self, t):
        return t if self.pad_size_row == 0 else torch.nn.functional.pad(t, (self.pad_size_col, self.pad_size_col, self.pad_size_row, self.pad_size_row))

    def cropping_any(self, t):
        return t if self.pad_size_row == 0 else t[..., self.pad_size_row:-self.pad_size_row, self.pad_size_col:-self.pad_size_col]

    def _masked_H_for_call(self, distances):
        key = tuple(float(x) for x in torch.as_tensor(distances).reshape(-1).tolist())
        if self._H_call_cache[0] != key:
            H = (self._H_host(distances) * self._mask_host).to(torch.complex64)
            self._H_call_cache = (key, H.reshape(-1, self.samplingRowNum, self.samplingColNum).to(self.device))
        return self._H_call_cache[1]

    def set_mask(self, mask):
        """Install a recorded low-pass mask (its boundary pixels also depend on the host's sqrt)."""
        self._mask_host = mask.detach().cpu().to(torch.float32)
        self.diffraction_limited_mask = self._mask_host.to(self.device)
        self._mask_slab = (self._mask_host + 0j).to(torch.complex64).unsqueeze(0).to(self.device)
        if hasattr(self, "H"):
            self.set_transfer_function(self.H)

    def set_call_transfer_function(self, distances, H):
        """Use a given H (D,3,R,C) for ``__call__(…, distances)`` instead of evaluating it on this host.
        ATen's CPU sqrt (MKL VML) differs in the last bit between CPU models and one ulp of the w grid is
        8e-4 rad of phase, so transfer functions are only reproducible on the host that built them; parity
        tests against recorded reference outputs inject the recorded H."""
        key = tuple(float(x) for x in torch.as_tensor(distances).reshape(-1).tolist())
        Hm = (H.detach().cpu().to(torch.complex64) * self._mask_host).to(torch.complex64)
        self._H_call_cache = (key, Hm.reshape(-1, self.samplingRowNum, self.samplingColNum).to(self.device))

    # ------------------------------------------------------------------ reference API
    def __call__(self, amplitute_tensor, phase_tensor, distances):
        """|crop(ifft2(fft2(pad(a e^{i phi})) * H(d) * mask))|; dim 0 is batch OR distances
        (ref: angular_spectrum_method.py:68-94)."""
        D = int(torch.as_tensor(distances).numel())
        B = amplitute_tensor.shape[0]
        n = max(B, D)
        if B not in (1, n) or D not in (1, n):
            raise RuntimeError(f"batch {B} and {D} distances do not broadcast (the reference has the same limit)")
        a = amplitute_tensor.expand(n, *amplitute_tensor.shape[1:]).contiguous()
        p = phase_tensor.expand(n, *phase_tensor.shape[1:]).contiguous()
        idx = self._colour_index(3 * n, torch.arange(n) if D > 1 else torch.zeros(n, dtype=torch.int64))
        amp, _, _ = self._run(a, p, IN_POLAR, OUT_ABS, [Factor(self._masked_H_for_call(distances), F_MUL, idx)])
        return amp

    def propagate_AP2AP(self, amp_phs_tensor_0, distances):
        """(B,6,R,C) interleaved [amp_r, phs_r, amp_g, ...] -> cat(|g|, angle g) without the mask
        (ref: angular_spectrum_method.py:96-129; only valid for pad 0 there, as here)."""
        v = amp_phs_tensor_0.view(-1, 3, 2, self.samplingRowNum, self.samplingColNum)
        H = self._H_host(distances).to(torch.complex64).reshape(-1, self.samplingRowNum, self.samplingColNum).to(self.device)
        n = v.shape[0]
        D = H.shape[0] // 3
        idx = self._colour_index(3 * n, torch.arange(n) if D > 1 else torch.zeros(n, dtype=torch.int64))
        amp, phs, _ = self._run(v[:, :, 0].contiguous(), v[:, :, 1].contiguous(), IN_POLAR, OUT_ABS_ANGLE, [Factor(H, F_MUL, idx)])
        return torch.cat((amp, phs), dim=1)

    def propagate_P2I(self, phase_tensor, distances):
        """Intensity of a phase-only field (ref: angular_spectrum_method.py:131-139)."""
        return self(torch.ones_like(phase_tensor), phase_tensor, distances) ** 2


class bandLimitedAngularSpectrumMethod_for_single_fixed_distance(bandLimitedAngularSpectrumMethod):
    """One fixed distance baked into ``H`` (3,R,C); used inside the generator (z = 1 mm)."""

    def __init__(self, sample_row_num=192, sample_col_num=192, pad_size=0, filter_radius_coefficient=0.5,
                 pixel_pitch=3.74e-6, wave_length=torch.tensor(_DEFAULT_WL), band_limit=False, cuda=False,
                 distance=torch.tensor([1e-3])):
        super().__init__(sample_row_num, sample_col_num, pad_size, filter_radius_coefficient, pixel_pitch, wave_length, band_limit, cuda)
        self.distance = distance
        self.circular_frequency_mask_differentiable_grid = utilities.prepare_circular_frequency_mask_grid(
            self.samplingRowNum, self.samplingColNum).to(self.device)
        self.band_limited_mask = self.generate_band_limited_mask().to(self.device)
        H_host = self._H_host(self.distance)[0].to(torch.complex64)
        self.set_transfer_function(H_host)

    def set_transfer_function(self, H):
        """Install H (3,R,C) (see set_call_transfer_function for why this hook exists)."""
        H_host = H.detach().cpu().to(torch.complex64)
        self.H = H_host.to(self.device)
        self._H_masked = (H_host * self._mask_host).to(torch.complex64).to(self.device)

    def generate_transfer_function(self, distances=None):
        return self._H_host(self.distance if distances is None else distances)[0].to(self.device)

    def generate_band_limited_mask(self, distances=None):
        return super().generate_band_limited_mask(self.distance if distances is None else distances)

    def generate_circular_frequency_mask_differentiable(self, filter_radius_coefficient):
        radius = min(self.samplingRowNum, self.samplingColNum) * filter_radius_coefficient
        return torch.sigmoid(1.0 * (radius - self.circular_frequency_mask_differentiable_grid))

    def _H_factor(self, planes, masked, op=F_MUL):
        return Factor(self._H_masked if masked else self.H, op, self._colour_index(planes))

    def __call__(self, amplitute_tensor, phase_tensor):
        """ref: angular_spectrum_method.py:323-336."""
        planes = amplitute_tensor.shape[0] * 3
        return self._run(amplitute_tensor, phase_tensor, IN_POLAR, OUT_ABS, [self._H_factor(planes, True)])[0]

    def propagate_AP2AP(self, amp_phs_tensor_0):
        """Backward propagation, (B,6,..) interleaved in, cat(|g|, angle g) out (ref: :338-368)."""
        v = amp_phs_tensor_0.view(-1, 3, 2, self.samplingRowNum, self.samplingColNum)
        amp, phs, _ = self._run(v[:, :, 0].contiguous(), v[:, :, 1].contiguous(), IN_POLAR, OUT_ABS_ANGLE,
                                [self._H_factor(v.shape[0] * 3, False, F_DIV)])
        return torch.cat((amp, phs), dim=1)

    def propagate_AP2C_backward(self, amp_z, phs_z):
        """g0 = crop(ifft2(fft2(pad(amp e^{i phs})) / H)) — complex (B,3,h,w). ref: :374-384 (A5)."""
        return self._run(amp_z, phs_z, IN_POLAR, OUT_COMPLEX, [self._H_factor(amp_z.shape[0] * 3, False, F_DIV)])[2]

    def propagate_POH2Freq_forward(self, POH):
        """fft2(pad(e^{i POH})) * H * mask — full (B,3,R,C) spectrum. ref: :386-392 (A8)."""
        f = [self._H_factor(POH.shape[0] * 3, True)]
        if not self._geom.supported():
            return self._apply_factors(torch.fft.fft2(self.padding_any(torch.exp(1j * POH))), f)
        return ToSpectrumFn.apply(POH, None, Spec(self._geom, IN_PHASE, OUT_COMPLEX, 1.0, tuple(f)))

    def propagate_POH2AP_forward(self, phs_0):
        """(|g|, angle g) at the fixed distance. ref: :414-424."""
        amp, phs, _ = self._run(phs_0, None, IN_PHASE, OUT_ABS_ANGLE, [self._H_factor(phs_0.shape[0] * 3, True)])
        return amp, phs

    def propagate_POH2AP_forward_with_spectrum_loss(self, phs_0, filter_radius_coefficient=torch.tensor(0.5)):
        """Pre-training helper with a differentiable sigmoid mask (ref: :394-412).
        Built from the spectrum ops: G0 -> G0*H*sigmoid_mask -> crop(ifft2)."""
        planes = phs_0.shape[0] * 3
        if not self._geom.supported():
            G0 = torch.fft.fft2(self.padding_any(torch.exp(1j * phs_0)))
        else:
            G0 = ToSpectrumFn.apply(phs_0, None, Spec(self._geom, IN_PHASE, OUT_COMPLEX, 1.0, ()))
        Gz = G0 * self.H * self.generate_circular_frequency_mask_differentiable(filter_radius_coefficient)
        loss = torch.mean(torch.abs(G0) - torch.abs(Gz))
        if not self._geom.supported():
            g = self.cropping_any(torch.fft.ifft2(Gz))
            return torch.abs(g), torch.angle(g), loss
        amp, phs, _ = FromSpectrumFn.apply(Gz.contiguous(), Spec(self._geom, IN_PHASE, OUT_ABS_ANGLE, 1.0, ()))
        del planes
        return amp, phs, loss


class bandLimitedAngularSpectrumMethod_for_multiple_distances(bandLimitedAngularSpectrumMethod):
    """Batch x distances at once; ``H`` (D,3,R,C) for a fixed stack of distances."""

    def __init__(self, sample_row_num=192, sample_col_num=192, distances=None, pad_size=160, filter_radius_coefficient=0.5,
                 pixel_pitch=3.74e-6, wave_length=torch.tensor(_DEFAULT_WL), band_limit=False, cuda=True):
        super().__init__(sample_row_num, sample_col_num, pad_size, filter_radius_coefficient, pixel_pitch, wave_length, band_limit, cuda)
        self.distances = distances.to(self.device)
        H_host = self._H_host(distances).to(torch.complex64)
        self.set_transfer_function(H_host)
        self.last_indices = None

    def set_transfer_function(self, H):
        """Install the stack H (D,3,R,C) (see set_call_transfer_function for why this hook exists)."""
        H_host = H.detach().cpu().to(torch.complex64)
        self.H = H_host.to(self.device)
        self._H_masked = (H_host * self._mask_host).to(torch.complex64).reshape(-1, self.samplingRowNum, self.samplingColNum).to(self.device)

    def __call__(self, amplitute_tensor, phase_tensor, distances):
        """(B,3,h,w) x D distances -> |g| (B*D,3,h,w), sample-major. ref: :503-522."""
        D = int(torch.as_tensor(distances).numel())
        B = amplitute_tensor.shape[0]
        if (self._geom.supported() and amplitute_tensor.is_cuda and amplitute_tensor.shape[1] == 3
                and not (torch.is_grad_enabled() and (amplitute_tensor.requires_grad or phase_tensor.requires_grad))):
            # every field goes to D planes: the first pass (polar -> complex, row transforms) once per FIELD, the D filtered column / inverse
            # passes read it by index — no (B D, 3, h, w) copies of the inputs, B 3 instead of B D 3 row-transform planes
            key = ("src", B, D)
            if key not in self._index_cache:
                b_, c_ = torch.arange(B).view(B, 1, 1), torch.arange(3).view(1, 1, 3)
                self._index_cache[key] = (b_ * 3 + c_).expand(B, D, 3).reshape(-1).to(torch.int32).to(self.device)
            idx = self._colour_index(3 * B * D, torch.arange(D).repeat(B))
            f = (Factor(self._masked_H_for_call(distances), F_MUL, idx),)
            if D >= _SHARE_SPECTRUM_FROM:
                # ... and the forward column transforms too: the fields' spectra once (B 3 planes), then D filtered inverse passes each
                S = to_spectrum_raw(amplitute_tensor, phase_tensor, Spec(self._geom, IN_POLAR, OUT_COMPLEX, 1.0, ()))
                return from_spectrum_raw(S, Spec(self._geom, IN_PHASE, OUT_ABS, 1.0, f), plane_src=self._index_cache[key], out_lead=(B * D, 3))[0]
            return propagate_raw(amplitute_tensor, phase_tensor, Spec(self._geom, IN_POLAR, OUT_ABS, 1.0, f), plane_src=self._index_cache[key],
                                 out_lead=(B * D, 3))[0]
        a = amplitute_tensor.unsqueeze(1).expand(B, D, *amplitute_tensor.shape[1:]).reshape(B * D, *amplitute_tensor.shape[1:])
        p = phase_tensor.unsqueeze(1).expand(B, D, *phase_tensor.shape[1:]).reshape(B * D, *phase_tensor.shape[1:])
        idx = self._colour_index(3 * B * D, torch.arange(D).repeat(B))
        return self._run(a.contiguous(), p.contiguous(), IN_POLAR, OUT_ABS, [Factor(self._masked_H_for_call(distances), F_MUL, idx)])[0]

    def _from_spectrum(self, G, plane_offsets):
        idx = self._colour_index(G.shape[0] * 3, plane_offsets)
        f = [Factor(self._H_masked, F_MUL, idx)]
        if not self._geom.supported():
            g = self.cropping_any(torch.fft.ifft2(self._apply_factors(G, f)))
            return torch.abs(g), torch.angle(g)
        amp, phs, _ = FromSpectrumFn.apply(G.contiguous(), Spec(self._geom, IN_PHASE, OUT_ABS_ANGLE, 1.0, tuple(f)))
        return amp, phs

    def propagate_multiple_samples_with_all_fixed_multiple_distances_freq2amp(self, G_0):
        """Every sample to every plane, out[b*D + d]. ref: :524-531."""
        B, D = G_0.shape[0], self.H.shape[0]
        G = G_0.unsqueeze(1).expand(B, D, *G_0.shape[1:]).reshape(B * D, *G_0.shape[1:])
        return self._from_spectrum(G, torch.arange(D).repeat(B))

    def draw_indices(self, half_batch):
        """The reference's draw: torch.randperm(D)[:B] on the CPU generator (:536)."""
        self.last_indices = torch.randperm(self.H.size(0))[0:half_batch]
        return self.last_indices

    def propagate_multiple_samples_with_random_fixed_multiple_distances_freq2amp(self, G_0, indices=None):
        """G_0 = cat(hat, target) on dim 0; sample b of both halves goes to plane indices[b]. ref: :533-546."""
        half = G_0.size(0) // 2
        if indices is None:
            indices = self.draw_indices(half)
        return self._from_spectrum(G_0, torch.cat((indices, indices)))

    def filter_AP2filteredFreq(self, amp, phs):
        """fft2(pad(amp e^{i 2 pi phs})) * mask. ref: :548-552."""
        f = [Factor(self._mask_slab, F_MUL, None)]
        if not self._geom.supported():
            return self._apply_factors(torch.fft.fft2(self.padding_any(amp * torch.exp(1j * (2 * torch.pi * phs)))), f)
        return ToSpectrumFn.apply(amp, phs, Spec(self._geom, IN_POLAR, OUT_COMPLEX, 2 * torch.pi, tuple(f)))

    # ------------------------------------------------------------------ fused training path (no full spectra in HBM)
    def reconstruct_planes(self, fixed_propagator, POH, target_amp, target_phs01, indices):
        """hat and target amplitude/phase at plane indices[b], equal to
        propagate_POH2Freq_forward + filter_AP2filteredFreq + ..._random_fixed_..._freq2amp of the
        reference step (watermelon.py:219-234) but without materialising the (2B,3,R,C) spectra:
        hat    = crop(ifft2(fft2(pad(e^{i POH})) * (H_z*mask) * (H_d[idx]*mask)))
        target = crop(ifft2(fft2(pad(a e^{i 2 pi phi})) * (H_d[idx]*mask)))"""
        B = POH.shape[0]
        idx = self._colour_index(3 * B, indices)
        f_stack = Factor(self._H_masked, F_MUL, idx)
        hat = self._run(POH, None, IN_PHASE, OUT_ABS_ANGLE, [fixed_propagator._H_factor(3 * B, True), f_stack])
        tgt = self._run(target_amp, target_phs01, IN_POLAR, OUT_ABS_ANGLE, [f_stack], phase_scale=2 * torch.pi)
        return hat[0], hat[1], tgt[0], tgt[1]
