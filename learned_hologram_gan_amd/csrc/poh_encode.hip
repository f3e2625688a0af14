// AP2POH tail: per-colour radially symmetric 3x3 stencil on Re and Im of the
// back-propagated field, amplitude normalisation by 1.01 * per-plane max, double-phase encoding
// on a unit checkerboard — forward and backward.  ref: AP2POH.py:105-116, utilities.py:53-66,
// neural_network_components.py:68-75.  Streaming kernels, 8-byte lanes on interleaved complex64.
//
//   mod = stencil_c(field) + bias_c (same real stencil on Re and Im)       c = plane % 3
//   A = |mod|, M = max_plane A, a = A / (1.01 M), phi = angle(mod)
//   POH = phi + s * acos(a),  s = +1 where (x+y) even, -1 where odd
#include <algorithm>

#include "common.h"

namespace lhg {

// per-plane peak packed as (float bits of A) << 32 | (0xFFFFFFFF - pixel index): a 64-bit atomicMax keeps the
// largest amplitude and, among equal amplitudes, the smallest index (what two successive torch.max calls select).
__device__ __forceinline__ unsigned long long pack_peak(float a, unsigned idx) {
  return ((unsigned long long)__float_as_uint(a) << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}
__device__ __forceinline__ float peak_value(unsigned long long p) { return __uint_as_float((unsigned)(p >> 32)); }
__device__ __forceinline__ unsigned peak_index(unsigned long long p) { return 0xFFFFFFFFu - (unsigned)(p & 0xFFFFFFFFull); }

__device__ __forceinline__ float stencil_weight(int dy, int dx, float w0, float w1, float w2) {
  return (dy == 0 && dx == 0) ? w0 : ((dy == 0 || dx == 0) ? w1 : w2);
}

// The nine taps are loaded first — from clamped coordinates, a tap outside the image replaced by zero — and accumulated afterwards in the
// same order: with a `continue` per tap every load sat behind its own branch and the kernel ran at 0.36 TB/s on the 4K frame (round 4).
__device__ __forceinline__ float2 stencil_at(const float2* __restrict__ f, int y, int x, int rows, int cols, float w0, float w1, float w2) {
  float2 v[3][3];
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    const int yy = y + dy;
    const bool oky = (unsigned)yy < (unsigned)rows;
    const size_t rowoff = (size_t)(oky ? yy : y) * cols;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int xx = x + dx;
      const bool ok = oky && (unsigned)xx < (unsigned)cols;
      const float2 l = f[rowoff + (ok ? xx : x)];
      v[dy + 1][dx + 1] = ok ? l : make_float2(0.f, 0.f);
    }
  }
  float2 acc = make_float2(0.f, 0.f);
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const float w = stencil_weight(dy, dx, w0, w1, w2);
      acc.x += w * v[dy + 1][dx + 1].x;
      acc.y += w * v[dy + 1][dx + 1].y;
    }
  return acc;
}

__global__ __launch_bounds__(256) void symconv_field_kernel(const float2* __restrict__ field, int planes, int rows, int cols,
                                                            const float* __restrict__ taps, const float* __restrict__ bias,
                                                            float2* __restrict__ mod, unsigned long long* __restrict__ peak) {
  const int plane = blockIdx.y;
  const int colour = plane % 3;
  const float w0 = taps[colour * 3 + 0], w1 = taps[colour * 3 + 1], w2 = taps[colour * 3 + 2], b = bias[colour];
  const float2* f = field + (size_t)plane * rows * cols;
  unsigned long long best = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += gridDim.x * blockDim.x) {
    const int y = i / cols, x = i - y * cols;
    float2 acc = stencil_at(f, y, x, rows, cols, w0, w1, w2);
    acc.x += b;
    acc.y += b;
    mod[(size_t)plane * rows * cols + i] = acc;
    best = max(best, pack_peak(hypotf(acc.x, acc.y), (unsigned)i));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = max(best, (unsigned long long)__shfl_down((long long)best, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(peak + plane, best);
}

__global__ __launch_bounds__(256) void double_phase_kernel(const float2* __restrict__ mod, const unsigned long long* __restrict__ peak,
                                                           int planes, int rows, int cols, float* __restrict__ poh) {
  const int plane = blockIdx.y;
  const float denom = peak_value(peak[plane]) * 1.01f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += gridDim.x * blockDim.x) {
    const int y = i / cols, x = i - y * cols;
    const float2 z = mod[(size_t)plane * rows * cols + i];
    const float a = hypotf(z.x, z.y) / denom;
    const float phi = atan2f(z.y, z.x);
    const float ac = acosf(a);
    poh[(size_t)plane * rows * cols + i] = ((x + y) & 1) ? phi - ac : phi + ac;
  }
}

// ---- backward, stage 1: d POH -> d mod without the max term; per-block partial of S = sum g_a * a
__global__ __launch_bounds__(256) void poh_bwd_stage1(const float* __restrict__ g_poh, const float2* __restrict__ mod,
                                                      const unsigned long long* __restrict__ peak, int planes, int rows, int cols,
                                                      float2* __restrict__ g_mod, float* __restrict__ partial /* [planes][gridDim.x] */) {
  __shared__ float red[4];
  const int plane = blockIdx.y;
  const float M = peak_value(peak[plane]);
  const float denom = M * 1.01f;
  float s_acc = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += gridDim.x * blockDim.x) {
    const int y = i / cols, x = i - y * cols;
    const size_t o = (size_t)plane * rows * cols + i;
    const float2 z = mod[o];
    const float g = g_poh[o];
    const float A2 = z.x * z.x + z.y * z.y;
    float2 out = make_float2(0.f, 0.f);
    if (A2 > 0.f) {
      const float A = sqrtf(A2);
      const float a = A / denom;
      const float sgn = ((x + y) & 1) ? -1.f : 1.f;
      const float g_a = -sgn * g / sqrtf(fmaxf(1.f - a * a, 1e-30f));  // d acos
      const float g_A = g_a / denom;
      s_acc += g_a * a;
      // d|z| -> z/|z| ; d angle -> (-im, re)/|z|^2
      out.x = g_A * z.x / A - g * z.y / A2;
      out.y = g_A * z.y / A + g * z.x / A2;
    }
    g_mod[o] = out;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s_acc += __shfl_down(s_acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s_acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)plane * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- stage 2: the max term.  a = A/(1.01 M) => dL/dM = -sum(g_a a)/M, routed to the arg-max pixel along z/|z|
__global__ void poh_bwd_stage2(const float2* __restrict__ mod, const unsigned long long* __restrict__ peak, const float* __restrict__ partial,
                               int nblk, int planes, int rows, int cols, float2* __restrict__ g_mod) {
  const int plane = blockIdx.x * blockDim.x + threadIdx.x;
  if (plane >= planes) return;
  double S = 0;
  for (int b = 0; b < nblk; ++b) S += partial[(size_t)plane * nblk + b];
  const float M = peak_value(peak[plane]);
  if (!(M > 0.f)) return;
  const size_t o = (size_t)plane * rows * cols + peak_index(peak[plane]);
  const float2 z = mod[o];
  const float gM = (float)(-S / (double)M);
  g_mod[o].x += gM * z.x / M;
  g_mod[o].y += gM * z.y / M;
}

// ---- stage 3: stencil backward.  g_field = stencil_c(g_mod) (the kernel is symmetric); per-block partials of the
//      four parameter gradients of this plane's colour: centre / edge / corner taps and the bias.
__global__ __launch_bounds__(256) void symconv_bwd_kernel(const float2* __restrict__ g_mod, const float2* __restrict__ field, int planes,
                                                          int rows, int cols, const float* __restrict__ taps,
                                                          float2* __restrict__ g_field, float* __restrict__ partial /* [planes][gridDim.x][4] */) {
  __shared__ float red[4][4];
  const int plane = blockIdx.y;
  const int colour = plane % 3;
  const float w0 = taps[colour * 3 + 0], w1 = taps[colour * 3 + 1], w2 = taps[colour * 3 + 2];
  const float2* gm = g_mod + (size_t)plane * rows * cols;
  const float2* f = field + (size_t)plane * rows * cols;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += gridDim.x * blockDim.x) {
    const int y = i / cols, x = i - y * cols;
    g_field[(size_t)plane * rows * cols + i] = stencil_at(gm, y, x, rows, cols, w0, w1, w2);
    const float2 g = gm[i];
    const float2 c = stencil_at(f, y, x, rows, cols, 1.f, 0.f, 0.f);
    const float2 e = stencil_at(f, y, x, rows, cols, 0.f, 1.f, 0.f);
    const float2 k = stencil_at(f, y, x, rows, cols, 0.f, 0.f, 1.f);
    acc[0] += g.x * c.x + g.y * c.y;
    acc[1] += g.x * e.x + g.y * e.y;
    acc[2] += g.x * k.x + g.y * k.y;
    acc[3] += g.x + g.y;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_down(acc[q], off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = acc[q];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int q = threadIdx.x;
    partial[((size_t)plane * gridDim.x + blockIdx.x) * 4 + q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
  }
}

static int plane_blocks(int rows, int cols) { return std::min(64, (rows * cols + 255) / 256); }

// ------------------------------------------------------------------ polar <-> cartesian Jacobians at the two ends of the angular-spectrum operator
// The operator itself is linear in the complex field (asm_ops.py); what its backward adds are the pointwise Jacobians of |z| / angle(z)
// at the output and of a * exp(i s phi) at the input.  As torch expressions they were ~25 elementwise launches per operator call
// (55 per train step); here one launch each.  Same formulas, same guards (asm_ops._output_cotangent / _input_cotangent).
//   gz = scale * ( g_a * z / |z|  +  g_b * (-Im z, Re z) / |z|^2 ),   zero where z == 0      (PyTorch convention: dL/dRe + i dL/dIm)
__global__ __launch_bounds__(256) void polar_out_cotangent_kernel(const float2* __restrict__ z, const float* __restrict__ g_a,
                                                                  const float* __restrict__ g_b, float scale, float2* __restrict__ gz, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float2 v = z[i];
    const float mag2 = v.x * v.x + v.y * v.y;
    float2 o = make_float2(0.f, 0.f);
    if (mag2 > 0.f) {
      if (g_a) {
        const float inv = rsqrtf(fmaxf(mag2, 1e-45f)), ga = g_a[i];
        o.x = ga * v.x * inv;
        o.y = ga * v.y * inv;
      }
      if (g_b) {
        const float inv2 = 1.0f / fmaxf(mag2, 1e-45f), gb = g_b[i];
        o.x += -gb * v.y * inv2;
        o.y += gb * v.x * inv2;
      }
    }
    gz[i] = make_float2(o.x * scale, o.y * scale);
  }
}
// u = a * exp(i s phi) (polar: a amplitude, phi phase) or exp(i s a) (phase only: phi == nullptr), g = pre_scale * g_in:
//   radial = Re(conj(e) g), tangential = Im(conj(e) g), e = exp(i s phi);   polar: (g_amp, g_phi) = (radial, tangential a s);  phase: g_a = tangential s
__global__ __launch_bounds__(256) void polar_in_cotangent_kernel(const float2* __restrict__ g_in, const float* __restrict__ a,
                                                                 const float* __restrict__ phi, float phase_scale, float pre_scale,
                                                                 float* __restrict__ g_a, float* __restrict__ g_phi, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float2 g = g_in[i];
    const float gre = g.x * pre_scale, gim = g.y * pre_scale;
    const float av = a[i];
    const float ph = (phi ? phi[i] : av) * phase_scale;
    const float c = cosf(ph), sn = sinf(ph);
    const float radial = gre * c + gim * sn, tangential = -gre * sn + gim * c;
    if (phi) {
      g_a[i] = radial;
      g_phi[i] = tangential * av * phase_scale;
    } else {
      g_a[i] = tangential * phase_scale;
    }
  }
}

}  // namespace lhg

using namespace lhg;

extern "C" {

int lhg_poh_partial_blocks(int rows, int cols) { return plane_blocks(rows, cols); }

int lhg_symconv_field(const float* field, int planes, int rows, int cols, const float* taps, const float* bias, float* mod,
                      unsigned long long* plane_peak, lhg_stream_t s) {
  LHG_REQUIRE(planes > 0 && rows > 0 && cols > 0, "symconv_field: bad extents");
  hipLaunchKernelGGL(symconv_field_kernel, dim3(plane_blocks(rows, cols), planes), dim3(256), 0, as_stream(s),
                     reinterpret_cast<const float2*>(field), planes, rows, cols, taps, bias, reinterpret_cast<float2*>(mod), plane_peak);
  return check_launch("symconv_field");
}

int lhg_double_phase_encode(const float* mod, const unsigned long long* plane_peak, int planes, int rows, int cols, float* poh,
                            lhg_stream_t s) {
  LHG_REQUIRE(planes > 0 && rows > 0 && cols > 0, "double_phase_encode: bad extents");
  hipLaunchKernelGGL(double_phase_kernel, dim3(plane_blocks(rows, cols), planes), dim3(256), 0, as_stream(s),
                     reinterpret_cast<const float2*>(mod), plane_peak, planes, rows, cols, poh);
  return check_launch("double_phase_encode");
}

int lhg_double_phase_encode_backward(const float* g_poh, const float* mod, const unsigned long long* plane_peak, int planes, int rows,
                                     int cols, float* g_mod, float* ws, lhg_stream_t s) {
  LHG_REQUIRE(planes > 0 && rows > 0 && cols > 0, "double_phase_encode_backward: bad extents");
  const int nblk = plane_blocks(rows, cols);
  hipLaunchKernelGGL(poh_bwd_stage1, dim3(nblk, planes), dim3(256), 0, as_stream(s), g_poh, reinterpret_cast<const float2*>(mod),
                     plane_peak, planes, rows, cols, reinterpret_cast<float2*>(g_mod), ws);
  hipLaunchKernelGGL(poh_bwd_stage2, dim3((planes + 63) / 64), dim3(64), 0, as_stream(s), reinterpret_cast<const float2*>(mod), plane_peak,
                     ws, nblk, planes, rows, cols, reinterpret_cast<float2*>(g_mod));
  return check_launch("double_phase_encode_backward");
}

int lhg_symconv_field_backward(const float* g_mod, const float* field, int planes, int rows, int cols, const float* taps, float* g_field,
                               float* partial, lhg_stream_t s) {
  LHG_REQUIRE(planes > 0 && rows > 0 && cols > 0, "symconv_field_backward: bad extents");
  hipLaunchKernelGGL(symconv_bwd_kernel, dim3(plane_blocks(rows, cols), planes), dim3(256), 0, as_stream(s),
                     reinterpret_cast<const float2*>(g_mod), reinterpret_cast<const float2*>(field), planes, rows, cols, taps,
                     reinterpret_cast<float2*>(g_field), partial);
  return check_launch("symconv_field_backward");
}

int lhg_polar_output_cotangent(const float* z, const float* g_abs, const float* g_angle, float scale, float* gz, long long n, lhg_stream_t s) {
  LHG_REQUIRE(n > 0 && z != nullptr && gz != nullptr && (g_abs != nullptr || g_angle != nullptr), "polar_output_cotangent: missing tensor");
  const int grid = (int)std::min<long long>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(polar_out_cotangent_kernel, dim3(grid), dim3(256), 0, as_stream(s), reinterpret_cast<const float2*>(z), g_abs, g_angle, scale,
                     reinterpret_cast<float2*>(gz), (size_t)n);
  return check_launch("polar_output_cotangent");
}

int lhg_polar_input_cotangent(const float* g_in, const float* a, const float* phi, float phase_scale, float pre_scale, float* g_a, float* g_phi,
                              long long n, lhg_stream_t s) {
  LHG_REQUIRE(n > 0 && g_in != nullptr && a != nullptr && g_a != nullptr && (phi == nullptr || g_phi != nullptr), "polar_input_cotangent: missing tensor");
  const int grid = (int)std::min<long long>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(polar_in_cotangent_kernel, dim3(grid), dim3(256), 0, as_stream(s), reinterpret_cast<const float2*>(g_in), a, phi, phase_scale,
                     pre_scale, g_a, g_phi, (size_t)n);
  return check_launch("polar_input_cotangent");
}

}  // extern "C"
