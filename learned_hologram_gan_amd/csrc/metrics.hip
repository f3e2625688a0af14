// PSNR and SSIM of reconstructed amplitudes against the targets, as recorded every batch by the reference's training and
// validation loops (watermelon.py:134-135, 447-456: torchmetrics PeakSignalNoiseRatio() / StructuralSimilarityIndexMeasure()
// with their defaults; torchmetrics itself is not vendored in the reference, so its published definitions are restated):
//   PSNR = 10 log10( (max(max t, 0) - min(min t, 0))^2 / mean (h - t)^2 )                      over the whole batch tensor
//   SSIM : 11x11 Gaussian window (sigma 1.5), data range L = max(range h, range t), c1 = (0.01 L)^2, c2 = (0.03 L)^2,
//          the map is kept where the window lies inside the image (the reflect padding of the definition only reaches the
//          5-pixel border that it crops away again) and averaged over all kept pixels of all planes.
// Three launches on planar (planes, H, W) tensors: range / squared-error partials, SSIM tiles (separable window through LDS, the
// data range is read on the device: no host synchronisation), final reduction.  Reductions have a fixed order.
#include "common.h"

namespace lhg {

constexpr int MT = 32;            // SSIM output tile (MT x MT) per workgroup
constexpr int MH = 5;             // window half width
constexpr int MW = MT + 2 * MH;   // staged tile extent

__constant__ float kGauss[11] = {1.028380084e-03f, 7.598758135e-03f, 3.600077213e-02f, 1.093606895e-01f, 2.130055377e-01f, 2.660117249e-01f,
                                 2.130055377e-01f, 1.093606895e-01f, 3.600077213e-02f, 7.598758135e-03f, 1.028380084e-03f};

__device__ __forceinline__ float wave_min(float v) {
  for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// partial[b] = {min h, max h, min t, max t} (floats) and sse[b] (double)
__global__ __launch_bounds__(256) void metric_range_kernel(const float* __restrict__ h, const float* __restrict__ t, size_t total,
                                                           float* __restrict__ part4, double* __restrict__ part_sse) {
  float mnh = INFINITY, mxh = -INFINITY, mnt = INFINITY, mxt = -INFINITY;
  double sse = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float a = h[i], b = t[i];
    mnh = fminf(mnh, a); mxh = fmaxf(mxh, a); mnt = fminf(mnt, b); mxt = fmaxf(mxt, b);
    const float d = a - b;
    sse += (double)(d * d);
  }
  __shared__ float s4[4][4];
  __shared__ double ss[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  mnh = wave_min(mnh); mxh = wave_max(mxh); mnt = wave_min(mnt); mxt = wave_max(mxt); sse = wave_sum(sse);
  if (lane == 0) { s4[wave][0] = mnh; s4[wave][1] = mxh; s4[wave][2] = mnt; s4[wave][3] = mxt; ss[wave] = sse; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      s4[0][0] = fminf(s4[0][0], s4[w][0]); s4[0][1] = fmaxf(s4[0][1], s4[w][1]);
      s4[0][2] = fminf(s4[0][2], s4[w][2]); s4[0][3] = fmaxf(s4[0][3], s4[w][3]);
      ss[0] += ss[w];
    }
    for (int k = 0; k < 4; ++k) part4[blockIdx.x * 4 + k] = s4[0][k];
    part_sse[blockIdx.x] = ss[0];
  }
}

// scal = {min h, max h, min t, max t, L}, sse_total
__global__ __launch_bounds__(64) void metric_range_final_kernel(const float* __restrict__ part4, const double* __restrict__ part_sse, int nblk,
                                                                float* __restrict__ scal, double* __restrict__ sse_total) {
  float mnh = INFINITY, mxh = -INFINITY, mnt = INFINITY, mxt = -INFINITY;
  double sse = 0;
  for (int b = threadIdx.x; b < nblk; b += 64) {
    mnh = fminf(mnh, part4[b * 4]); mxh = fmaxf(mxh, part4[b * 4 + 1]);
    mnt = fminf(mnt, part4[b * 4 + 2]); mxt = fmaxf(mxt, part4[b * 4 + 3]);
    sse += part_sse[b];
  }
  mnh = wave_min(mnh); mxh = wave_max(mxh); mnt = wave_min(mnt); mxt = wave_max(mxt); sse = wave_sum(sse);
  if (threadIdx.x == 0) {
    scal[0] = mnh; scal[1] = mxh; scal[2] = mnt; scal[3] = mxt;
    scal[4] = fmaxf(mxh - mnh, mxt - mnt);
    *sse_total = sse;
  }
}

// one MT x MT tile of one plane per workgroup; part[b] = sum of the SSIM map over the tile's kept pixels (double)
__global__ __launch_bounds__(256) void ssim_tile_kernel(const float* __restrict__ h, const float* __restrict__ t, int H, int W, int tiles_x,
                                                        int tiles_y, const float* __restrict__ scal, double* __restrict__ part) {
  __shared__ float sh[MW][MW + 1], st[MW][MW + 1];
  __shared__ float hb[5][MW][MT + 1];  // horizontally filtered x, y, xx, yy, xy
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int plane = blockIdx.x / (tiles_x * tiles_y), rem = blockIdx.x - plane * tiles_x * tiles_y;
  const int y0 = (rem / tiles_x) * MT, x0 = (rem - (rem / tiles_x) * tiles_x) * MT;
  const float* hp = h + (size_t)plane * H * W;
  const float* tp = t + (size_t)plane * H * W;
  for (int i = tid; i < MW * MW; i += 256) {
    const int r = i / MW, c = i - r * MW;
    const int yy = y0 - MH + r, xx = x0 - MH + c;
    const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    sh[r][c] = ok ? hp[(size_t)yy * W + xx] : 0.f;
    st[r][c] = ok ? tp[(size_t)yy * W + xx] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < MW * MT; i += 256) {
    const int r = i / MT, c = i - r * MT;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = kGauss[k], x = sh[r][c + k], y = st[r][c + k];
      a0 = fmaf(g, x, a0); a1 = fmaf(g, y, a1); a2 = fmaf(g, x * x, a2); a3 = fmaf(g, y * y, a3); a4 = fmaf(g, x * y, a4);
    }
    hb[0][r][c] = a0; hb[1][r][c] = a1; hb[2][r][c] = a2; hb[3][r][c] = a3; hb[4][r][c] = a4;
  }
  __syncthreads();
  const float L = scal[4];
  const float c1 = (0.01f * L) * (0.01f * L), c2 = (0.03f * L) * (0.03f * L);
  double acc = 0;
  for (int i = tid; i < MT * MT; i += 256) {
    const int r = i / MT, c = i - r * MT;
    const int y = y0 + r, x = x0 + c;
    if (y < MH || y >= H - MH || x < MH || x >= W - MH) continue;  // window must lie inside the image
    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float g = kGauss[k];
#pragma unroll
      for (int q = 0; q < 5; ++q) m[q] = fmaf(g, hb[q][r + k][c], m[q]);
    }
    const float mx = m[0], my = m[1];
    const float vx = m[2] - mx * mx, vy = m[3] - my * my, cxy = m[4] - mx * my;
    const float s = ((2.f * mx * my + c1) * (2.f * cxy + c2)) / ((mx * mx + my * my + c1) * (vx + vy + c2));
    acc += (double)s;
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void metric_final_kernel(const double* __restrict__ part, int nblk, const float* __restrict__ scal,
                                                          const double* __restrict__ sse_total, double n_total, double n_kept,
                                                          float* __restrict__ out2) {
  double s = 0;
  for (int b = threadIdx.x; b < nblk; b += 64) s += part[b];
  s = wave_sum(s);
  if (threadIdx.x == 0) {
    // torchmetrics keeps min_target / max_target states that start at 0.0: range = max(max t, 0) - min(min t, 0)
    const double range_t = fmax((double)scal[3], 0.0) - fmin((double)scal[2], 0.0);
    out2[0] = (float)(10.0 * log10(range_t * range_t / (*sse_total / n_total)));
    out2[1] = (float)(s / n_kept);
  }
}

static int range_blocks(size_t total) {
  const size_t b = (total + 4095) / 4096;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace lhg

using namespace lhg;

extern "C" {

size_t lhg_psnr_ssim_workspace(int planes, int H, int W) {
  const size_t tiles = (size_t)planes * ((H + MT - 1) / MT) * ((W + MT - 1) / MT);
  const size_t nb = (size_t)range_blocks((size_t)planes * H * W);
  // [part4: nb*4 floats][scal: 8 floats][doubles: part_sse nb, sse_total 1, part tiles]
  return (nb * 4 + 8) * sizeof(float) + (nb + 1 + tiles) * sizeof(double) + 16;
}

int lhg_psnr_ssim(const float* hat, const float* tgt, int planes, int H, int W, float* out2, void* ws, size_t ws_bytes, lhg_stream_t s) {
  LHG_REQUIRE(planes > 0 && H > 2 * MH && W > 2 * MH, "psnr_ssim: extents %dx%d too small for the 11x11 window", H, W);
  const size_t need = lhg_psnr_ssim_workspace(planes, H, W);
  if (ws_bytes < need) return fail(LHG_E_WORKSPACE, "psnr_ssim: workspace %zu < %zu", ws_bytes, need);
  const size_t total = (size_t)planes * H * W;
  const int nb = range_blocks(total);
  const int tiles_x = (W + MT - 1) / MT, tiles_y = (H + MT - 1) / MT, tiles = planes * tiles_x * tiles_y;
  float* part4 = static_cast<float*>(ws);
  float* scal = part4 + (size_t)nb * 4;
  uintptr_t dbase = (reinterpret_cast<uintptr_t>(scal + 8) + 15) & ~(uintptr_t)15;
  double* part_sse = reinterpret_cast<double*>(dbase);
  double* sse_total = part_sse + nb;
  double* part = sse_total + 1;
  hipStream_t st = as_stream(s);
  hipLaunchKernelGGL(metric_range_kernel, dim3(nb), dim3(256), 0, st, hat, tgt, total, part4, part_sse);
  hipLaunchKernelGGL(metric_range_final_kernel, dim3(1), dim3(64), 0, st, part4, part_sse, nb, scal, sse_total);
  hipLaunchKernelGGL(ssim_tile_kernel, dim3(tiles), dim3(256), 0, st, hat, tgt, H, W, tiles_x, tiles_y, scal, part);
  const double kept = (double)planes * (H - 2 * MH) * (W - 2 * MH);
  hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(64), 0, st, part, tiles, scal, sse_total, (double)total, kept, out2);
  return check_launch("psnr_ssim");
}

}  // extern "C"
