// Fused reconstruction losses of the generator step, forward and backward.
// ref: watermelon_hologram/loss_func.py:66-98 (total variation), :135-163 (focal sin/cos phase-gradient loss),
//      watermelon.py:418-445 (G_loss: + F.mse_loss).
//
// With S = cat(sin p, cos p), first differences D_w, D_h, e = D(S_hat) - D(S_tgt), d = |e|:
//   focal = sum_w d^2 / (n_w max_w d) + sum_h d^2 / (n_h max_h d)      (weights d/max are detached in the reference,
//                                                                        so d focal / d e = e / (n max))
//   pixel = mean (a_hat - a_tgt)^2
//   tv    = | TV(a_hat) - TV(a_tgt) |,  TV(a) = mean |D_w a| + mean |D_h a|
// One streaming pass over the four (planes, H, W) inputs produces the nine partial reductions per block; a second
// pass produces both input gradients.  Planes are (B*3, H, W) fp32; phases are the raw angles.
#include <algorithm>

#include "common.h"

namespace lhg {

constexpr int NRED = 9;  // sum d_w^2, max d_w, sum d_h^2, max d_h, sse, sum|Dw a_hat|, sum|Dh a_hat|, sum|Dw a_tgt|, sum|Dh a_tgt|

__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) - (v < 0.f); }

__global__ __launch_bounds__(256) void recon_loss_fwd_kernel(const float* __restrict__ ha, const float* __restrict__ ta,
                                                             const float* __restrict__ hp, const float* __restrict__ tp, int planes,
                                                             int H, int W, float* __restrict__ partial /* [gridDim.x][NRED] */) {
  __shared__ float red[4][NRED];
  float acc[NRED];
#pragma unroll
  for (int k = 0; k < NRED; ++k) acc[k] = 0.f;
  const size_t total = (size_t)planes * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float a = ha[i], b = ta[i];
    acc[4] += (a - b) * (a - b);
    float sh, ch, st, ct;
    sincosf(hp[i], &sh, &ch);
    sincosf(tp[i], &st, &ct);
    if (x + 1 < W) {
      float s2, c2, s3, c3;
      sincosf(hp[i + 1], &s2, &c2);
      sincosf(tp[i + 1], &s3, &c3);
      const float es = (s2 - sh) - (s3 - st), ec = (c2 - ch) - (c3 - ct);
      acc[0] += es * es + ec * ec;
      acc[1] = fmaxf(acc[1], fmaxf(fabsf(es), fabsf(ec)));
      acc[5] += fabsf(ha[i + 1] - a);
      acc[7] += fabsf(ta[i + 1] - b);
    }
    if (y + 1 < H) {
      float s2, c2, s3, c3;
      sincosf(hp[i + W], &s2, &c2);
      sincosf(tp[i + W], &s3, &c3);
      const float es = (s2 - sh) - (s3 - st), ec = (c2 - ch) - (c3 - ct);
      acc[2] += es * es + ec * ec;
      acc[3] = fmaxf(acc[3], fmaxf(fabsf(es), fabsf(ec)));
      acc[6] += fabsf(ha[i + W] - a);
      acc[8] += fabsf(ta[i + W] - b);
    }
  }
#pragma unroll
  for (int k = 0; k < NRED; ++k) {
    const bool is_max = (k == 1 || k == 3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float o = __shfl_down(acc[k], off, 64);
      acc[k] = is_max ? fmaxf(acc[k], o) : acc[k] + o;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < NRED) {
    const int k = threadIdx.x;
    const bool is_max = (k == 1 || k == 3);
    float v = red[0][k];
    for (int w = 1; w < 4; ++w) v = is_max ? fmaxf(v, red[w][k]) : v + red[w][k];
    partial[(size_t)blockIdx.x * NRED + k] = v;
  }
}

// sums[9] (double-accumulated) and losses[3] = (focal, pixel, tv)
__global__ __launch_bounds__(64 * NRED) void recon_loss_final_kernel(const float* __restrict__ partial, int nblk, int planes, int H, int W,
                                                                      float* __restrict__ sums, float* __restrict__ losses) {
  // wave k reduces quantity k over the partial blocks: 64 strided lanes, then a fixed-order xor tree
  __shared__ double s[NRED];
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool is_max = (k == 1 || k == 3);
  double v = 0;
  for (int b = lane; b < nblk; b += 64) {
    const double q = partial[(size_t)b * NRED + k];
    v = is_max ? (q > v ? q : v) : v + q;
  }
  for (int m = 32; m >= 1; m >>= 1) {
    const double o = __shfl_xor(v, m, 64);
    v = is_max ? (o > v ? o : v) : v + o;
  }
  if (lane == 0) {
    s[k] = v;
    sums[k] = (float)v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double n_w = 2.0 * planes * H * (W - 1), n_h = 2.0 * planes * (H - 1) * W;  // sin and cos channels
    const double a_w = (double)planes * H * (W - 1), a_h = (double)planes * (H - 1) * W, n = (double)planes * H * W;
    losses[0] = (float)(s[0] / (n_w * s[1]) + s[2] / (n_h * s[3]));  // NaN when hat == target, as in the reference
    losses[1] = (float)(s[4] / n);
    const double tv_hat = s[5] / a_w + s[6] / a_h, tv_tgt = s[7] / a_w + s[8] / a_h;
    losses[2] = (float)fabs(tv_hat - tv_tgt);
  }
}

__global__ __launch_bounds__(256) void recon_loss_bwd_kernel(const float* __restrict__ ha, const float* __restrict__ ta,
                                                             const float* __restrict__ hp, const float* __restrict__ tp, int planes,
                                                             int H, int W, const float* __restrict__ sums,
                                                             const float* __restrict__ upstream /* d(focal, pixel, tv) */,
                                                             float* __restrict__ g_ha, float* __restrict__ g_hp) {
  const double n_wd = 2.0 * planes * H * (W - 1), n_hd = 2.0 * planes * (H - 1) * W;
  const float a_w = (float)planes * H * (W - 1), a_h = (float)planes * (H - 1) * W, n = (float)planes * H * W;
  const float gf = upstream[0], gp = upstream[1], gt = upstream[2];
  const float kw = gf / (float)(n_wd * (double)sums[1]), kh = gf / (float)(n_hd * (double)sums[3]);
  const float tv_hat = sums[5] / a_w + sums[6] / a_h, tv_tgt = sums[7] / a_w + sums[8] / a_h;
  const float tsign = gt * sgnf(tv_hat - tv_tgt);
  const size_t total = (size_t)planes * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H);
    const float a = ha[i];
    float ga = gp * 2.f * (a - ta[i]) / n;
    float sh, ch, st, ct;
    sincosf(hp[i], &sh, &ch);
    sincosf(tp[i], &st, &ct);
    float Gs = 0.f, Gc = 0.f;  // d focal / d sin(hat), d cos(hat) at this pixel (before the g_focal factor folded in kw/kh)
    auto edge = [&](size_t j, float k, float dir) {  // difference between pixel i and its neighbour j; dir=+1: j is "next"
      float s2, c2, s3, c3;
      sincosf(hp[j], &s2, &c2);
      sincosf(tp[j], &s3, &c3);
      // e = (S[next] - S[cur])_hat - (...)_tgt ; d focal/d S[next] = +k e, d focal/d S[cur] = -k e
      const float es = dir * ((s2 - sh) - (s3 - st)), ec = dir * ((c2 - ch) - (c3 - ct));
      Gs -= dir * k * es;
      Gc -= dir * k * ec;
    };
    if (x + 1 < W) { edge(i + 1, kw, 1.f); ga -= tsign * sgnf(ha[i + 1] - a) / a_w; }
    if (x > 0)     { edge(i - 1, kw, -1.f); ga += tsign * sgnf(a - ha[i - 1]) / a_w; }
    if (y + 1 < H) { edge(i + W, kh, 1.f); ga -= tsign * sgnf(ha[i + W] - a) / a_h; }
    if (y > 0)     { edge(i - W, kh, -1.f); ga += tsign * sgnf(a - ha[i - W]) / a_h; }
    g_ha[i] = ga;
    g_hp[i] = ch * Gs - sh * Gc;
  }
}

static int loss_blocks(size_t total) { return (int)std::min<size_t>(1024, std::max<size_t>(1, (total + 255) / 256)); }

}  // namespace lhg

using namespace lhg;

extern "C" {

int lhg_recon_loss_blocks(int planes, int H, int W) { return loss_blocks((size_t)planes * H * W); }

int lhg_recon_loss_forward(const float* hat_amp, const float* tgt_amp, const float* hat_phs, const float* tgt_phs, int planes, int H,
                           int W, float* sums9, float* losses3, float* ws, lhg_stream_t s) {
  LHG_REQUIRE(planes > 0 && H > 1 && W > 1, "recon_loss_forward: bad extents");
  const int nblk = loss_blocks((size_t)planes * H * W);
  hipLaunchKernelGGL(recon_loss_fwd_kernel, dim3(nblk), dim3(256), 0, as_stream(s), hat_amp, tgt_amp, hat_phs, tgt_phs, planes, H, W, ws);
  hipLaunchKernelGGL(recon_loss_final_kernel, dim3(1), dim3(64 * NRED), 0, as_stream(s), ws, nblk, planes, H, W, sums9, losses3);
  return check_launch("recon_loss_forward");
}

int lhg_recon_loss_backward(const float* hat_amp, const float* tgt_amp, const float* hat_phs, const float* tgt_phs, int planes, int H,
                            int W, const float* sums9, const float* upstream3, float* g_hat_amp, float* g_hat_phs, lhg_stream_t s) {
  LHG_REQUIRE(planes > 0 && H > 1 && W > 1, "recon_loss_backward: bad extents");
  const int nblk = loss_blocks((size_t)planes * H * W);
  hipLaunchKernelGGL(recon_loss_bwd_kernel, dim3(nblk), dim3(256), 0, as_stream(s), hat_amp, tgt_amp, hat_phs, tgt_phs, planes, H, W,
                     sums9, upstream3, g_hat_amp, g_hat_phs);
  return check_launch("recon_loss_backward");
}

}  // extern "C"
