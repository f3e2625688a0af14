// Implicit-GEMM convolution engine for gfx950 (MI355X), fp32 in / fp32 accumulate on the
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chain, 157 TFLOP/s dense peak).
//
// Every convolution-like op of the hot path (Conv2d 3x3 s1/s2, 1x1, ConvTranspose2d 2x2 s2,
// their input gradients and their weight gradients) is one of
// two GEMM shapes over a common "gather geometry":
//
//   forward-type   out[pix(p)][n]   = sum_t sum_k  in[src(p,t)][k] * Wp[t][n][k]      (gg_kernel)
//   wgrad-type     dW[t][m][n]      = sum_p        in[src(p,t)][m] * gout[pix(p)][n]  (wg_kernel)
//
// p runs over a sub-grid of output pixels (all of them, or one 2x2 parity class for the
// stride-2 transposed forms), src(p,t) = istep*(i,j) + (dy[t],dx[t]) with zero fill outside the
// image.  No im2col buffer exists anywhere: the gather happens while staging LDS tiles.
//
// Tiling (wave64, 4 waves / workgroup):
//   gg: block tile BM pixels x BN channels x 32 k; A tile [BM][32] and B tile [BN][32] are both
//       K-contiguous in global memory, staged through registers into LDS rows of 36 floats
//       (16-byte aligned, conflict-free for ds_read_b128: 36*m mod 64 is injective on 16 rows).
//       A lane reads 4 consecutive k per ds_read_b128 and feeds them to 4 successive MFMAs; the
//       k -> (instruction, lane-half) assignment is a permutation shared by A and B, which the
//       k-sum does not see.  Register prefetch of the next K step overlaps the MFMAs.
//   wg: block tile 64 x 64 channels, K = 32 pixels per step, LDS tiles [32 px][64 ch]; lanes read
//       consecutive channels (conflict-free ds_read_b32).  K (pixels) is split over blockIdx.z
//       into partial slabs that lhg_wgrad_reduce sums (deterministic, no float atomics).
#include <algorithm>
#include <array>
#include <cstdlib>
#include <map>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "common.h"
#include "wg6_api.h"

namespace lhg {

struct Geom {
  int N, Hi, Wi, Ci, ldi;  // gathered tensor
  int Ho, Wo, Co, ldo;     // scattered tensor (full extents)
  int gh, gw;              // sub-grid points per image
  int oy0, ox0, ostep;     // out pixel = (oy0 + ostep*i, ox0 + ostep*j)
  int istep;               // in pixel  = (istep*i + dy[t], istep*j + dx[t])
  int T;
  int dy[9], dx[9], ws[9];
  int M;                   // N*gh*gw
};

struct GGParams {
  Geom g;
  const float* in;
  const float* wp;  // [slab][rows_pad][Ci]
  float* out;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;
  int ldres, rows_pad, act, planar_out;
  float slope;
  int xcd;  // 1: XCD-contiguous tile order (xcd_contiguous)
  const float* a_amax;  // fp16-split mode: max|in| over the gathered tensor (device, lhg_absmax) and max|w| (behind the weight panels)
  const float* w_amax;
  float* out_amax;      // fp16-split mode, optional: max-accumulates max|out| (NHWC outputs)
  float* stat_part;     // optional: BatchNorm statistics' partial rows [consumer-wave row of the tiling][2][Co] (gg_epilogue.inc, STATS)
  // split K (gg3s_kernel / gg4s_kernel, fp16-split mode): ks > 1 cuts the 32-channel chunks of the K axis into ks ranges of ks_chunks
  // chunks; the grid holds ks copies of the tile grid (range-major), every workgroup leaves its RAW accumulators in
  // ks_slab[range][tile row][rows_pad] and gg_splitk_finish_kernel sums the ranges in order and runs the epilogue.
  int ks, ks_chunks;
  float* ks_slab;
  int ks_off;  // set by launch_gg_classes: one launch of several over the same output (parity classes) never splits
  // thin residual (gg4s_kernel with BN = 64, lhg_conv2d_forward_thin_res): instead of reading a residual tensor the epilogue evaluates a 1x1
  // convolution of a thin NCHW tensor, res[pixel][col] = tres_b[col] + sum_{c < tres_c} tres_x[n][c][hw] * tres_w[col * tres_c + c]
  const float* tres_x; const float* tres_w; const float* tres_b;
  int tres_c, tres_plane;
  int tap_of[9];        // gg4s_kernel: tap index of the 3x3 offset (dy + 1) * 3 + (dx + 1)
  int strip_rev;        // gg4s_kernel: tap_of is the reversed map (input gradient): walk the offsets downwards = taps upwards
  int prio; // gg3s_kernel: 0 no s_setprio, 1 consumers (MFMA waves) raised, 2 producers (load / split waves) raised
  // gg3s_kernel, merged launch of the output-parity classes of a stride-2 input gradient / a 2x2 transposed conv (same tensors,
  // different sub-grids and tap sets): class c > 0 has geometry gc[c - 1] and owns workgroups cblk[c] .. cblk[c + 1] - 1 (cblk[0] = 0).
  int ncls;             // 0 / 1: `g` only
  unsigned cblk[5];
  Geom gc[3];
};

struct WGParams {
  Geom g;
  const float* in;    // [.., Ci]  -> m
  const float* gout;  // [.., Co]  -> n
  float* slabs;       // [S][Tslabs][m_pad][n_pad]
  int m_pad, n_pad, Tslabs, kchunk;  // kchunk: pixels per split (multiple of 32)
  int xcd;
  const float* in_amax;    // fp16-split mode: PER-CHANNEL max|in| (Ci floats) and max|gout| (Co floats), device, lhg_channel_absmax
  const float* gout_amax;
};

// (xcd_contiguous: common.h)
struct Block3 { int x, y, z; };
__device__ __forceinline__ Block3 xcd_block3(bool enabled) {
  if (!enabled) return {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  const unsigned gx = gridDim.x, gy = gridDim.y;
  const unsigned v = xcd_contiguous(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
  const unsigned z = v / (gx * gy), rem = v - z * gx * gy;
  return {(int)(rem % gx), (int)(rem / gx), (int)z};
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

constexpr int BK = 32;
constexpr int LDS_ROW = BK + 4;  // floats

// ------------------------------------------------------------------------------------ gg_kernel
template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void gg_kernel(const GGParams p) {
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int A_LOADS = BM * 8 / 256, B_LOADS = BN * 8 / 256;
  static_assert(WGM * WGN == 4 && TM >= 1 && TN >= 1, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_ROW];
  float* sA = smem;
  float* sB = smem + BM * LDS_ROW;

  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int n_tiles = p.rows_pad / BN;
  const int bid = p.xcd ? (int)xcd_contiguous(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int mt = bid / n_tiles, nt = bid - mt * n_tiles;
  const int ghw = g.gh * g.gw;

  // per-thread gather rows
  int a_n[A_LOADS], a_y[A_LOADS], a_x[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    const int m = mt * BM + (tid >> 3) + 32 * i;
    if (m < g.M) {
      const int n = m / ghw, rem = m - n * ghw;
      const int gi = rem / g.gw, gj = rem - gi * g.gw;
      a_n[i] = n;
      a_y[i] = gi * g.istep;
      a_x[i] = gj * g.istep;
    } else {
      a_n[i] = -1;
      a_y[i] = a_x[i] = 0;
    }
  }
  const int c4 = (tid & 7) * 4;
  const int kch = g.Ci / BK;
  const int steps = g.T * kch;

  f32x4 ra[A_LOADS], rb[B_LOADS];
  auto gload = [&](int s) {
    const int t = s / kch, kc = s - t * kch;
    const int dy = g.dy[t], dx = g.dx[t];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int iy = a_y[i] + dy, ix = a_x[i] + dx;
      const bool ok = a_n[i] >= 0 && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(p.in + ((size_t)(a_n[i] * g.Hi + iy) * g.Wi + ix) * g.ldi + kc * BK + c4);
      ra[i] = v;
    }
    const float* wb = p.wp + ((size_t)g.ws[t] * p.rows_pad + nt * BN) * g.Ci + kc * BK + c4;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      rb[i] = *reinterpret_cast<const f32x4*>(wb + (size_t)((tid >> 3) + 32 * i) * g.Ci);
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) *reinterpret_cast<f32x4*>(sA + ((tid >> 3) + 32 * i) * LDS_ROW + c4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) *reinterpret_cast<f32x4*>(sB + ((tid >> 3) + 32 * i) * LDS_ROW + c4) = rb[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const float* fa = sA + (wm * TM * 32 + lr) * LDS_ROW + 4 * lh;
  const float* fb = sB + (wn * TN * 32 + lr) * LDS_ROW + 4 * lh;

  gload(0);
  sstore();
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) gload(s + 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(fa + i * 32 * LDS_ROW + 8 * q);
#pragma unroll
      for (int i = 0; i < TN; ++i) b[i] = *reinterpret_cast<const f32x4*>(fb + i * 32 * LDS_ROW + 8 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i][e], b[j][e], acc[i][j]);
    }
    __syncthreads();
    if (s + 1 < steps) sstore();
    __syncthreads();
  }

  // ---- epilogue: scatter table for the BM rows of this tile, then fused bias/affine/residual/act
  int* srow = reinterpret_cast<int*>(smem);
  for (int r = tid; r < BM; r += 256) {
    const int m = mt * BM + r;
    int pix = -1;
    if (m < g.M) {
      const int n = m / ghw, rem = m - n * ghw;
      const int gi = rem / g.gw, gj = rem - gi * g.gw;
      pix = (n * g.Ho + g.oy0 + g.ostep * gi) * g.Wo + g.ox0 + g.ostep * gj;
    }
    srow[r] = pix;
  }
  __syncthreads();
  const int plane = g.Ho * g.Wo;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = nt * BN + (wn * TN + j) * 32 + lr;
    if (col >= g.Co) continue;
    const float bias = p.bias ? p.bias[col] : 0.f;
    const float sc = p.scale ? p.scale[col] : 1.f;
    const float sh = p.shift ? p.shift[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int pix = srow[row];
        if (pix < 0) continue;
        float v = acc[i][j][r] + bias;
        v = v * sc + sh;
        if (p.res) v += p.res[(size_t)pix * p.ldres + col];
        v = apply_act(v, p.act, p.slope);
        if (p.planar_out) {
          const int n = pix / plane, hw = pix - n * plane;
          p.out[((size_t)n * g.Co + col) * plane + hw] = v;
        } else {
          p.out[(size_t)pix * g.ldo + col] = v;
        }
      }
    }
  }
}

#include "gg_epilogue.inc"
#include "gg2_kernel.inc"
#include "gg2b_kernel.inc"
#include "gg3s_kernel.inc"
#include "gg4s_kernel.inc"

// ------------------------------------------------------------------------------------ split-K finish
// The layers at the bottom of the UNet (24^2 x 1024 channels at batch 4: 2304 pixels) have 144 output tiles of 128 x 128 for 256 CUs and
// a K axis of 288 steps: one workgroup per tile leaves almost half of the chip idle for the whole launch.  With GGParams::ks the K axis
// is cut into ks ranges of whole 32-channel chunks (a FUNCTION OF THE GEOMETRY, never of a timing: every tiling variant of the launch
// sums the same ranges in the same order, so the tuner's choice still does not show in the bits), each workgroup stores raw
// accumulators, and this kernel adds the ranges in ascending order and runs gg_epilogue's arithmetic — scales undone, + bias, * scale
// + shift, + residual, activation, max|.| — plus, when asked (stat), the BatchNorm statistics' partial rows (one per workgroup of
// SK_ROWS slab rows: sum and sum of squares of the bias-free value over its valid pixels, rows in ascending order).
constexpr int SK_ROWS = 16;  // slab rows per workgroup (x 64 columns: 256 threads, four columns each)
struct SplitKFinish {
  const float* slab;
  long long slab_stride;   // floats between two ranges: m_rows * rows_pad
  int ks, m_rows, rows_pad;
  int strip;               // rows are PADDED pixel coordinates (gg4s_kernel: every image row two pixels longer)
  int N, H, W;             // strip: image extents (input = output extents)
  int M, gh, gw, Ho, Wo, oy0, ox0, ostep;  // otherwise: the sub-grid of output pixels
  float* out; int ldo, Co;
  const float* bias; const float* scale; const float* shift; const float* res; int ldres; int act; float slope;
  const float* a_amax; const float* w_amax; float* out_amax;
  float* stat;             // [gridDim.x][2][Co] or null
};
__global__ __launch_bounds__(256) void gg_splitk_finish_kernel(const SplitKFinish q) {
  __shared__ float red[2][SK_ROWS][64];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m = blockIdx.x * SK_ROWS + ty, c0 = blockIdx.y * 64 + tx * 4;
  long long pix = -1;
  if (m < q.m_rows) {
    if (q.strip) {
      const int Wp = q.W + 2;
      const long long Mp = (long long)q.N * q.H * Wp;
      if (m < Mp) {
        const int n = m / (q.H * Wp), rem = m - n * (q.H * Wp);
        const int y = rem / Wp, xp = rem - y * Wp;
        if (xp >= 1 && xp <= q.W) pix = ((long long)n * q.Ho + y) * q.Wo + xp - 1;
      }
    } else if (m < q.M) {
      const int ghw = q.gh * q.gw;
      const int n = m / ghw, rem = m - n * ghw;
      const int gi = rem / q.gw, gj = rem - gi * q.gw;
      pix = ((long long)n * q.Ho + q.oy0 + q.ostep * gi) * q.Wo + q.ox0 + q.ostep * gj;
    }
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (pix >= 0 && c0 < q.rows_pad) {
    const float* src = q.slab + (size_t)m * q.rows_pad + c0;
    for (int z = 0; z < q.ks; ++z) acc += *reinterpret_cast<const f32x4*>(src + (size_t)z * q.slab_stride);
  }
  const float un_a = split_unscale(operand_amax(q.a_amax)), un_w = split_unscale(*q.w_amax);
  float d[4], amax = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int col = c0 + e;
    const bool ok = pix >= 0 && col < q.Co;
    d[e] = ok ? (acc[e] * un_a) * un_w : 0.f;
    if (ok) {
      float v = __builtin_fmaf(acc[e] * un_a, un_w, q.bias ? q.bias[col] : 0.f);
      v = __builtin_fmaf(v, q.scale ? q.scale[col] : 1.f, q.shift ? q.shift[col] : 0.f);
      if (q.res) v += q.res[(size_t)pix * q.ldres + col];
      v = apply_act(v, q.act, q.slope);
      q.out[(size_t)pix * q.ldo + col] = v;
      amax = fmaxf(amax, fabsf(v));
    }
  }
  if (q.out_amax) {  // (uniform)
    unsigned mx = __float_as_uint(amax);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o, 64));
    unsigned* slot = reinterpret_cast<unsigned*>(q.out_amax);
    if ((threadIdx.x & 63) == 0 && mx > *reinterpret_cast<volatile unsigned*>(slot)) atomicMax(slot, mx);
  }
  if (q.stat) {  // (uniform)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[0][ty][tx * 4 + e] = d[e];
      red[1][ty][tx * 4 + e] = d[e] * d[e];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      const int k = threadIdx.x >> 6, cl = threadIdx.x & 63, col = blockIdx.y * 64 + cl;
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < SK_ROWS; ++r) t += red[k][r][cl];
      if (col < q.Co) q.stat[((size_t)blockIdx.x * 2 + k) * q.Co + col] = t;
    }
  }
}

// ------------------------------------------------------------------------------------ wg_kernel
constexpr int WG_TILE = 64;
constexpr int WG_LDS_ROW = WG_TILE + 4;

__global__ __launch_bounds__(256, 2) void wg_kernel(const WGParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * WG_LDS_ROW];
  float* sA = smem;                    // [32 px][64 m]
  float* sB = smem + BK * WG_LDS_ROW;  // [32 px][64 n]
  const Geom& g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int n_tiles = p.n_pad / WG_TILE;
  const Block3 blk = xcd_block3(p.xcd != 0);
  const int mt = blk.x / n_tiles, nt = blk.x - mt * n_tiles;
  const int t = blk.y;
  const int split = blk.z;
  const int ghw = g.gh * g.gw;
  const int dy = g.dy[t], dx = g.dx[t];

  const int p_begin = split * p.kchunk;
  const int p_end = min(g.M, p_begin + p.kchunk);
  const int steps = (p_end - p_begin + BK - 1) / BK;

  const int c4 = (tid & 15) * 4;
  const int prow = tid >> 4;  // 0..15, +16
  const bool a_col_ok = mt * WG_TILE + c4 < g.Ci;  // Ci, Co are multiples of 4 (ld alignment)
  const bool b_col_ok = nt * WG_TILE + c4 < g.Co;

  f32x4 ra[2], rb[2];
  auto gload = [&](int s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = p_begin + s * BK + prow + 16 * i;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (q < p_end) {
        const int n = q / ghw, rem = q - n * ghw;
        const int gi = rem / g.gw, gj = rem - gi * g.gw;
        const int iy = gi * g.istep + dy, ix = gj * g.istep + dx;
        if (a_col_ok && (unsigned)iy < (unsigned)g.Hi && (unsigned)ix < (unsigned)g.Wi)
          va = *reinterpret_cast<const f32x4*>(p.in + ((size_t)(n * g.Hi + iy) * g.Wi + ix) * g.ldi + mt * WG_TILE + c4);
        if (b_col_ok) {
          const int oy = g.oy0 + g.ostep * gi, ox = g.ox0 + g.ostep * gj;
          vb = *reinterpret_cast<const f32x4*>(p.gout + ((size_t)(n * g.Ho + oy) * g.Wo + ox) * g.ldo + nt * WG_TILE + c4);
        }
      }
      ra[i] = va;
      rb[i] = vb;
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4*>(sA + (prow + 16 * i) * WG_LDS_ROW + c4) = ra[i];
      *reinterpret_cast<f32x4*>(sB + (prow + 16 * i) * WG_LDS_ROW + c4) = rb[i];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lr = lane & 31, lh = lane >> 5;
  const float* fa = sA + lh * WG_LDS_ROW + wm * 32 + lr;
  const float* fb = sB + lh * WG_LDS_ROW + wn * 32 + lr;

  if (steps > 0) {
    gload(0);
    sstore();
  }
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) gload(s + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) acc = mfma32(fa[2 * kk * WG_LDS_ROW], fb[2 * kk * WG_LDS_ROW], acc);
    __syncthreads();
    if (s + 1 < steps) sstore();
    __syncthreads();
  }

  float* slab = p.slabs + (((size_t)split * p.Tslabs + g.ws[t]) * p.m_pad + mt * WG_TILE + wm * 32) * p.n_pad + nt * WG_TILE + wn * 32 + lr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    slab[(size_t)row * p.n_pad] = acc[r];
  }
}

#include "wg2_kernel.inc"
#include "wg2b_kernel.inc"
#include "wg2s_kernel.inc"
#include "wg3b_kernel.inc"
#include "wg4s_kernel.inc"
#include "wg5p_kernel.inc"
#include "wg3_kernel.inc"

// ------------------------------------------------------------------------------------ small kernels
__global__ void pack_weight_kernel(const float* __restrict__ w, int D0, int D1, int T, int rows_from_d0,
                                   float* __restrict__ dst, int rows_pad, int k_pad) {
  const size_t total = (size_t)T * rows_pad * k_pad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % k_pad);
    const int row = (int)((i / k_pad) % rows_pad);
    const int t = (int)(i / ((size_t)k_pad * rows_pad));
    const int rows = rows_from_d0 ? D0 : D1, K = rows_from_d0 ? D1 : D0;
    float v = 0.f;
    if (row < rows && k < K) {
      const int d0 = rows_from_d0 ? row : k, d1 = rows_from_d0 ? k : row;
      v = w[((size_t)d0 * D1 + d1) * T + t];
    }
    dst[i] = v;
  }
}

__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, int D0, int D1, int T, int rows_from_d0,
                                        __bf16* __restrict__ dst, int rows_pad, int k_pad) {
  const size_t total = (size_t)T * rows_pad * k_pad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % k_pad);
    const int row = (int)((i / k_pad) % rows_pad);
    const int t = (int)(i / ((size_t)k_pad * rows_pad));
    const int rows = rows_from_d0 ? D0 : D1, K = rows_from_d0 ? D1 : D0;
    float v = 0.f;
    if (row < rows && k < K) {
      const int d0 = rows_from_d0 ? row : k, d1 = rows_from_d0 ? k : row;
      v = w[((size_t)d0 * D1 + d1) * T + t];
    }
    dst[i] = (__bf16)v;  // round to nearest even
  }
}

// Split panels for gg3s_kernel: dst[t][row][k chunk][plane][32 k], plane q of w = q-th term of the exact bf16 expansion, or (TE =
// _Float16) of the fp16 expansion of w * 2^k with k from *amax = max|w| (split_scale, gg3s_kernel.inc).
template <int NP, class TE>
__global__ void pack_weight_split_kernel(const float* __restrict__ w, int D0, int D1, int T, int rows_from_d0,
                                         TE* __restrict__ dst, int rows_pad, int k_pad, const float* __restrict__ amax) {
  const size_t total = (size_t)T * rows_pad * k_pad;
  const int kch = k_pad / 32;
  const float sc = std::is_same<TE, _Float16>::value ? split_scale(*amax) : 1.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % k_pad);
    const int row = (int)((i / k_pad) % rows_pad);
    const int t = (int)(i / ((size_t)k_pad * rows_pad));
    const int rows = rows_from_d0 ? D0 : D1, K = rows_from_d0 ? D1 : D0;
    float v = 0.f;
    if (row < rows && k < K) {
      const int d0 = rows_from_d0 ? row : k, d1 = rows_from_d0 ? k : row;
      v = w[((size_t)d0 * D1 + d1) * T + t] * sc;
    }
    TE* out = dst + ((((size_t)t * rows_pad + row) * kch + k / 32) * NP) * 32 + (k & 31);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const TE h = (TE)v;  // round to nearest even
      out[q * 32] = h;
      v -= (float)h;       // exact
    }
  }
}

// max |x| over an NHWC slice (pixels x C, row stride ld) as an unsigned compare of the magnitude bits (NaN > inf > finite, so a
// non-finite tensor yields a non-finite maximum); *out must be zero before the launch.
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long pixels, int C, int ld, unsigned* __restrict__ out) {
  unsigned m = 0;
  const bool vec = (C % 4 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const long long stride = (long long)gridDim.x * blockDim.x, first = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (vec) {
    const int C4 = C / 4;
    const long long total = pixels * C4;
    auto at = [&](long long i) {
      const long long q = i / C4;
      return *reinterpret_cast<const u32x4*>(x + q * ld + (int)(i - q * C4) * 4);
    };
    auto fold = [&](u32x4 v) { m = max(max(m, v[0] & 0x7fffffffu), max(v[1] & 0x7fffffffu, max(v[2] & 0x7fffffffu, v[3] & 0x7fffffffu))); };
    long long i = first;
    for (; i + 3 * stride < total; i += 4 * stride) {  // four loads in flight per thread: the pass is bound by memory latency otherwise
      const u32x4 a = at(i), b = at(i + stride), c = at(i + 2 * stride), d = at(i + 3 * stride);
      fold(a); fold(b); fold(c); fold(d);
    }
    for (; i < total; i += stride) fold(at(i));
  } else {
    const long long total = pixels * C;
    for (long long i = first; i < total; i += stride) {
      const long long q = i / C;
      m = max(m, __float_as_uint(x[q * ld + (i - q * C)]) & 0x7fffffffu);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  __shared__ unsigned red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(red[0], red[1]), max(red[2], red[3]));
    // one atomic per workgroup, and only when it can raise the running maximum (the unsynchronised read is a filter, not the result)
    if (m > *reinterpret_cast<volatile unsigned*>(out)) atomicMax(out, m);
  }
}

__global__ void zero_word_kernel(unsigned* p) { *p = 0u; }

// *out = max(*out, max|x|): `zero_first` for a fresh measurement into memory of unknown content
static int launch_absmax(const float* x, long long pixels, int C, int ld, float* out, hipStream_t st, bool zero_first) {
  if (zero_first) hipLaunchKernelGGL(zero_word_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned*>(out));
  const long long items = pixels * (C % 4 == 0 ? C / 4 : C);
  const int blocks = (int)std::max<long long>(1, std::min<long long>((items + 255) / 256 / 4, 1024));
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, st, x, pixels, C, ld, reinterpret_cast<unsigned*>(out));
  return check_launch("absmax");
}

// ------------------------------------------------------------------ batched weight packing (fp16-split mode)
// After an optimiser step every conv weight of the model needs its panels again: per weight that is a fill of the max|w| word, the
// max|w| pass and the pack — three small launches, ~200 per train step.  lhg_pack_weights does the same work for up to PACK_BATCH
// weights in three launches: the descriptors travel in the kernel arguments, `blk0` / `ablk0` are running block counts and a
// workgroup finds its weight with a (wave-uniform) scan over them.
constexpr int PACK_BATCH = 64;
struct PackItem {
  const float* w;
  _Float16* dst;
  unsigned* amax;
  const unsigned* amax_src;  // where the pack reads max|w|: `amax`, or the slot of an earlier item of the batch with the SAME weight (its other
                             // panel form): that weight is measured once, and the first workgroup of this item copies the value into `amax`
  int D0, D1, T, rows_from_d0, rows_pad, k_pad;
  unsigned blk0, ablk0;  // first workgroup of this weight in the pack / max|w| launches (a duplicate owns no max|w| workgroups)
};
struct PackBatch {
  int n;
  unsigned blocks, ablocks;
  PackItem it[PACK_BATCH];
};

__global__ void pack_batch_zero_kernel(const PackBatch b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b.n) *b.it[i].amax = 0u;
}

__device__ __forceinline__ int pack_batch_find(const PackBatch& b, unsigned blk, bool amax_pass) {
  int i = 0;
  while (i + 1 < b.n && (amax_pass ? b.it[i + 1].ablk0 : b.it[i + 1].blk0) <= blk) ++i;
  return i;
}

__global__ __launch_bounds__(256) void pack_batch_absmax_kernel(const PackBatch b) {
  const int i = pack_batch_find(b, blockIdx.x, true);
  const PackItem& it = b.it[i];
  const unsigned nblk = (i + 1 < b.n ? b.it[i + 1].ablk0 : b.ablocks) - it.ablk0;
  const size_t total = (size_t)it.D0 * it.D1 * it.T;
  unsigned m = 0;
  const unsigned* w = reinterpret_cast<const unsigned*>(it.w);
  const size_t stride = (size_t)nblk * 256;
  size_t e = (size_t)(blockIdx.x - it.ablk0) * 256 + threadIdx.x;
  for (; e + 3 * stride < total; e += 4 * stride) {  // four loads in flight
    const unsigned a = w[e], b2 = w[e + stride], c = w[e + 2 * stride], d = w[e + 3 * stride];
    m = max(max(m, a & 0x7fffffffu), max(b2 & 0x7fffffffu, max(c & 0x7fffffffu, d & 0x7fffffffu)));
  }
  for (; e < total; e += stride) m = max(m, w[e] & 0x7fffffffu);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  __shared__ unsigned red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(red[0], red[1]), max(red[2], red[3]));
    if (m > *reinterpret_cast<volatile unsigned*>(it.amax)) atomicMax(it.amax, m);
  }
}

__global__ __launch_bounds__(256) void pack_batch_kernel(const PackBatch b) {  // pack_weight_split_kernel<2, _Float16> per weight
  const int i = pack_batch_find(b, blockIdx.x, false);
  const PackItem& it = b.it[i];
  const unsigned nblk = (i + 1 < b.n ? b.it[i + 1].blk0 : b.blocks) - it.blk0;
  const size_t total = (size_t)it.T * it.rows_pad * it.k_pad;
  const int kch = it.k_pad / 32;
  const float sc = split_scale(__uint_as_float(*it.amax_src));
  if (it.amax_src != it.amax && blockIdx.x == it.blk0 && threadIdx.x == 0) *it.amax = *it.amax_src;
  const int rows = it.rows_from_d0 ? it.D0 : it.D1, K = it.rows_from_d0 ? it.D1 : it.D0;
  for (size_t e = (size_t)(blockIdx.x - it.blk0) * 256 + threadIdx.x; e < total; e += (size_t)nblk * 256) {
    const int k = (int)(e % it.k_pad);
    const int row = (int)((e / it.k_pad) % it.rows_pad);
    const int t = (int)(e / ((size_t)it.k_pad * it.rows_pad));
    float v = 0.f;
    if (row < rows && k < K) {
      const int d0 = it.rows_from_d0 ? row : k, d1 = it.rows_from_d0 ? k : row;
      v = it.w[((size_t)d0 * it.D1 + d1) * it.T + t] * sc;
    }
    _Float16* out = it.dst + ((((size_t)t * it.rows_pad + row) * kch + k / 32) * 2) * 32 + (k & 31);
    const _Float16 h0 = (_Float16)v;
    out[0] = h0;
    out[32] = (_Float16)(v - (float)h0);
  }
}

// bf16 operand mode: pack_weight_bf16_kernel per weight of the batch (dst[t][row][k] bf16, round to nearest even), one launch for up to
// PACK_BATCH weights — the bf16 configurations re-packed every conv weight with a launch of its own after each optimiser step (91 per
// train step, 0.7 ms of kernel time and as many host dispatches in a host-bound step).
__global__ __launch_bounds__(256) void pack_batch_bf16_kernel(const PackBatch b) {
  const int i = pack_batch_find(b, blockIdx.x, false);
  const PackItem& it = b.it[i];
  const unsigned nblk = (i + 1 < b.n ? b.it[i + 1].blk0 : b.blocks) - it.blk0;
  const size_t total = (size_t)it.T * it.rows_pad * it.k_pad;
  const int rows = it.rows_from_d0 ? it.D0 : it.D1, K = it.rows_from_d0 ? it.D1 : it.D0;
  __bf16* dst = reinterpret_cast<__bf16*>(it.dst);
  for (size_t e = (size_t)(blockIdx.x - it.blk0) * 256 + threadIdx.x; e < total; e += (size_t)nblk * 256) {
    const int k = (int)(e % it.k_pad);
    const int row = (int)((e / it.k_pad) % it.rows_pad);
    const int t = (int)(e / ((size_t)it.k_pad * it.rows_pad));
    float v = 0.f;
    if (row < rows && k < K) {
      const int d0 = it.rows_from_d0 ? row : k, d1 = it.rows_from_d0 ? k : row;
      v = it.w[((size_t)d0 * it.D1 + d1) * it.T + t];
    }
    dst[e] = (__bf16)v;
  }
}

// The same panels from 32 x 32 (row, k) tiles staged through LDS: the source w[d0][d1][t] is contiguous over (d1, t), so a tile is 32
// runs of 32 * T floats — every byte of every line used once — where pack_batch_kernel above reads with a stride of T floats (row-major
// panels) or D1 * T floats (input-gradient panels), once per tap: 9 passes over a weight that does not fit an XCD's L2.  The stores are
// 64-byte runs per (tap, row, plane).  Same values as pack_batch_kernel (one multiply, two roundings per element).
constexpr int PACK_TILE_TMAX = 9;
__global__ __launch_bounds__(256) void pack_batch_tiled_kernel(const PackBatch b) {
  __shared__ float tile[32][32 * PACK_TILE_TMAX + 1];
  const int i = pack_batch_find(b, blockIdx.x, false);
  const PackItem& it = b.it[i];
  const unsigned nblk = (i + 1 < b.n ? b.it[i + 1].blk0 : b.blocks) - it.blk0;
  const int T = it.T, kt = it.k_pad / 32, rt = it.rows_pad / 32, kch = it.k_pad / 32;
  const float sc = split_scale(__uint_as_float(*it.amax_src));
  if (it.amax_src != it.amax && blockIdx.x == it.blk0 && threadIdx.x == 0) *it.amax = *it.amax_src;
  const int run = 32 * T;  // floats per contiguous source run
  for (unsigned tl = blockIdx.x - it.blk0; tl < (unsigned)(kt * rt); tl += nblk) {
    const int row0 = (int)(tl / kt) * 32, k0 = (int)(tl % kt) * 32;
    // segment g: row-major panels: row row0 + g, its k0 .. k0 + 31 (d1) and all taps; input-gradient panels: k = k0 + g, rows row0 .. + 31 (d1)
    const int d1_0 = it.rows_from_d0 ? k0 : row0;
    if (((it.D1 * T) & 3) == 0 && d1_0 + 32 <= it.D1 && (reinterpret_cast<uintptr_t>(it.w) & 15) == 0) {
      // whole runs inside the weight, 16-byte aligned (D1 T and 32 T floats are multiples of four): 16-byte loads, all of a thread's
      // loads requested before the first is used (with one 4-byte load per iteration the pass ran at a fifth of the memory rate)
      const int run4 = run / 4;                    // 8 T
      constexpr int PL = (32 * 8 * PACK_TILE_TMAX + 255) / 256;  // <= 9 loads per thread
      f32x4 q[PL];
#pragma unroll
      for (int u = 0; u < PL; ++u) {
        const int e = threadIdx.x + 256 * u;
        const int g = e / run4, o4 = e - g * run4;
        const int d0 = it.rows_from_d0 ? row0 + g : k0 + g;
        const bool ok = e < 32 * run4 && d0 < it.D0;
        q[u] = ok ? *reinterpret_cast<const f32x4*>(it.w + ((size_t)d0 * it.D1 + d1_0) * T + 4 * o4) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < PL; ++u) {
        const int e = threadIdx.x + 256 * u;
        if (e < 32 * run4) {
          const int g = e / run4, o4 = e - g * run4;
#pragma unroll
          for (int c = 0; c < 4; ++c) tile[g][4 * o4 + c] = q[u][c] * sc;
        }
      }
    } else {
      for (int e = threadIdx.x; e < 32 * run; e += 256) {
        const int g = e / run, o = e - g * run;
        const int d0 = it.rows_from_d0 ? row0 + g : k0 + g;
        const int d1 = d1_0 + o / T;
        tile[g][o] = (d0 < it.D0 && d1 < it.D1) ? it.w[((size_t)d0 * it.D1 + d1_0) * T + o] * sc : 0.f;
      }
    }
    __syncthreads();
    // outputs: (t, row, k pair): 16 lanes cover the 32 k of one (t, row): two 64-byte runs (plane 0 / plane 1)
    for (int e = threadIdx.x; e < T * 32 * 16; e += 256) {
      const int kp = e & 15, r = (e >> 4) & 31, t = e >> 9;
      float v[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int kl = 2 * kp + q;
        v[q] = it.rows_from_d0 ? tile[r][kl * T + t] : tile[kl][r * T + t];
      }
      const _Float16 h00 = (_Float16)v[0], h01 = (_Float16)v[1];
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      _Float16* out = it.dst + ((((size_t)t * it.rows_pad + row0 + r) * kch + k0 / 32) * 2) * 32 + 2 * kp;
      *reinterpret_cast<f16x2*>(out) = f16x2{h00, h01};
      *reinterpret_cast<f16x2*>(out + 32) = f16x2{(_Float16)(v[0] - (float)h00), (_Float16)(v[1] - (float)h01)};
    }
    __syncthreads();
  }
}

// grad[d0][d1][t] = sum_s slabs[s][t][m][n].  Block = 64 consecutive outputs (n fastest: coalesced slab reads) x 4 slab
// slices; each thread sums every 4th slab, then the slices are combined through LDS.
template <int LANES>  // outputs per block; 256 / LANES slab slices
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int S, int T, int m_pad, int n_pad,
                                                           float* __restrict__ grad, int D0, int D1, int m_is_d1, int accumulate) {
  constexpr int SLICES = 256 / LANES;
  __shared__ float red[SLICES][LANES];
  const int M = m_is_d1 ? D1 : D0, Nn = m_is_d1 ? D0 : D1;
  const size_t total = (size_t)T * M * Nn;
  const size_t slab_stride = (size_t)T * m_pad * n_pad;
  const int lane_o = threadIdx.x % LANES, slice = threadIdx.x / LANES;
  for (size_t base = (size_t)blockIdx.x * LANES; base < total; base += (size_t)gridDim.x * LANES) {
    const size_t i = base + lane_o;
    float acc = 0.f;
    size_t dst = 0;
    if (i < total) {
      const int n = (int)(i % Nn);
      const int m = (int)((i / Nn) % M);
      const int t = (int)(i / ((size_t)Nn * M));
      const float* src = slabs + ((size_t)t * m_pad + m) * n_pad + n;
      for (int s = slice; s < S; s += SLICES) acc += src[s * slab_stride];
      const int d0 = m_is_d1 ? n : m, d1 = m_is_d1 ? m : n;
      dst = ((size_t)d0 * D1 + d1) * T + t;
    }
    red[slice][lane_o] = acc;
    __syncthreads();
    if (slice == 0 && i < total) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < SLICES; ++k) v += red[k][lane_o];
      grad[dst] = accumulate ? grad[dst] + v : v;  // one writer per element: accumulating into a gradient slot is ordered by the stream
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ host side
// Optional per-launch timing of the two GEMM kernels with HIP events recorded on the launch stream
// (bench.py's roofline leg).  Off by default; never enabled while a graph is being captured.
struct KernelTimer {
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  size_t used = 0;
  double executed_flops = 0;
  std::vector<std::string> tags;  // per launch "geometry;variant;flops" (only filled while LHG_PROFILE_LOG names a file)
  std::pair<hipEvent_t, hipEvent_t>& next() {
    if (used == pool.size()) {
      hipEvent_t a, b;
      (void)hipEventCreate(&a);
      (void)hipEventCreate(&b);
      pool.emplace_back(a, b);
    }
    return pool[used++];
  }
};
static KernelTimer g_timer[2];  // 0: gg_kernel, 1: wg_kernel
// LHG_AUTOTUNE=0: every launcher falls back to its fixed heuristic (until round 5 the split-mode and bf16 launchers ignored the variable).
// With the BatchNorm statistics folded from the epilogues' partial rows the tiling choice shows in the statistics' last bits: a fixed
// choice (this switch, or one LHG_TUNE_CACHE file shared by the processes) is what makes two PROCESSES agree bit for bit.
static bool g_autotune_enabled = [] { const char* e = getenv("LHG_AUTOTUNE"); return e ? atoi(e) != 0 : true; }();
static std::map<std::array<int, 12>, int> g_gg_choice, g_wg_choice;

// One-time timing of the valid kernel variants for a geometry not seen before: the device is drained first (the weight-gradient
// stream may be busy), then every variant runs once untimed and TUNE_RUNS times between two HIP events on `st`.  The only place the
// library synchronises; skipped while the stream is being captured.  Returns the cached / measured winner or -1.
// LHG_TUNE_CACHE=<file>: choices are appended to that file and read back by later processes (one line per geometry, tagged with
// TUNE_SCHEMA so that a build with a different variant numbering ignores stale lines).
constexpr int TUNE_RUNS = 5;
constexpr const char* TUNE_SCHEMA = "lhg-tune-8";
static void tune_cache_load() {
  static bool done = false;
  if (done) return;
  done = true;
  const char* path = getenv("LHG_TUNE_CACHE");
  if (!path) return;
  FILE* f = fopen(path, "r");
  if (!f) return;
  char tag[32], which[8];
  while (fscanf(f, "%31s %7s", tag, which) == 2) {
    std::array<int, 12> key;
    int choice = -1;
    bool ok = true;
    for (int i = 0; i < 12; ++i) ok = ok && fscanf(f, "%d", &key[i]) == 1;
    ok = ok && fscanf(f, "%d", &choice) == 1;
    if (!ok) break;
    if (std::string(tag) != TUNE_SCHEMA) continue;
    (which[0] == 'g' ? g_gg_choice : g_wg_choice)[key] = choice;
  }
  fclose(f);
}
static void tune_cache_append(bool gg, const std::array<int, 12>& key, int choice) {
  const char* path = getenv("LHG_TUNE_CACHE");
  if (!path) return;
  if (FILE* f = fopen(path, "a")) {
    fprintf(f, "%s %s", TUNE_SCHEMA, gg ? "gg" : "wg");
    for (int v : key) fprintf(f, " %d", v);
    fprintf(f, " %d\n", choice);
    fclose(f);
  }
}

template <class Valid, class Run>
static int autotuned_variant(std::map<std::array<int, 12>, int>& cache, const std::array<int, 12>& key, int variants, Valid valid, Run run,
                             hipStream_t st) {
  tune_cache_load();
  auto it = cache.find(key);
  if (it != cache.end()) {
    if (it->second >= 0 && it->second < variants && valid(it->second)) return it->second;
    cache.erase(it);  // a persisted choice this build cannot run
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return -1;
  static const bool idle = [] { const char* e = getenv("LHG_TUNE_IDLE"); return e ? atoi(e) != 0 : true; }();
  static const int runs = [] { const char* e = getenv("LHG_TUNE_RUNS"); return e ? std::max(1, atoi(e)) : TUNE_RUNS; }();
  if (idle && hipDeviceSynchronize() != hipSuccess) return -1;  // idle device: no other stream shares the CUs with the timed launches
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  int choice = -1;
  for (int v = 0; v < variants; ++v) {
    if (!valid(v)) continue;
    run(v);  // warm-up (code object load, caches)
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < runs; ++r) run(v);
    (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess || hipGetLastError() != hipSuccess) continue;
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) { best = ms; choice = v; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (choice >= 0) {
    cache[key] = choice;
    tune_cache_append(&cache == &g_gg_choice, key, choice);
  }
  return choice;
}

struct ScopedKernelTime {
  KernelTimer& t;
  hipStream_t st;
  std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
  ScopedKernelTime(int which, hipStream_t s, double flops) : t(g_timer[which]), st(s) {
    if (t.on) {
      ev = &t.next();
      t.executed_flops += flops;
      (void)hipEventRecord(ev->first, st);
    }
  }
  ~ScopedKernelTime() {
    if (ev) (void)hipEventRecord(ev->second, st);
  }
  void tag(const Geom& g, int rows, int cols, int variant, double flops) {
    static const bool log = getenv("LHG_PROFILE_LOG") != nullptr;
    if (!ev || !log) return;
    char b[160];
    snprintf(b, sizeof b, "%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%.0f", g.M, rows, cols, g.Ci, g.Co, g.T, g.istep, g.ostep, g.Hi, variant, flops);
    t.tags.resize(t.used);
    t.tags[t.used - 1] = b;
  }
};

// LHG_XCD=0 keeps the hardware's round-robin tile order (for A/B measurements of the L2 effect)
static int xcd_order() {
  static const int on = [] { const char* e = getenv("LHG_XCD"); return e ? atoi(e) : 1; }();  // 2: merged-class launches in the round-3 order (gg3s_kernel)
  return on;
}

// process-wide arithmetic of the conv GEMMs (lhg_set_conv_precision).  Default: the fp32-faithful two-term fp16 split kernels; the environment
// variable LHG_CONV_PRECISION = fp32 | fp32_split | fp32_split2 | fp32_split_f16 | bf16 overrides it for programs whose flags must stay the reference's.
static int default_precision() {
  static const int d = [] {
    const char* e = getenv("LHG_CONV_PRECISION");
    if (!e) return (int)LHG_PRECISION_F32_SPLIT_F16;
    const std::string v(e);
    if (v == "fp32" || v == "f32") return (int)LHG_PRECISION_F32;
    if (v == "bf16") return (int)LHG_PRECISION_BF16;
    if (v == "fp32_split2") return (int)LHG_PRECISION_F32_SPLIT2;
    if (v == "fp32_split") return (int)LHG_PRECISION_F32_SPLIT;
    return (int)LHG_PRECISION_F32_SPLIT_F16;
  }();
  return d;
}
static int g_precision = default_precision();


// gg_epilogue.inc stores through a descriptor based at the tile's first output pixel with 32-bit offsets below 2^30: bound the bytes a
// tile of up to 256 consecutive sub-grid points can span (whole image rows it touches, at the output's step), for either row pitch.
static bool epilogue_window_fits(const Geom& g, int ldres, int es) {
  const long long rows = (long long)g.ostep * (256 / std::max(1, g.gw) + 2) + 1;
  const long long bytes = rows * g.Wo * (long long)std::max(g.ldo, ldres) * es;
  return bytes < (1ll << 30);
}

static int launch_gg_bf16(GGParams& p, hipStream_t st);
static int launch_gg_split(GGParams& p, hipStream_t st);

static inline bool split_f16() { return g_precision == LHG_PRECISION_F32_SPLIT_F16; }
static inline bool split_mode() { return g_precision == LHG_PRECISION_F32_SPLIT || g_precision == LHG_PRECISION_F32_SPLIT2 || split_f16(); }
static inline int split_planes() { return g_precision == LHG_PRECISION_F32_SPLIT ? 3 : 2; }
static int launch_gg(GGParams& p, hipStream_t st);
// BatchNorm statistics from the epilogue (GGParams::stat_part): the rows the chosen tiling writes — one per consumer-wave row of every
// M tile — handed back to lhg_conv2d_forward_stats (the finish kernel needs the count; the buffer is sized by lhg_conv2d_stats_rows_bound)
static int g_stat_rows = 0;
static void note_stat_rows(const GGParams& p, long long m_total, int bm, int wgm) {
  if (p.stat_part) g_stat_rows = (int)((m_total + bm - 1) / bm) * wgm;
}
static bool merge_classes() {
  static const bool on = [] { const char* e = getenv("LHG_MERGE_CLASSES"); return !e || atoi(e) != 0; }();
  return on;
}
// The output-parity classes of one operator (same tensors and weights; sub-grid, tap set and pixel count differ): one merged launch
// in the fp16-split mode — the short K loops of the one- and two-tap classes leave the chip two thirds empty between the launches'
// last and first workgroups otherwise — one launch per class in the other modes.  Heaviest classes first.
static int launch_gg_classes(GGParams& p, Geom* cls, int n, hipStream_t st) {
  if (n <= 0) return LHG_OK;
  std::stable_sort(cls, cls + n, [](const Geom& a, const Geom& b) { return (long long)a.M * a.T > (long long)b.M * b.T; });
  if (n > 1 && n <= 4 && merge_classes() && ((g_precision == LHG_PRECISION_F32_SPLIT_F16 && !act_is_bf16()) || g_precision == LHG_PRECISION_BF16)) {
    p.g = cls[0];
    for (int c = 1; c < n; ++c) p.gc[c - 1] = cls[c];
    p.ncls = n;
    return launch_gg(p, st);
  }
  for (int c = 0; c < n; ++c) {
    p.g = cls[c];
    p.ncls = 0;
    p.ks_off = 1;
    const int rc = launch_gg(p, st);
    if (rc) return rc;
  }
  return LHG_OK;
}

// fp16-split mode: max|w| sits in the 16 bytes behind the panels of a packed weight (lhg_pack_weight put it there)
static inline const float* weight_amax(const float* wp, int taps, int rows_pad, int k_pad) {
  return split_f16() ? wp + (size_t)taps * rows_pad * k_pad : nullptr;
}

static int launch_gg(GGParams& p, hipStream_t st) {
  const Geom& g = p.g;
  if (g.M <= 0) return LHG_OK;
  p.xcd = xcd_order();
  // LHG_GG_PRIO: 0 / 1 / 2 only (s_setprio placement: same bits).  Rounds 2 - 4 ran timing ablations through values >= 10 that produced
  // WRONG results on purpose; the shipped kernels no longer contain them and such a value is refused loudly, here and in native.load().
  static const int prio = [] { const char* e = getenv("LHG_GG_PRIO"); return e ? atoi(e) : 1; }();
  LHG_REQUIRE(prio >= 0 && prio <= 2, "LHG_GG_PRIO=%d: only 0, 1, 2 exist (the timing ablations of earlier rounds gave wrong results and were removed)", prio);
  p.prio = prio;
  if (g_precision == LHG_PRECISION_BF16) return launch_gg_bf16(p, st);
  LHG_REQUIRE(!act_is_bf16(), "bf16 activation storage needs the bf16 conv precision (lhg_set_conv_precision(LHG_PRECISION_BF16))");
  if (split_mode()) return launch_gg_split(p, st);
  LHG_REQUIRE(g.Ci % BK == 0 && g.Ci > 0, "gather-GEMM: K channels (%d) must be a positive multiple of 32", g.Ci);
  LHG_REQUIRE(g.ldi % 4 == 0 && (reinterpret_cast<uintptr_t>(p.in) & 15) == 0, "gather-GEMM: input must be 16-byte aligned (ld %d)", g.ldi);
  LHG_REQUIRE((reinterpret_cast<uintptr_t>(p.wp) & 15) == 0, "gather-GEMM: packed weights must be 16-byte aligned");
  LHG_REQUIRE(p.rows_pad % 64 == 0 && p.rows_pad >= g.Co, "gather-GEMM: rows_pad %d must be a multiple of 64 covering Co=%d", p.rows_pad, g.Co);
  LHG_REQUIRE((long long)g.N * g.Ho * g.Wo < (1ll << 31) && (long long)g.N * g.Hi * g.Wi < (1ll << 31), "gather-GEMM: more than 2^31 pixels");
  LHG_REQUIRE(p.planar_out || epilogue_window_fits(g, p.res ? p.ldres : 0, 4), "gather-GEMM: an output tile spans 1 GiB or more (Wo %d, ld %d)", g.Wo, g.ldo);
  // extents addressed through 32-bit buffer descriptors by the pipelined kernels
  int max_ws = 0;
  for (int t = 0; t < g.T; ++t) max_ws = std::max(max_ws, g.ws[t]);
  const unsigned long long in_bytes = (((unsigned long long)g.N * g.Hi * g.Wi - 1) * g.ldi + g.Ci) * 4ull;
  const unsigned long long wp_bytes = (unsigned long long)(max_ws + 1) * p.rows_pad * g.Ci * 4ull;
  const bool small = wp_bytes < (1ull << 32) - 64;  // the pipelined kernels window the gathered tensor per workgroup: any size
  const unsigned long long ib = in_bytes;
  const unsigned wb = (unsigned)wp_bytes;
  auto blocks = [&](int bm, int bn) { return (unsigned)(((g.M + bm - 1) / bm) * (p.rows_pad / bn)); };
  const bool n128 = p.rows_pad % 128 == 0;

  // tiling variants; every variant produces the same values (the K order is identical)
  constexpr int NV = 6;
  if (!small) p.stat_part = nullptr;  // gg_kernel has its own epilogue: no statistics rows (g_stat_rows stays 0, the caller runs the pass)
  auto valid = [&](int v) { return (v < 3 ? small : p.stat_part == nullptr) && ((v == 0 || v == 3) ? n128 : true); };
  auto run = [&](int v) {
    switch (v) {
      case 0: hipLaunchKernelGGL((gg2_kernel<128, 128, 2, 2>), dim3(blocks(128, 128)), dim3(256), 0, st, p, ib, wb); break;
      case 1: hipLaunchKernelGGL((gg2_kernel<128, 64, 2, 2>), dim3(blocks(128, 64)), dim3(256), 0, st, p, ib, wb); break;
      case 2: hipLaunchKernelGGL((gg2_kernel<64, 64, 2, 2>), dim3(blocks(64, 64)), dim3(256), 0, st, p, ib, wb); break;
      case 3: hipLaunchKernelGGL((gg_kernel<128, 128, 2, 2>), dim3(blocks(128, 128)), dim3(256), 0, st, p); break;
      case 4: hipLaunchKernelGGL((gg_kernel<256, 64, 4, 1>), dim3(blocks(256, 64)), dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL((gg_kernel<64, 64, 2, 2>), dim3(blocks(64, 64)), dim3(256), 0, st, p); break;
    }
  };
  auto heuristic = [&]() {
    if (small) return n128 && blocks(128, 128) >= 400 ? 0 : (blocks(128, 64) >= 400 ? 1 : 2);
    return n128 && blocks(128, 128) >= 400 ? 3 : (blocks(256, 64) >= 400 ? 4 : 5);
  };

  static const int forced = [] { const char* e = getenv("LHG_GG_VARIANT"); return e ? atoi(e) : -1; }();
  static const bool tune = [] { const char* e = getenv("LHG_AUTOTUNE"); return e ? atoi(e) != 0 : true; }();
  int choice = -1;
  if (forced >= 0 && forced < NV && valid(forced)) choice = forced;
  if (choice < 0 && tune && g_autotune_enabled) {
    const std::array<int, 12> key = {g.M, p.rows_pad, g.Ci, g.Co, g.T, g.istep, g.ostep, g.Hi, g.Wi, g.ldi, g.ldo, p.planar_out};
    choice = autotuned_variant(g_gg_choice, key, NV, valid, run, st);
  }
  if (choice < 0) choice = heuristic();
  if (choice < 3) note_stat_rows(p, g.M, choice == 2 ? 64 : 128, 2);
  ScopedKernelTime timed(0, st, 2.0 * g.M * (double)p.rows_pad * g.Ci * g.T);
  timed.tag(g, p.rows_pad, g.Ci, choice, 2.0 * g.M * (double)p.rows_pad * g.Ci * g.T);
  run(choice);
  return check_launch("gg_kernel");
}

// bf16 operands: `p.wp` holds bf16 panels (lhg_pack_weight in the same mode).  Tilings 128x128 / 128x64 / 64x64, autotuned.
// In the bf16 storage mode (lhg_set_activation_dtype) the gathered tensor, the residual and the NHWC output are bf16 in HBM.
template <class TA>
static int launch_gg_bf16_t(GGParams& p, hipStream_t st) {
  const Geom& g = p.g;
  constexpr int ES = (int)sizeof(TA);
  LHG_REQUIRE(g.Ci % BK == 0 && g.Ci > 0, "gather-GEMM: K channels (%d) must be a positive multiple of 32", g.Ci);
  LHG_REQUIRE((g.ldi * ES) % 16 == 0 && (reinterpret_cast<uintptr_t>(p.in) & 15) == 0, "gather-GEMM: input must be 16-byte aligned (ld %d)", g.ldi);
  LHG_REQUIRE((reinterpret_cast<uintptr_t>(p.wp) & 15) == 0, "gather-GEMM: packed weights must be 16-byte aligned");
  LHG_REQUIRE(p.rows_pad % 64 == 0 && p.rows_pad >= g.Co, "gather-GEMM: rows_pad %d must be a multiple of 64 covering Co=%d", p.rows_pad, g.Co);
  // merged output-parity classes (round 5, second session: the bf16 modes launched every class of a stride-2 input gradient / transposed
  // conv by itself — 6 extra launches per such operator in a host-bound step): gg3s_kernel's class logic does not depend on the planes
  const int ncls = std::max(1, p.ncls);
  auto cls_geom = [&](int c) -> const Geom& { return c == 0 ? p.g : p.gc[c - 1]; };
  int max_ws = 0;
  for (int c = 0; c < ncls; ++c)
    for (int t = 0; t < cls_geom(c).T; ++t) max_ws = std::max(max_ws, cls_geom(c).ws[t]);
  const unsigned long long in_bytes = (((unsigned long long)g.N * g.Hi * g.Wi - 1) * g.ldi + g.Ci) * (unsigned long long)ES;
  const unsigned long long wp_bytes = (unsigned long long)(max_ws + 1) * p.rows_pad * g.Ci * 2ull;
  LHG_REQUIRE(wp_bytes < (1ull << 32) - 64, "gather-GEMM (bf16 mode): weight panels of 4 GiB and more are not supported");
  LHG_REQUIRE(p.planar_out || epilogue_window_fits(g, p.res ? p.ldres : 0, ES), "gather-GEMM: an output tile spans 1 GiB or more (Wo %d, ld %d)", g.Wo, g.ldo);
  const unsigned long long ib = in_bytes;
  const unsigned wb = (unsigned)wp_bytes;
  auto blocks = [&](int bm, int bn) {  // also the classes' first workgroups (written into p before it is copied to the kernel)
    unsigned total = 0;
    for (int c = 0; c < ncls; ++c) {
      p.cblk[c] = total;
      total += (unsigned)(((cls_geom(c).M + bm - 1) / bm) * (p.rows_pad / bn));
    }
    p.cblk[ncls] = total;
    return total;
  };
  const bool n128 = p.rows_pad % 128 == 0;
  // 0..2: 32-k steps; 3..5: 64-k steps (half the barriers and address arithmetic per flop; Ci % 64 == 0); 6..9: the 8-wave
  // producer / consumer kernel (gg3s_kernel with one bf16 plane): 128x128 / 128x64 / 64x64 tiles with 64-k steps, 128x128 with 32-k
  constexpr int NV = 10;
  const bool k64 = g.Ci % 64 == 0;
  auto valid = [&](int v) {
    if (v >= 6) return (v == 6 || v == 9 ? n128 : true) && (v == 9 || k64);
    if (ncls > 1) return false;  // gg2b_kernel knows one geometry
    return (v % 3 == 0 ? n128 : true) && (v < 3 || k64);
  };
  auto run = [&](int v) {
    // (the grid first: blocks() writes the class offsets into p, which the launch then copies)
    const unsigned nb = v == 6 || v == 9 || v == 0 || v == 3 ? blocks(128, 128) : (v == 7 || v == 1 || v == 4 ? blocks(128, 64) : blocks(64, 64));
    (void)nb;
    switch (v) {
      case 6: hipLaunchKernelGGL((gg3s_kernel<128, 128, 1, 4, 64, TA>), dim3(blocks(128, 128)), dim3(512), 0, st, p, ib, wb); break;
      case 7: hipLaunchKernelGGL((gg3s_kernel<128, 64, 1, 4, 64, TA>), dim3(blocks(128, 64)), dim3(512), 0, st, p, ib, wb); break;
      case 8: hipLaunchKernelGGL((gg3s_kernel<64, 64, 1, 8, 64, TA>), dim3(blocks(64, 64)), dim3(512), 0, st, p, ib, wb); break;
      case 9: hipLaunchKernelGGL((gg3s_kernel<128, 128, 1, 4, 32, TA>), dim3(blocks(128, 128)), dim3(512), 0, st, p, ib, wb); break;
      case 0: hipLaunchKernelGGL((gg2b_kernel<128, 128, 2, 2, 32, TA>), dim3(blocks(128, 128)), dim3(256), 0, st, p, ib, wb); break;
      case 1: hipLaunchKernelGGL((gg2b_kernel<128, 64, 2, 2, 32, TA>), dim3(blocks(128, 64)), dim3(256), 0, st, p, ib, wb); break;
      case 2: hipLaunchKernelGGL((gg2b_kernel<64, 64, 2, 2, 32, TA>), dim3(blocks(64, 64)), dim3(256), 0, st, p, ib, wb); break;
      case 3: hipLaunchKernelGGL((gg2b_kernel<128, 128, 2, 2, 64, TA>), dim3(blocks(128, 128)), dim3(256), 0, st, p, ib, wb); break;
      case 4: hipLaunchKernelGGL((gg2b_kernel<128, 64, 2, 2, 64, TA>), dim3(blocks(128, 64)), dim3(256), 0, st, p, ib, wb); break;
      case 5: hipLaunchKernelGGL((gg2b_kernel<64, 64, 2, 2, 64, TA>), dim3(blocks(64, 64)), dim3(256), 0, st, p, ib, wb); break;
      default: break;
    }
  };
  static const int forced = [] { const char* e = getenv("LHG_GGB_VARIANT"); return e ? atoi(e) : -1; }();
  int choice = (forced >= 0 && forced < NV && valid(forced)) ? forced : -1;
  if (choice < 0 && g_autotune_enabled) {
    const std::array<int, 12> key = {g.M, p.rows_pad, g.Ci, g.Co, g.T, g.istep, g.ostep, g.Hi, g.Wi, g.ldi, g.ldo, p.planar_out + 2 + 16 * ES + 256 * ncls};
    choice = autotuned_variant(g_gg_choice, key, NV, valid, run, st);
  }
  if (choice < 0) {
    if (ncls > 1) choice = k64 ? (n128 && blocks(128, 128) >= 256 ? 6 : (blocks(128, 64) >= 256 ? 7 : 8)) : 9;
    else choice = n128 && blocks(128, 128) >= 256 ? 0 : (blocks(128, 64) >= 256 ? 1 : 2);
  }
  LHG_REQUIRE(valid(choice), "gather-GEMM (bf16 mode): no kernel for this launch (%d parity classes, rows_pad %d, Ci %d)", ncls, p.rows_pad, g.Ci);
  if (ncls == 1) note_stat_rows(p, g.M, (choice == 2 || choice == 5 || choice == 8) ? 64 : 128, 2);  // every variant: 2 x 2 consumer waves
  double mt_sum = 0;  // sum over the classes of pixels x taps
  for (int c = 0; c < ncls; ++c) mt_sum += (double)cls_geom(c).M * cls_geom(c).T;
  ScopedKernelTime timed(0, st, 2.0 * mt_sum * (double)p.rows_pad * g.Ci);
  if (ncls == 1) timed.tag(g, p.rows_pad, g.Ci, choice, 2.0 * mt_sum * (double)p.rows_pad * g.Ci);
  else {
    Geom all = g;  // logged as one launch: total pixels, total taps
    all.M = 0; all.T = 0;
    for (int c = 0; c < ncls; ++c) { all.M += cls_geom(c).M; all.T += cls_geom(c).T; }
    timed.tag(all, p.rows_pad, g.Ci, choice, 2.0 * mt_sum * (double)p.rows_pad * g.Ci);
  }
  run(choice);
  return check_launch("gg2b_kernel");
}

static int launch_gg_bf16(GGParams& p, hipStream_t st) {
  return act_is_bf16() ? launch_gg_bf16_t<__bf16>(p, st) : launch_gg_bf16_t<float>(p, st);
}

// The split-K rule: at most 160 tiles of 128 x 128 and at least 64 K steps (the UNet's 24^2 x 1024-channel bottleneck: 144 tiles, 288
// steps).  The number of ranges follows from the GEOMETRY alone (about 480 workgroups, at most four ranges of whole 32-channel chunks),
// so that every tiling variant adds the same ranges in the same order.  LHG_SPLITK=0: never.  Returns the ranges (1: no split).
static int splitk_plan(long long M, int rows_pad, int K, int taps, int* chunks) {
  static const bool on = [] { const char* e = getenv("LHG_SPLITK"); return !e || atoi(e) != 0; }();
  // (measurement switches: the rule's three constants)
  static const int max_tiles = [] { const char* e = getenv("LHG_SPLITK_TILES"); return e ? atoi(e) : 160; }();
  static const int target = [] { const char* e = getenv("LHG_SPLITK_TARGET"); return e ? atoi(e) : 480; }();
  static const int max_ranges = [] { const char* e = getenv("LHG_SPLITK_MAX"); return e ? atoi(e) : 4; }();
  const long long tiles128 = ((M + 127) / 128) * ((rows_pad + 127) / 128);
  const int kch = K / 32;
  if (!on || g_precision != LHG_PRECISION_F32_SPLIT_F16 || M <= 0 || tiles128 > max_tiles || (long long)taps * kch < 64 || kch < 2) return 1;
  const int want = (int)std::min<long long>(std::min<long long>(max_ranges, std::max<long long>(1, (target + tiles128 / 2) / tiles128)), kch);
  const int c = (kch + want - 1) / want;
  if (chunks) *chunks = c;
  return (kch + c - 1) / c;  // no empty range
}
static size_t splitk_floats(int ks, long long M, long long M_padded, int rows_pad) {
  return (size_t)ks * (size_t)((std::max(M, M_padded) + 255) / 256 * 256) * (size_t)rows_pad;
}
// the caller's workspace for the NEXT gather-GEMM launch (lhg_gather_gemm_workspace): consumed — used or not — by that launch
static float* g_next_ws = nullptr;
static size_t g_next_ws_floats = 0;

// Split-K slabs without a caller's workspace: one grow-only device buffer per stream (launches of one stream are ordered: a launch's finish kernel has read the slabs
// before the next launch's GEMM overwrites them).  hipMalloc / hipFree block, so the buffer starts at 64 MiB — the step's largest need
// is 28 MB — and growing it waits for the stream first; never inside a graph capture (the eager warm-up steps have sized it by then).
static float* splitk_slab(hipStream_t st, size_t floats) {
  static std::map<hipStream_t, std::pair<float*, size_t>> pool;
  auto& e = pool[st];
  if (e.second >= floats) return e.first;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return nullptr;
  if (e.first) {
    (void)hipStreamSynchronize(st);
    (void)hipFree(e.first);
    e = {nullptr, 0};
  }
  const size_t want = std::max(floats, (size_t)16 << 20);
  float* ptr = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&ptr), want * sizeof(float)) != hipSuccess) return nullptr;
  e = {ptr, want};
  return ptr;
}

// fp32 operands as exact sums of bf16 terms (gg3s_kernel): `p.wp` holds split panels (lhg_pack_weight in the same mode).
static int launch_gg_split(GGParams& p, hipStream_t st) {
  const Geom& g = p.g;
  const int NP = split_planes();
  LHG_REQUIRE(g.Ci % BK == 0 && g.Ci > 0, "gather-GEMM: K channels (%d) must be a positive multiple of 32", g.Ci);
  LHG_REQUIRE(g.ldi % 4 == 0 && (reinterpret_cast<uintptr_t>(p.in) & 15) == 0, "gather-GEMM: input must be 16-byte aligned (ld %d)", g.ldi);
  LHG_REQUIRE((reinterpret_cast<uintptr_t>(p.wp) & 15) == 0, "gather-GEMM: packed weights must be 16-byte aligned");
  LHG_REQUIRE(p.rows_pad % 64 == 0 && p.rows_pad >= g.Co, "gather-GEMM: rows_pad %d must be a multiple of 64 covering Co=%d", p.rows_pad, g.Co);
  LHG_REQUIRE((long long)g.N * g.Ho * g.Wo < (1ll << 31) && (long long)g.N * g.Hi * g.Wi < (1ll << 31), "gather-GEMM: more than 2^31 pixels");
  int max_ws = 0;
  const int ncls = std::max(1, p.ncls);
  auto cls_geom = [&](int c) -> const Geom& { return c == 0 ? p.g : p.gc[c - 1]; };
  for (int c = 0; c < ncls; ++c)
    for (int t = 0; t < cls_geom(c).T; ++t) max_ws = std::max(max_ws, cls_geom(c).ws[t]);
  const unsigned long long ib = (((unsigned long long)g.N * g.Hi * g.Wi - 1) * g.ldi + g.Ci) * 4ull;
  const unsigned long long wp_bytes = (unsigned long long)(max_ws + 1) * p.rows_pad * g.Ci * 2ull * NP;
  LHG_REQUIRE(wp_bytes < (1ull << 32) - 64, "gather-GEMM (split mode): weight panels of 4 GiB and more are not supported");
  LHG_REQUIRE(p.planar_out || epilogue_window_fits(g, p.res ? p.ldres : 0, 4), "gather-GEMM: an output tile spans 1 GiB or more (Wo %d, ld %d)", g.Wo, g.ldo);
  const unsigned wb = (unsigned)wp_bytes;
  // workgroups of a bm x bn tiling; for a merged launch also the classes' first workgroups (written into p before it is copied)
  GGParams* q = &p;  // the parameter block being launched: p, or (variants 10..19) a copy of it holding one class
  auto blocks = [&](int bm, int bn) {
    const int nq = std::max(1, q->ncls);
    unsigned total = 0;
    for (int c = 0; c < nq; ++c) {
      q->cblk[c] = total;
      total += (unsigned)((((c == 0 ? q->g : q->gc[c - 1]).M + bm - 1) / bm) * (q->rows_pad / bn));
    }
    q->cblk[nq] = total;
    return total * (unsigned)std::max(1, q->ks);  // split K (single class): ks copies of the tile grid
  };
  auto launch3 = [&](auto kern, int bm, int bn, int threads) {
    const unsigned nb = blocks(bm, bn);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(threads), 0, st, *q, ib, wb);
  };
  const bool n128 = p.rows_pad % 128 == 0;
  // 10..19 (merged parity classes only): variant v - 10 in one launch per class — what wins on the layers whose classes are long
  // enough by themselves (512 -> 1024 @ 48^2: 493 us against 572 merged; 128 -> 256 @ 96^2: 267 against 206)
  const int NV = ncls > 1 ? 20 : 16;  // 14, 15 (round 5): gg4s 128 x 64 / 64 x 64 with swizzled unpadded LDS rows (three / four workgroups per CU);  // 12, 13 (single class, round 5): gg4s with two taps per barrier on 128 x 64 / 64 x 64 strips (Ci % 64 == 0); 3, 4: eight consumer waves (two per SIMD) on the 128x128 / 128x64 tiles; 5..8: strip-staged 3x3 kernel (gg4s);
                          // 9: 64 pixels x 128 output channels (half the activation splits of the 64x64 tile, same pixel granularity);
                          // 10, 11 (single class only): gg4s with eight consumer waves on 256 x 128 / 256 x 64 strips, one workgroup per CU
  // gg4s_kernel: fp16 planes, full 3x3 tap set, stride 1 both ways, same extents in and out
  bool strips = split_f16() && ncls == 1 && g.T == 9 && g.istep == 1 && g.ostep == 1 && g.oy0 == 0 && g.ox0 == 0 && g.gh == g.Hi && g.gw == g.Wi && g.Ho == g.Hi &&
                g.Wo == g.Wi && (long long)g.N * g.Hi * (g.Wi + 2) < (1ll << 31);
  if (strips) {
    int seen = 0;
    for (int t = 0; t < 9; ++t) {
      const int dyi = g.dy[t] + 1, dxi = g.dx[t] + 1;
      if ((unsigned)dyi < 3u && (unsigned)dxi < 3u) { p.tap_of[dyi * 3 + dxi] = t; seen |= 1 << (dyi * 3 + dxi); }
    }
    strips = seen == 0x1ff;
    bool fwd = true, rev = true;
    for (int i = 0; i < 9; ++i) { fwd = fwd && p.tap_of[i] == i; rev = rev && p.tap_of[i] == 8 - i; }
    p.strip_rev = rev ? 1 : 0;
    strips = strips && (fwd || rev);  // any other tap numbering would accumulate in another order than gg3s_kernel does
  }
  auto blocks_strip = [&](int bm, int bn) { return (unsigned)((((long long)g.N * g.Hi * (g.Wi + 2) + bm - 1) / bm) * (p.rows_pad / bn)) * (unsigned)std::max(1, p.ks); };
  const bool f16 = split_f16();
  // Split K for launches that cannot fill the chip (splitk_plan): ranges from the geometry alone
  p.ks = 1;
  {
    float* const ws = g_next_ws;
    const size_t ws_floats = g_next_ws_floats;
    g_next_ws = nullptr;
    g_next_ws_floats = 0;
    if (ncls == 1 && !p.planar_out && !p.ks_off && !p.tres_x) p.ks = splitk_plan(g.M, p.rows_pad, g.Ci, g.T, &p.ks_chunks);
    if (p.ks > 1) {
      const size_t floats = splitk_floats(p.ks, g.M, strips ? (long long)g.N * g.Hi * (g.Wi + 2) : 0, p.rows_pad);
      p.ks_slab = ws_floats >= floats ? ws : splitk_slab(st, floats);
      LHG_REQUIRE(p.ks_slab != nullptr, "gather-GEMM: no split-K workspace (%zu floats: pass one with lhg_gather_gemm_workspace — the library's own buffer cannot grow inside a graph capture)", floats);
    }
  }
  if (f16) LHG_REQUIRE(p.a_amax != nullptr && p.w_amax != nullptr, "gather-GEMM (fp32_split_f16 mode): the operand's absmax pointer is missing (lhg_absmax)");
  auto valid = [&](int v) {
    // 12 .. 15: round-5 experiments, bit-identical to the others, measured and NOT faster on any layer (tools/gg_tps_sweep.sh):
    // two taps per barrier runs level with the one-tap strips (the waits of the 64-output-channel layers are not the barriers), the
    // swizzled 128 x 64 tile reaches three workgroups per CU only by spilling 131 registers (4 x slower), the swizzled 64 x 64 tile is
    // level with the padded one.  Offered to the tuner / a forced choice only under LHG_GG_EXPERIMENTAL=1.
    static const bool experimental = [] { const char* e = getenv("LHG_GG_EXPERIMENTAL"); return e && atoi(e) != 0; }();
    if (p.tres_x) return ncls == 1 && f16 && strips && (v == 5 || v == 7 || v == 11);  // thin residual: the 64-column strip kernels
    if (ncls == 1 && v >= 12 && !experimental) return false;
    if (p.ks > 1 && (v == 12 || v == 13)) return false;  // two taps per barrier need an even number of tap steps in every K range
    if (ncls == 1 && v >= 14) return f16 && strips;
    if (ncls == 1 && v >= 12) return f16 && strips && g.Ci % 64 == 0;
    if (ncls == 1 && v >= 10) return f16 && strips && (v == 11 || n128);
    if (v >= 10) v -= 10;
    if (v == 9) return f16 && n128;
    if (v >= 5) return strips && (v == 5 || v == 7 || n128);
    return (v == 0 || v == 3) ? n128 : (NP == 3 || f16 || v < 3);
  };
  // (fourth template argument of gg3s_kernel: waves per SIMD the register allocation must leave room for — what the tile's LDS lets
  //  a CU hold; without it the straight-line epilogue is scheduled into twice the main loop's registers and halves the occupancy)
  auto run_one = [&](int v) {
    if (NP == 3) {
      switch (v) {
        case 3: launch3(gg3s_kernel<128, 128, 3, 1, 32, float, 8>, 128, 128, 768); break;
        case 4: launch3(gg3s_kernel<128, 64, 3, 1, 32, float, 8>, 128, 64, 768); break;
        case 0: launch3(gg3s_kernel<128, 128, 3, 2>, 128, 128, 512); break;
        case 1: launch3(gg3s_kernel<128, 64, 3, 2>, 128, 64, 512); break;
        case 2: launch3(gg3s_kernel<64, 64, 3, 2>, 64, 64, 512); break;
        default: break;
      }
    } else if (f16) {
      switch (v) {
        case 3: launch3(gg3s_kernel<128, 128, 2, 1, 32, float, 8, _Float16>, 128, 128, 768); break;
        case 4: launch3(gg3s_kernel<128, 64, 2, 6, 32, float, 8, _Float16>, 128, 64, 768); break;
        case 0: launch3(gg3s_kernel<128, 128, 2, 4, 32, float, 4, _Float16>, 128, 128, 512); break;
        case 1: launch3(gg3s_kernel<128, 64, 2, 4, 32, float, 4, _Float16>, 128, 64, 512); break;
        case 2: launch3(gg3s_kernel<64, 64, 2, 8, 32, float, 4, _Float16>, 64, 64, 512); break;
        case 5: hipLaunchKernelGGL((gg4s_kernel<64, 64>), dim3(blocks_strip(64, 64)), dim3(512), 0, st, p, ib, wb); break;
        case 6: hipLaunchKernelGGL((gg4s_kernel<64, 128>), dim3(blocks_strip(64, 128)), dim3(512), 0, st, p, ib, wb); break;
        case 7: hipLaunchKernelGGL((gg4s_kernel<128, 64>), dim3(blocks_strip(128, 64)), dim3(512), 0, st, p, ib, wb); break;
        case 8: hipLaunchKernelGGL((gg4s_kernel<128, 128>), dim3(blocks_strip(128, 128)), dim3(512), 0, st, p, ib, wb); break;
        case 9: launch3(gg3s_kernel<64, 128, 2, 4, 32, float, 4, _Float16>, 64, 128, 512); break;
        case 10: hipLaunchKernelGGL((gg4s_kernel<256, 128, 4, 2>), dim3(blocks_strip(256, 128)), dim3(768), 0, st, p, ib, wb); break;
        case 11: hipLaunchKernelGGL((gg4s_kernel<256, 64, 4, 2>), dim3(blocks_strip(256, 64)), dim3(768), 0, st, p, ib, wb); break;
        case 12: hipLaunchKernelGGL((gg4s_kernel<128, 64, 2, 2, 2>), dim3(blocks_strip(128, 64)), dim3(512), 0, st, p, ib, wb); break;
        case 13: hipLaunchKernelGGL((gg4s_kernel<64, 64, 2, 2, 2>), dim3(blocks_strip(64, 64)), dim3(512), 0, st, p, ib, wb); break;
        case 14: hipLaunchKernelGGL((gg4s_kernel<128, 64, 2, 2, 1, true>), dim3(blocks_strip(128, 64)), dim3(512), 0, st, p, ib, wb); break;
        case 15: hipLaunchKernelGGL((gg4s_kernel<64, 64, 2, 2, 1, true>), dim3(blocks_strip(64, 64)), dim3(512), 0, st, p, ib, wb); break;
        default: break;
      }
    } else {
      switch (v) {
        case 0: launch3(gg3s_kernel<128, 128, 2, 4>, 128, 128, 512); break;
        case 1: launch3(gg3s_kernel<128, 64, 2, 4>, 128, 64, 512); break;
        default: launch3(gg3s_kernel<64, 64, 2, 8>, 64, 64, 512); break;
      }
    }
  };
  auto run = [&](int v) {
    if (v < 10 || ncls == 1) { run_one(v); return; }
    for (int c = 0; c < ncls; ++c) {
      GGParams one = p;
      one.g = cls_geom(c);
      one.ncls = 0;
      q = &one;
      run_one(v - 10);
    }
    q = &p;
  };
  static const int forced = [] { const char* e = getenv("LHG_GGS_VARIANT"); return e ? atoi(e) : -1; }();
  int choice = (forced >= 0 && forced < NV && valid(forced)) ? forced : -1;
  if (choice < 0 && g_autotune_enabled) {
    const std::array<int, 12> key = {g.M, p.rows_pad, g.Ci, g.Co, g.T, g.istep, g.ostep, g.Hi, g.Wi, g.ldi, g.ldo,
                                     p.planar_out + 4 * g_precision + 64 * ncls};
    choice = autotuned_variant(g_gg_choice, key, NV, valid, run, st);
  }
  if (choice < 0) choice = p.tres_x ? 7 : (n128 && blocks(128, 128) >= 200 ? 0 : (blocks(128, 64) >= 256 ? 1 : 2));
  if (p.tres_x) LHG_REQUIRE(valid(choice), "gather-GEMM: the thin-residual epilogue needs a 3x3 stride-1 launch with 64 padded output channels in the fp32_split_f16 mode");
  // shape of the chosen tiling: pixels per M tile, consumer-wave rows per tile, padded strip coordinates or not
  int sh_bm = 128, sh_wgm = 2;
  bool sh_strip = false;
  if (NP == 3) { sh_bm = choice == 2 ? 64 : 128; sh_wgm = choice == 4 ? 4 : 2; }
  else if (f16) {
    switch (choice % (ncls > 1 ? 10 : 100)) {
      case 2: case 9: sh_bm = 64; break;
      case 4: sh_wgm = 4; break;
      case 5: case 6: case 13: case 15: sh_bm = 64; sh_strip = true; break;
      case 7: case 8: case 12: case 14: sh_strip = true; break;
      case 10: case 11: sh_bm = 256; sh_wgm = 4; sh_strip = true; break;
      default: break;  // 0, 1, 3: 128 pixels, two wave rows
    }
  } else sh_bm = choice <= 1 ? 128 : 64;
  const long long sh_mtot = sh_strip ? (long long)g.N * g.Hi * (g.Wi + 2) : (long long)g.M;
  if (p.stat_part) {
    LHG_REQUIRE(ncls == 1, "gather-GEMM: BatchNorm statistics rows are not available for merged parity classes");
    if (p.ks > 1) g_stat_rows = (int)(((sh_mtot + sh_bm - 1) / sh_bm * sh_bm + SK_ROWS - 1) / SK_ROWS);  // written by the split-K finish kernel
    else note_stat_rows(p, sh_mtot, sh_bm, sh_wgm);
  }
  double mt_sum = 0;  // sum over the classes of pixels x taps
  for (int c = 0; c < ncls; ++c) mt_sum += (double)cls_geom(c).M * cls_geom(c).T;
  ScopedKernelTime timed(0, st, 2.0 * mt_sum * (double)p.rows_pad * g.Ci);
  if (ncls == 1) timed.tag(g, p.rows_pad, g.Ci, choice, 2.0 * mt_sum * (double)p.rows_pad * g.Ci);
  else {
    Geom all = g;  // logged as one launch: total pixels, total taps
    all.M = 0; all.T = 0;
    for (int c = 0; c < ncls; ++c) { all.M += cls_geom(c).M; all.T += cls_geom(c).T; }
    timed.tag(all, p.rows_pad, g.Ci, choice, 2.0 * mt_sum * (double)p.rows_pad * g.Ci);
  }
  run(choice);
  if (p.ks > 1) {  // the ranges' raw accumulators -> the output (epilogue arithmetic, max|.|, statistics rows)
    SplitKFinish f{};
    f.slab = p.ks_slab; f.ks = p.ks; f.m_rows = (int)((sh_mtot + sh_bm - 1) / sh_bm * sh_bm); f.rows_pad = p.rows_pad;
    f.slab_stride = (long long)f.m_rows * p.rows_pad;
    f.strip = sh_strip ? 1 : 0; f.N = g.N; f.H = g.Hi; f.W = g.Wi;
    f.M = g.M; f.gh = g.gh; f.gw = g.gw; f.Ho = g.Ho; f.Wo = g.Wo; f.oy0 = g.oy0; f.ox0 = g.ox0; f.ostep = g.ostep;
    f.out = p.out; f.ldo = g.ldo; f.Co = g.Co;
    f.bias = p.bias; f.scale = p.scale; f.shift = p.shift; f.res = p.res; f.ldres = p.ldres; f.act = p.act; f.slope = p.slope;
    f.a_amax = p.a_amax; f.w_amax = p.w_amax; f.out_amax = p.out_amax; f.stat = p.stat_part;
    hipLaunchKernelGGL(gg_splitk_finish_kernel, dim3((f.m_rows + SK_ROWS - 1) / SK_ROWS, (p.rows_pad + 63) / 64), dim3(256), 0, st, f);
  }
  return check_launch("gg3s_kernel");
}

static int launch_wg(WGParams& p, int S, hipStream_t st, const WG3Params* p3 = nullptr) {
  const Geom& g = p.g;
  p.xcd = xcd_order();
  LHG_REQUIRE(g.ldi % 4 == 0 && g.ldo % 4 == 0 && g.Ci % 4 == 0 && g.Co % 4 == 0, "wgrad: channel counts / strides must be multiples of 4");
  LHG_REQUIRE((reinterpret_cast<uintptr_t>(p.in) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.gout) & 15) == 0, "wgrad: inputs must be 16-byte aligned");
  LHG_REQUIRE(p.m_pad % 64 == 0 && p.n_pad % 64 == 0 && p.m_pad >= g.Ci && p.n_pad >= g.Co, "wgrad: bad padded extents");
  LHG_REQUIRE(S >= 1 && S <= 65535, "wgrad: bad split count %d", S);
  const int steps = (g.M + BK - 1) / BK;
  p.kchunk = ((steps + S - 1) / S) * BK;
  const bool act16 = act_is_bf16();
  LHG_REQUIRE(!act16 || g_precision == LHG_PRECISION_BF16, "bf16 activation storage needs the bf16 conv precision");
  const unsigned long long es = act16 ? 2ull : 4ull;
  const unsigned long long in_bytes = (((unsigned long long)g.N * g.Hi * g.Wi - 1) * g.ldi + g.Ci) * es;
  const unsigned long long go_bytes = (((unsigned long long)g.N * g.Ho * g.Wo - 1) * g.ldo + g.Co) * es;
  const bool small = in_bytes < (1ull << 32) - 64 && go_bytes < (1ull << 32) - 64;
  const unsigned ib = (unsigned)in_bytes, gb = (unsigned)go_bytes;
  const bool m128 = p.m_pad % 128 == 0, n128 = p.n_pad % 128 == 0;
  // 5: nine-tap fused kernel for 3x3 stride 1; 6..9: bf16-operand kernels; 10..13: split (fp32-faithful) kernels;
  // 14..17: bf16 STORAGE kernels with transposed LDS reads (wg3b_kernel); 18..21: fp16-split kernels with transposed LDS reads (wg4s_kernel)
  // 22..25: the same with producer / consumer waves (wg5p_kernel)
  constexpr int NV = 26;
  const bool bf16 = g_precision == LHG_PRECISION_BF16;
  const bool split = split_mode();
  const bool f16 = split_f16();
  const int NP = split_planes();
  if (f16) LHG_REQUIRE(p.in_amax != nullptr && p.gout_amax != nullptr, "wgrad (fp32_split_f16 mode): the operands' per-channel absmax vectors are missing (lhg_channel_absmax)");
  if (bf16 || split) LHG_REQUIRE(small, "wgrad (bf16 / split mode): tensors of 4 GiB and more are not supported");
  auto valid = [&](int v) {
    if (v >= 22) return f16 && (v == 22 ? m128 && n128 : v == 23 ? m128 : v == 24 ? n128 : true);
    if (v >= 18) return f16 && (v == 18 ? m128 && n128 : v == 19 ? m128 : v == 20 ? n128 : true);
    if (v >= 14) return bf16 && act16 && (v == 14 ? m128 && n128 : v == 15 ? m128 : v == 16 ? n128 : true);
    if (f16) return false;  // (wg2s_kernel's fp16 form scales per tensor: not offered next to the per-channel kernels above)
    if (split) return v >= 10 && (v == 10 ? m128 && n128 : v == 11 ? m128 : v == 12 ? n128 : true);
    if (v >= 10) return false;
    if (bf16) return v >= 6 && (v == 6 ? m128 && n128 : v == 7 ? m128 : v == 8 ? n128 : true);
    if (v >= 6) return false;
    if (v == 5) return p3 != nullptr && small;
    return v == 4 || (small && (v == 0 ? m128 && n128 : v == 1 ? m128 : v == 2 ? n128 : true));
  };
  auto run = [&](int v) {
    auto grid = [&](int bm, int bn) { return dim3((p.m_pad / bm) * (p.n_pad / bn), g.T, S); };
    switch (v) {
      case 0: hipLaunchKernelGGL((wg2_kernel<128, 128>), grid(128, 128), dim3(256), 0, st, p, ib, gb); break;
      case 1: hipLaunchKernelGGL((wg2_kernel<128, 64>), grid(128, 64), dim3(256), 0, st, p, ib, gb); break;
      case 2: hipLaunchKernelGGL((wg2_kernel<64, 128>), grid(64, 128), dim3(256), 0, st, p, ib, gb); break;
      case 3: hipLaunchKernelGGL((wg2_kernel<64, 64>), grid(64, 64), dim3(256), 0, st, p, ib, gb); break;
      case 5: {
        WG3Params q3 = *p3;
        q3.xcd = p.xcd;
        hipLaunchKernelGGL(wg3_kernel, dim3((p.m_pad / 64) * (p.n_pad / 64), 1, S), dim3(256), 0, st, q3, ib, gb);
        break;
      }
      case 6:
        if (act16) hipLaunchKernelGGL((wg2b_kernel<128, 128, __bf16>), grid(128, 128), dim3(256), 0, st, p, ib, gb);
        else hipLaunchKernelGGL((wg2b_kernel<128, 128, float>), grid(128, 128), dim3(256), 0, st, p, ib, gb);
        break;
      case 7:
        if (act16) hipLaunchKernelGGL((wg2b_kernel<128, 64, __bf16>), grid(128, 64), dim3(256), 0, st, p, ib, gb);
        else hipLaunchKernelGGL((wg2b_kernel<128, 64, float>), grid(128, 64), dim3(256), 0, st, p, ib, gb);
        break;
      case 8:
        if (act16) hipLaunchKernelGGL((wg2b_kernel<64, 128, __bf16>), grid(64, 128), dim3(256), 0, st, p, ib, gb);
        else hipLaunchKernelGGL((wg2b_kernel<64, 128, float>), grid(64, 128), dim3(256), 0, st, p, ib, gb);
        break;
      case 9:
        if (act16) hipLaunchKernelGGL((wg2b_kernel<64, 64, __bf16>), grid(64, 64), dim3(256), 0, st, p, ib, gb);
        else hipLaunchKernelGGL((wg2b_kernel<64, 64, float>), grid(64, 64), dim3(256), 0, st, p, ib, gb);
        break;
      case 18: hipLaunchKernelGGL((wg4s_kernel<128, 128>), grid(128, 128), dim3(256), 0, st, p, ib, gb); break;
      case 19: hipLaunchKernelGGL((wg4s_kernel<128, 64>), grid(128, 64), dim3(256), 0, st, p, ib, gb); break;
      case 20: hipLaunchKernelGGL((wg4s_kernel<64, 128>), grid(64, 128), dim3(256), 0, st, p, ib, gb); break;
      case 21: hipLaunchKernelGGL((wg4s_kernel<64, 64>), grid(64, 64), dim3(256), 0, st, p, ib, gb); break;
      case 22: hipLaunchKernelGGL((wg5p_kernel<128, 128>), grid(128, 128), dim3(512), 0, st, p, ib, gb); break;
      case 23: hipLaunchKernelGGL((wg5p_kernel<128, 64>), grid(128, 64), dim3(512), 0, st, p, ib, gb); break;
      case 24: hipLaunchKernelGGL((wg5p_kernel<64, 128>), grid(64, 128), dim3(512), 0, st, p, ib, gb); break;
      case 25: hipLaunchKernelGGL((wg5p_kernel<64, 64>), grid(64, 64), dim3(512), 0, st, p, ib, gb); break;
      case 14: hipLaunchKernelGGL((wg3b_kernel<128, 128>), grid(128, 128), dim3(256), 0, st, p, ib, gb); break;
      case 15: hipLaunchKernelGGL((wg3b_kernel<128, 64>), grid(128, 64), dim3(256), 0, st, p, ib, gb); break;
      case 16: hipLaunchKernelGGL((wg3b_kernel<64, 128>), grid(64, 128), dim3(256), 0, st, p, ib, gb); break;
      case 17: hipLaunchKernelGGL((wg3b_kernel<64, 64>), grid(64, 64), dim3(256), 0, st, p, ib, gb); break;
      case 10:
        if (NP == 3) hipLaunchKernelGGL((wg2s_kernel<128, 128, 3, 2>), grid(128, 128), dim3(512), 0, st, p, ib, gb);
        else if (f16) hipLaunchKernelGGL((wg2s_kernel<128, 128, 2, 2, _Float16>), grid(128, 128), dim3(512), 0, st, p, ib, gb);
        else hipLaunchKernelGGL((wg2s_kernel<128, 128, 2, 2>), grid(128, 128), dim3(512), 0, st, p, ib, gb);
        break;
      case 11:
        if (NP == 3) hipLaunchKernelGGL((wg2s_kernel<128, 64, 3, 2>), grid(128, 64), dim3(512), 0, st, p, ib, gb);
        else if (f16) hipLaunchKernelGGL((wg2s_kernel<128, 64, 2, 2, _Float16>), grid(128, 64), dim3(512), 0, st, p, ib, gb);
        else hipLaunchKernelGGL((wg2s_kernel<128, 64, 2, 2>), grid(128, 64), dim3(512), 0, st, p, ib, gb);
        break;
      case 12:
        if (NP == 3) hipLaunchKernelGGL((wg2s_kernel<64, 128, 3, 2>), grid(64, 128), dim3(512), 0, st, p, ib, gb);
        else if (f16) hipLaunchKernelGGL((wg2s_kernel<64, 128, 2, 2, _Float16>), grid(64, 128), dim3(512), 0, st, p, ib, gb);
        else hipLaunchKernelGGL((wg2s_kernel<64, 128, 2, 2>), grid(64, 128), dim3(512), 0, st, p, ib, gb);
        break;
      case 13:
        if (NP == 3) hipLaunchKernelGGL((wg2s_kernel<64, 64, 3, 2>), grid(64, 64), dim3(512), 0, st, p, ib, gb);
        else if (f16) hipLaunchKernelGGL((wg2s_kernel<64, 64, 2, 2, _Float16>), grid(64, 64), dim3(512), 0, st, p, ib, gb);
        else hipLaunchKernelGGL((wg2s_kernel<64, 64, 2, 2>), grid(64, 64), dim3(512), 0, st, p, ib, gb);
        break;
      default: hipLaunchKernelGGL(wg_kernel, grid(64, 64), dim3(256), 0, st, p); break;
    }
  };
  static const int forced = [] { const char* e = getenv("LHG_WG_VARIANT"); return e ? atoi(e) : -1; }();
  static const bool tune = [] { const char* e = getenv("LHG_AUTOTUNE"); return e ? atoi(e) != 0 : true; }();
  int choice = -1;
  if (forced >= 0 && forced < NV && valid(forced)) choice = forced;
  if (choice < 0 && tune && g_autotune_enabled) {
    const std::array<int, 12> key = {g.M, p.m_pad, p.n_pad, g.T, S, g.istep, g.ostep, g.Hi, g.Wi, g.ldi, g.ldo, g.ws[0] + 100 * g_precision + (act16 ? 1000 : 0)};
    choice = autotuned_variant(g_wg_choice, key, NV, valid, run, st);
  }
  if (choice < 0 && f16) choice = m128 && n128 && (p.m_pad / 128) * (p.n_pad / 128) * g.T * S >= 400 ? 18 : 21;
  if (choice < 0 && split) choice = m128 && n128 && (p.m_pad / 128) * (p.n_pad / 128) * g.T * S >= 200 ? 10 : 13;
  if (choice < 0 && bf16 && act16) choice = m128 && n128 && (p.m_pad / 128) * (p.n_pad / 128) * g.T * S >= 400 ? 14 : 17;
  if (choice < 0 && bf16) choice = m128 && n128 && (p.m_pad / 128) * (p.n_pad / 128) * g.T * S >= 400 ? 6 : 9;
  if (choice < 0) choice = small ? (p3 ? 5 : (m128 && n128 && (p.m_pad / 128) * (p.n_pad / 128) * g.T * S >= 400 ? 0 : 3)) : 4;
  ScopedKernelTime timed(1, st, 2.0 * g.M * (double)p.m_pad * p.n_pad * g.T);
  timed.tag(g, p.m_pad, p.n_pad, choice + 100 * S, 2.0 * g.M * (double)p.m_pad * p.n_pad * g.T);
  run(choice);
  return check_launch("wg_kernel");
}

static inline int pad64(int c) { return (c + 63) / 64 * 64; }

// Split count for the pixel (K) axis of a weight gradient.  Every split writes a full dW-sized slab that lhg_wgrad_reduce reads back, so S
// is the factor by which the launch's output traffic exceeds dW itself: round 2 aimed at 1536 workgroups of the LARGEST tile the
// extents allow (S = 6 on a 1024 -> 512 layer whose 288 tile x tap pairs already fill the chip: 113 MB of slabs for an 18.9 MB
// gradient; 2.5 GB per train step in all, read again by the reduction).  Two policies (LHG_WG_POLICY):
//   "fill" (default)  the SMALLEST S whose workgroups of the largest tile fill >= 90 % (LHG_WG_FILL) of the chip's resident slots in their last wave
//                     (512 slots for 128-wide tiles: two workgroups per CU; 768 for 64 x 64) — quantisation is what an unsplit or thinly
//                     split launch loses (288 workgroups on 512 slots: 0.56), while a slab costs only ~2 x dW bytes of traffic;
//   "bytes"           S from the smallest tile and LHG_WG_TARGET = 1024 workgroups: the fewest slab bytes (S = 1 on the wide layers).
// LHG_WG_BASE_TILE=128 LHG_WG_POLICY=bytes LHG_WG_TARGET=1536 is round 2's rule.
static int pick_splits(long long pixels, int m_pad, int n_pad, int taps) {
  static const bool fill = [] { const char* e = getenv("LHG_WG_POLICY"); return !e || std::string(e) != "bytes"; }();
  const long long steps = (pixels + BK - 1) / BK;
  const long long cap = std::min<long long>(steps / 8 > 1 ? steps / 8 : 1, 4096);
  if (fill) {
    const bool big = m_pad % 128 == 0 && n_pad % 128 == 0;
    const int t = big ? 128 : 64;
    static const double want = [] { const char* e = getenv("LHG_WG_FILL"); return e ? atof(e) : 0.9; }();
    static const long long slots_big = [] { const char* e = getenv("LHG_WG_SLOTS"); return e ? atoll(e) : 512ll; }();
    const long long tiles = (long long)(m_pad / t) * (n_pad / t) * taps, slots = big ? slots_big : slots_big * 3 / 2;
    long long best = 1;
    double best_util = 0;
    for (long long s = 1; s <= cap && tiles * s <= 4 * slots; ++s) {
      const long long wg = tiles * s, waves = (wg + slots - 1) / slots;
      const double util = (double)wg / (double)(waves * slots);
      if (util >= want) return (int)s;
      if (util > best_util) { best_util = util; best = s; }
    }
    return (int)best;
  }
  static const int base = [] { const char* e = getenv("LHG_WG_BASE_TILE"); return e && atoi(e) == 128 ? 128 : 64; }();
  const int bm = (base == 128 && m_pad % 128 == 0) ? 128 : 64, bn = (base == 128 && n_pad % 128 == 0) ? 128 : 64;
  const long long tiles = (long long)(m_pad / bm) * (n_pad / bn) * taps;
  static const long long target = [] { const char* e = getenv("LHG_WG_TARGET"); return e ? atoll(e) : 1024ll; }();
  long long s = (target + tiles - 1) / tiles;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  return (int)s;
}


// geometry of y = conv2d(x) (also used, read as "gout", for its weight gradient)
static void conv_fwd_geom(Geom& g, int N, int H, int W, int Ci, int ldx, int Co, int ldy, int KH, int KW, int stride) {
  const int ph = KH / 2, pw = KW / 2;
  g.N = N; g.Hi = H; g.Wi = W; g.Ci = Ci; g.ldi = ldx;
  g.Ho = (H + 2 * ph - KH) / stride + 1; g.Wo = (W + 2 * pw - KW) / stride + 1; g.Co = Co; g.ldo = ldy;
  g.gh = g.Ho; g.gw = g.Wo; g.oy0 = g.ox0 = 0; g.ostep = 1; g.istep = stride;
  g.T = KH * KW;
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw) {
      const int t = kh * KW + kw;
      g.dy[t] = kh - ph; g.dx[t] = kw - pw; g.ws[t] = t;
    }
  g.M = N * g.gh * g.gw;
}

static bool conv_args_ok(int KH, int KW, int stride) {
  return KH == KW && (KH == 1 || KH == 3) && (stride == 1 || stride == 2);
}

}  // namespace lhg

using namespace lhg;

extern "C" {

int lhg_autotune(int on) {
  g_autotune_enabled = on != 0;
  return LHG_OK;
}

int lhg_profile_enable(int kernel, int on) {
  LHG_REQUIRE(kernel == 0 || kernel == 1, "profile_enable: kernel must be 0 (gather-GEMM) or 1 (wgrad-GEMM)");
  g_timer[kernel].on = on != 0;
  if (on) { g_timer[kernel].used = 0; g_timer[kernel].executed_flops = 0; g_timer[kernel].tags.clear(); }
  return LHG_OK;
}

int lhg_profile_read(int kernel, double* total_ms, long long* launches, double* executed_flops) {
  LHG_REQUIRE(kernel == 0 || kernel == 1, "profile_read: kernel must be 0 or 1");
  KernelTimer& t = g_timer[kernel];
  double ms = 0;
  for (size_t i = 0; i < t.used; ++i) {
    hipError_t e = hipEventSynchronize(t.pool[i].second);
    if (e != hipSuccess) return fail(LHG_E_LAUNCH, "profile_read: %s", hipGetErrorString(e));
    float dt = 0;
    (void)hipEventElapsedTime(&dt, t.pool[i].first, t.pool[i].second);
    ms += dt;
  }
  if (const char* path = getenv("LHG_PROFILE_LOG")) {  // debugging aid: one CSV line per launch (tools/layer_table.py)
    if (FILE* f = fopen(path, "a")) {
      for (size_t i = 0; i < t.used && i < t.tags.size(); ++i) {
        float dt = 0;
        (void)hipEventElapsedTime(&dt, t.pool[i].first, t.pool[i].second);
        fprintf(f, "%d,%s,%.4f\n", kernel, t.tags[i].c_str(), dt);
      }
      fclose(f);
    }
  }
  *total_ms = ms;
  *launches = (long long)t.used;
  *executed_flops = t.executed_flops;
  return LHG_OK;
}

int lhg_pack_weight(const float* w, int D0, int D1, int KH, int KW, int rows_from_d0, float* dst, int rows_pad, int k_pad, lhg_stream_t s) {
  const int rows = rows_from_d0 ? D0 : D1, K = rows_from_d0 ? D1 : D0;
  LHG_REQUIRE(rows_pad >= rows && k_pad >= K && rows_pad % 64 == 0 && k_pad % 32 == 0, "pack_weight: bad padding (%d>=%d, %d>=%d)", rows_pad, rows, k_pad, K);
  const size_t total = (size_t)KH * KW * rows_pad * k_pad;
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
  if (g_precision == LHG_PRECISION_F32_SPLIT)
    hipLaunchKernelGGL((pack_weight_split_kernel<3, __bf16>), dim3(blocks), dim3(256), 0, as_stream(s), w, D0, D1, KH * KW, rows_from_d0,
                       reinterpret_cast<__bf16*>(dst), rows_pad, k_pad, (const float*)nullptr);
  else if (g_precision == LHG_PRECISION_F32_SPLIT2)
    hipLaunchKernelGGL((pack_weight_split_kernel<2, __bf16>), dim3(blocks), dim3(256), 0, as_stream(s), w, D0, D1, KH * KW, rows_from_d0,
                       reinterpret_cast<__bf16*>(dst), rows_pad, k_pad, (const float*)nullptr);
  else if (g_precision == LHG_PRECISION_F32_SPLIT_F16) {
    float* amax = dst + total;  // two fp16 planes = 4 bytes per element: the panels end `total` floats in
    int rc = launch_absmax(w, (long long)D0 * D1 * KH * KW, 1, 1, amax, as_stream(s), true);
    if (rc) return rc;
    hipLaunchKernelGGL((pack_weight_split_kernel<2, _Float16>), dim3(blocks), dim3(256), 0, as_stream(s), w, D0, D1, KH * KW, rows_from_d0,
                       reinterpret_cast<_Float16*>(dst), rows_pad, k_pad, (const float*)amax);
  }
  else if (g_precision == LHG_PRECISION_BF16)
    hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(blocks), dim3(256), 0, as_stream(s), w, D0, D1, KH * KW, rows_from_d0,
                       reinterpret_cast<__bf16*>(dst), rows_pad, k_pad);
  else
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, as_stream(s), w, D0, D1, KH * KW, rows_from_d0, dst, rows_pad, k_pad);
  return check_launch("pack_weight");
}

int lhg_pack_weights(const lhg_pack_item* items, int n, lhg_stream_t s) {
  LHG_REQUIRE(n >= 0 && (n == 0 || items != nullptr), "pack_weights: bad item list");
  if (g_precision == LHG_PRECISION_BF16) {  // bf16 panels: one launch per PACK_BATCH weights (no max|w|)
    for (int first = 0; first < n; first += PACK_BATCH) {
      PackBatch b;
      b.n = std::min(PACK_BATCH, n - first);
      unsigned blk = 0;
      for (int i = 0; i < b.n; ++i) {
        const lhg_pack_item& q = items[first + i];
        const int rows = q.rows_from_d0 ? q.D0 : q.D1, K = q.rows_from_d0 ? q.D1 : q.D0;
        LHG_REQUIRE(q.w && q.dst && q.D0 > 0 && q.D1 > 0 && q.KH > 0 && q.KW > 0, "pack_weights: item %d is empty", first + i);
        LHG_REQUIRE(q.rows_pad >= rows && q.k_pad >= K && q.rows_pad % 64 == 0 && q.k_pad % 32 == 0, "pack_weights: bad padding of item %d (%d>=%d, %d>=%d)",
                    first + i, q.rows_pad, rows, q.k_pad, K);
        const size_t total = (size_t)q.KH * q.KW * q.rows_pad * q.k_pad;
        PackItem& it = b.it[i];
        it.w = q.w;
        it.dst = reinterpret_cast<_Float16*>(q.dst);
        it.amax = nullptr;
        it.amax_src = nullptr;
        it.D0 = q.D0; it.D1 = q.D1; it.T = q.KH * q.KW; it.rows_from_d0 = q.rows_from_d0; it.rows_pad = q.rows_pad; it.k_pad = q.k_pad;
        it.blk0 = blk;
        it.ablk0 = 0;
        blk += (unsigned)std::min<size_t>((total + 1023) / 1024, 16384);  // 4 elements per thread
      }
      b.blocks = blk;
      b.ablocks = 0;
      hipLaunchKernelGGL(pack_batch_bf16_kernel, dim3(blk), dim3(256), 0, as_stream(s), b);
    }
    return check_launch("pack_weights");
  }
  if (g_precision != LHG_PRECISION_F32_SPLIT_F16) {  // the other modes have no max|w| pass to share: one pack launch per weight
    for (int i = 0; i < n; ++i) {
      const lhg_pack_item& q = items[i];
      const int rc = lhg_pack_weight(q.w, q.D0, q.D1, q.KH, q.KW, q.rows_from_d0, q.dst, q.rows_pad, q.k_pad, s);
      if (rc) return rc;
    }
    return LHG_OK;
  }
  for (int first = 0; first < n; first += PACK_BATCH) {
    PackBatch b;
    b.n = std::min(PACK_BATCH, n - first);
    unsigned blk = 0, ablk = 0;
    bool tiled = true;  // every weight of the batch has at most PACK_TILE_TMAX taps (3 x 3): the LDS-tiled pack
    for (int i = 0; i < b.n; ++i) tiled = tiled && items[first + i].KH * items[first + i].KW <= PACK_TILE_TMAX;
    for (int i = 0; i < b.n; ++i) {
      const lhg_pack_item& q = items[first + i];
      const int rows = q.rows_from_d0 ? q.D0 : q.D1, K = q.rows_from_d0 ? q.D1 : q.D0;
      LHG_REQUIRE(q.w && q.dst && q.D0 > 0 && q.D1 > 0 && q.KH > 0 && q.KW > 0, "pack_weights: item %d is empty", first + i);
      LHG_REQUIRE(q.rows_pad >= rows && q.k_pad >= K && q.rows_pad % 64 == 0 && q.k_pad % 32 == 0, "pack_weights: bad padding of item %d (%d>=%d, %d>=%d)",
                  first + i, q.rows_pad, rows, q.k_pad, K);
      const size_t total = (size_t)q.KH * q.KW * q.rows_pad * q.k_pad, elems = (size_t)q.D0 * q.D1 * q.KH * q.KW;
      PackItem& it = b.it[i];
      it.w = q.w;
      it.dst = reinterpret_cast<_Float16*>(q.dst);
      it.amax = reinterpret_cast<unsigned*>(q.dst + total);
      it.D0 = q.D0; it.D1 = q.D1; it.T = q.KH * q.KW; it.rows_from_d0 = q.rows_from_d0; it.rows_pad = q.rows_pad; it.k_pad = q.k_pad;
      it.blk0 = blk;
      it.ablk0 = ablk;
      it.amax_src = it.amax;
      for (int j = 0; j < i; ++j)  // the same weight earlier in the batch (forward panels and input-gradient panels): measured once
        if (b.it[j].w == it.w && b.it[j].D0 == it.D0 && b.it[j].D1 == it.D1 && b.it[j].T == it.T && b.it[j].amax_src == b.it[j].amax) { it.amax_src = b.it[j].amax; break; }
      blk += tiled ? (unsigned)std::min<size_t>((size_t)(q.rows_pad / 32) * (q.k_pad / 32), 16384)  // one 32 x 32 tile (all taps) per workgroup
                   : (unsigned)std::min<size_t>((total + 1023) / 1024, 16384);                    // 4 elements per thread
      if (it.amax_src == it.amax) ablk += (unsigned)std::min<size_t>((elems + 2047) / 2048, 2048);
    }
    b.blocks = blk;
    b.ablocks = ablk;
    hipLaunchKernelGGL(pack_batch_zero_kernel, dim3(1), dim3(PACK_BATCH), 0, as_stream(s), b);
    if (ablk) hipLaunchKernelGGL(pack_batch_absmax_kernel, dim3(ablk), dim3(256), 0, as_stream(s), b);
    if (tiled) hipLaunchKernelGGL(pack_batch_tiled_kernel, dim3(blk), dim3(256), 0, as_stream(s), b);
    else hipLaunchKernelGGL(pack_batch_kernel, dim3(blk), dim3(256), 0, as_stream(s), b);
  }
  return check_launch("pack_weights");
}

int lhg_set_conv_precision(int precision) {
  LHG_REQUIRE(precision >= LHG_PRECISION_F32 && precision <= LHG_PRECISION_F32_SPLIT_F16, "set_conv_precision: unknown precision %d", precision);
  g_precision = precision;
  return LHG_OK;
}

int lhg_get_conv_precision(void) { return g_precision; }
int lhg_default_conv_precision(void) { return default_precision(); }

long long lhg_packed_weight_floats(int taps, int rows_pad, int k_pad) {
  const long long elems = (long long)taps * rows_pad * k_pad;
  switch (g_precision) {
    case LHG_PRECISION_F32_SPLIT: return elems * 3 / 2;   // three bf16 planes
    case LHG_PRECISION_F32_SPLIT2: return elems;          // two bf16 planes
    case LHG_PRECISION_F32_SPLIT_F16: return elems + 4;   // two fp16 planes + 16 bytes: max|w| behind the panels
    default: return elems;                                // fp32 panels (bf16 panels use half of it)
  }
}

int lhg_conv2d_forward(const float* x, int N, int H, int W, int Ci, int ldx, const float* wp, int rows_pad, int KH, int KW, int stride,
                       float* y, int Co, int ldy, const float* bias, const float* scale, const float* shift,
                       const float* res, int ldres, int act, float slope, int planar_out, const float* x_absmax, float* y_absmax,
                       lhg_stream_t s) {
  LHG_REQUIRE(conv_args_ok(KH, KW, stride), "conv2d_forward: unsupported kernel %dx%d stride %d", KH, KW, stride);
  GGParams p{};
  conv_fwd_geom(p.g, N, H, W, Ci, ldx, Co, ldy, KH, KW, stride);
  p.a_amax = x_absmax; p.w_amax = weight_amax(wp, KH * KW, rows_pad, Ci); p.out_amax = y_absmax;
  p.in = x; p.wp = wp; p.out = y; p.bias = bias; p.scale = scale; p.shift = shift; p.res = res; p.ldres = ldres;
  p.rows_pad = rows_pad; p.act = act; p.slope = slope; p.planar_out = planar_out;
  return launch_gg(p, as_stream(s));
}

int lhg_conv2d_forward_thin_res(const float* x, int N, int H, int W, int Ci, int ldx, const float* wp, int rows_pad, int KH, int KW, int stride,
                                float* y, int Co, int ldy, const float* bias, const float* scale, const float* shift,
                                const float* res_x_nchw, int res_c, const float* res_w, const float* res_b, int act, float slope,
                                const float* x_absmax, float* y_absmax, lhg_stream_t s) {
  LHG_REQUIRE(KH == 3 && KW == 3 && stride == 1, "conv2d_forward_thin_res: 3x3 stride-1 convolutions only (%dx%d stride %d)", KH, KW, stride);
  LHG_REQUIRE(g_precision == LHG_PRECISION_F32_SPLIT_F16 && !act_is_bf16(), "conv2d_forward_thin_res: fp32_split_f16 mode, fp32 tensors");
  LHG_REQUIRE(rows_pad == 64 && Co <= 64, "conv2d_forward_thin_res: at most 64 output channels (Co %d, rows_pad %d)", Co, rows_pad);
  LHG_REQUIRE(res_x_nchw != nullptr && res_w != nullptr && res_c >= 1 && res_c <= 4, "conv2d_forward_thin_res: the shortcut's input has 1 - 4 channels (%d)", res_c);
  LHG_REQUIRE((long long)H * W >= 512 && (long long)N * H * W < (1ll << 31), "conv2d_forward_thin_res: images of at least 512 pixels (an output tile spans at most two)");
  LHG_REQUIRE(act == LHG_ACT_NONE || act == LHG_ACT_RELU || act == LHG_ACT_LEAKY, "conv2d_forward_thin_res: activation %d", act);
  GGParams p{};
  conv_fwd_geom(p.g, N, H, W, Ci, ldx, Co, ldy, KH, KW, stride);
  p.a_amax = x_absmax; p.w_amax = weight_amax(wp, KH * KW, rows_pad, Ci); p.out_amax = y_absmax;
  p.in = x; p.wp = wp; p.out = y; p.bias = bias; p.scale = scale; p.shift = shift;
  p.rows_pad = rows_pad; p.act = act; p.slope = slope; p.planar_out = 0;
  p.tres_x = res_x_nchw; p.tres_w = res_w; p.tres_b = res_b; p.tres_c = res_c; p.tres_plane = H * W;
  return launch_gg(p, as_stream(s));
}

long long lhg_gather_gemm_splitk_floats(long long M, long long M_padded, int rows_pad, int K, int taps) {
  const int ks = splitk_plan(M, rows_pad, K, taps, nullptr);
  return ks > 1 ? (long long)splitk_floats(ks, M, M_padded, rows_pad) : 0;
}

int lhg_gather_gemm_workspace(float* ws, long long floats) {
  g_next_ws = floats > 0 ? ws : nullptr;
  g_next_ws_floats = (ws && floats > 0) ? (size_t)floats : 0;
  return LHG_OK;
}

long long lhg_conv2d_stats_rows_bound(int N, int Ho, int Wo) {
  // the finest tiling writes one row per 32 pixels of the PADDED pixel axis of the strip kernels (every image row two pixels longer)
  // (the split-K finish kernel: one row per 16)
  return ((long long)N * Ho * (Wo + 2) + 255) / 256 * 16 + 16;
}

int lhg_conv2d_forward_stats(const float* x, int N, int H, int W, int Ci, int ldx, const float* wp, int rows_pad, int KH, int KW, int stride,
                             float* y, int Co, int ldy, const float* bias, const float* x_absmax, float* y_absmax, float* stat_partial,
                             int* stat_rows, lhg_stream_t s) {
  LHG_REQUIRE(conv_args_ok(KH, KW, stride), "conv2d_forward_stats: unsupported kernel %dx%d stride %d", KH, KW, stride);
  LHG_REQUIRE(stat_partial != nullptr && stat_rows != nullptr, "conv2d_forward_stats: the partial-row buffer and the row count are required");
  GGParams p{};
  conv_fwd_geom(p.g, N, H, W, Ci, ldx, Co, ldy, KH, KW, stride);
  p.a_amax = x_absmax; p.w_amax = weight_amax(wp, KH * KW, rows_pad, Ci); p.out_amax = y_absmax;
  p.in = x; p.wp = wp; p.out = y; p.bias = bias;
  p.rows_pad = rows_pad; p.act = LHG_ACT_NONE; p.slope = 0.f; p.planar_out = 0;
  p.stat_part = stat_partial;
  g_stat_rows = 0;
  const int rc = launch_gg(p, as_stream(s));
  *stat_rows = rc == LHG_OK ? g_stat_rows : 0;  // 0: this launch left no rows (a kernel without the shared epilogue): run lhg_bn_stats
  LHG_REQUIRE(rc != LHG_OK || (long long)*stat_rows <= lhg_conv2d_stats_rows_bound(N, p.g.Ho, p.g.Wo), "conv2d_forward_stats: %d rows exceed the bound", *stat_rows);
  return rc;
}

int lhg_conv2d_backward_input(const float* gy, int N, int H, int W, int Co, int ldgy, const float* wp, int rows_pad, int KH, int KW, int stride,
                              float* gx, int Ci, int ldgx, const float* gy_absmax, lhg_stream_t s) {
  return lhg_conv2d_backward_input_add(gy, N, H, W, Co, ldgy, wp, rows_pad, KH, KW, stride, gx, Ci, ldgx, nullptr, 0, gy_absmax, s);
}

int lhg_conv2d_backward_input_add(const float* gy, int N, int H, int W, int Co, int ldgy, const float* wp, int rows_pad, int KH, int KW, int stride,
                                  float* gx, int Ci, int ldgx, const float* res, int ldres, const float* gy_absmax, lhg_stream_t s) {
  return lhg_conv2d_backward_input_add_amax(gy, N, H, W, Co, ldgy, wp, rows_pad, KH, KW, stride, gx, Ci, ldgx, res, ldres, gy_absmax, nullptr, s);
}

int lhg_conv2d_backward_input_add_amax(const float* gy, int N, int H, int W, int Co, int ldgy, const float* wp, int rows_pad, int KH, int KW, int stride,
                                       float* gx, int Ci, int ldgx, const float* res, int ldres, const float* gy_absmax, float* gx_absmax,
                                       lhg_stream_t s) {
  LHG_REQUIRE(conv_args_ok(KH, KW, stride), "conv2d_backward_input: unsupported kernel %dx%d stride %d", KH, KW, stride);
  LHG_REQUIRE(res == nullptr || ldres >= Ci, "conv2d_backward_input: the added gradient has %d floats per pixel, gx has %d channels", ldres, Ci);
  const int ph = KH / 2, pw = KW / 2;
  const int Ho = (H + 2 * ph - KH) / stride + 1, Wo = (W + 2 * pw - KW) / stride + 1;
  GGParams p{};
  Geom& g = p.g;
  g.N = N; g.Hi = Ho; g.Wi = Wo; g.Ci = Co; g.ldi = ldgy;  // gathered tensor = gy
  g.Ho = H; g.Wo = W; g.Co = Ci; g.ldo = ldgx;             // scattered tensor = gx
  p.in = gy; p.wp = wp; p.out = gx; p.rows_pad = rows_pad; p.act = LHG_ACT_NONE;
  p.res = res; p.ldres = ldres;  // indexed by the gx pixel in every parity class
  p.a_amax = gy_absmax; p.w_amax = weight_amax(wp, KH * KW, rows_pad, Co);
  p.out_amax = gx_absmax;  // max|gx| (with the added gradient) max-accumulated by the epilogue: gx is the next backward GEMM's operand
  if (stride == 1) {
    g.gh = H; g.gw = W; g.oy0 = g.ox0 = 0; g.ostep = 1; g.istep = 1; g.T = KH * KW;
    for (int kh = 0; kh < KH; ++kh)
      for (int kw = 0; kw < KW; ++kw) {
        const int t = kh * KW + kw;
        g.dy[t] = ph - kh; g.dx[t] = pw - kw; g.ws[t] = t;
      }
    g.M = N * H * W;
    return launch_gg(p, as_stream(s));
  }
  // stride 2: one launch per output parity class; input row y receives kernel rows kh with
  // (y + ph - kh) even, from gy row (y + ph - kh)/2.
  Geom cls[4];
  int n = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      g.gh = (H - py + 1) / 2; g.gw = (W - px + 1) / 2;
      if (g.gh <= 0 || g.gw <= 0) continue;
      g.oy0 = py; g.ox0 = px; g.ostep = 2; g.istep = 1;
      int T = 0;
      for (int kh = 0; kh < KH; ++kh) {
        if ((py + ph - kh) & 1) continue;
        for (int kw = 0; kw < KW; ++kw) {
          if ((px + pw - kw) & 1) continue;
          g.dy[T] = (py + ph - kh) / 2; g.dx[T] = (px + pw - kw) / 2; g.ws[T] = kh * KW + kw;
          ++T;
        }
      }
      g.T = T;
      g.M = N * g.gh * g.gw;
      if (T == 0) return fail(LHG_E_ARG, "conv2d_backward_input: empty tap set");
      cls[n++] = g;
    }
  return launch_gg_classes(p, cls, n, as_stream(s));
}

// 3x3 stride-1: S = strips (image x 32-pixel column segments) x chunks per strip, so that the fused nine-tap kernel
// (wg3) and the per-tap kernels can both use it.
static void fused3x3_split(int N, int H, int W, int Ci, int Co, int& segs, int& cps, int& rpc) {
  segs = (W + 31) / 32;
  const long long strips = (long long)N * segs;
  const long long tiles = (long long)(pad64(Ci) / 64) * (pad64(Co) / 64);
  // one balanced round: ~512 workgroups (2 resident per CU), each walking the same number of rows
  long long c = 512 / (tiles * strips);
  c = std::max<long long>(1, std::min<long long>(c, std::max(1, H / 6)));
  rpc = (int)((H + c - 1) / c);
  cps = (H + rpc - 1) / rpc;  // no empty chunks
}

int lhg_conv2d_wgrad_splits(int N, int H, int W, int Ci, int Co, int KH, int KW, int stride) {
  const int Ho = (H + 2 * (KH / 2) - KH) / stride + 1, Wo = (W + 2 * (KW / 2) - KW) / stride + 1;
  if (KH == 3 && KW == 3 && stride == 1 && g_precision == LHG_PRECISION_F32) {
    int segs, cps, rpc;
    fused3x3_split(N, H, W, Ci, Co, segs, cps, rpc);
    return N * segs * cps;
  }
  return pick_splits((long long)N * Ho * Wo, pad64(Ci), pad64(Co), KH * KW);
}

int lhg_conv2d_backward_weight(const float* x, int N, int H, int W, int Ci, int ldx, const float* gy, int Co, int ldgy,
                               int KH, int KW, int stride, float* slabs, int S, int ci_pad, int co_pad, const float* x_absmax,
                               const float* gy_absmax, lhg_stream_t s) {
  LHG_REQUIRE(conv_args_ok(KH, KW, stride), "conv2d_backward_weight: unsupported kernel %dx%d stride %d", KH, KW, stride);
  WGParams p{};
  p.in_amax = x_absmax; p.gout_amax = gy_absmax;
  conv_fwd_geom(p.g, N, H, W, Ci, ldx, Co, ldgy, KH, KW, stride);
  p.in = x; p.gout = gy; p.slabs = slabs; p.m_pad = ci_pad; p.n_pad = co_pad; p.Tslabs = KH * KW;
  if (KH == 3 && KW == 3 && stride == 1 && g_precision == LHG_PRECISION_F32) {
    WG3Params q{};
    fused3x3_split(N, H, W, Ci, Co, q.segs, q.cps, q.rpc);
    if (S == N * q.segs * q.cps && ci_pad == pad64(Ci) && co_pad == pad64(Co)) {
      q.x = x; q.gy = gy; q.slabs = slabs; q.N = N; q.H = H; q.W = W; q.Ci = Ci; q.ldi = ldx; q.Co = Co; q.ldo = ldgy;
      q.m_pad = ci_pad; q.n_pad = co_pad;
      return launch_wg(p, S, as_stream(s), &q);
    }
  }
  return launch_wg(p, S, as_stream(s));
}

// ---- ConvTranspose2d(kernel 2, stride 2): out[2i+py][2j+px][co] = sum_ci in[i][j][ci] * W[ci][co][py][px]
static void convt_geom(Geom& g, int N, int H, int W, int Ci, int ldx, int Co, int ldy, int py, int px) {
  g.N = N; g.Hi = H; g.Wi = W; g.Ci = Ci; g.ldi = ldx;
  g.Ho = 2 * H; g.Wo = 2 * W; g.Co = Co; g.ldo = ldy;
  g.gh = H; g.gw = W; g.oy0 = py; g.ox0 = px; g.ostep = 2; g.istep = 1;
  g.T = 1; g.dy[0] = g.dx[0] = 0; g.ws[0] = py * 2 + px;
  g.M = N * H * W;
}

int lhg_conv_transpose2x2_forward(const float* x, int N, int H, int W, int Ci, int ldx, const float* wp, int rows_pad,
                                  float* y, int Co, int ldy, const float* bias, const float* x_absmax, float* y_absmax, lhg_stream_t s) {
  GGParams p{};
  p.a_amax = x_absmax; p.w_amax = weight_amax(wp, 4, rows_pad, Ci); p.out_amax = y_absmax;
  p.in = x; p.wp = wp; p.out = y; p.bias = bias; p.rows_pad = rows_pad; p.act = LHG_ACT_NONE;
  Geom cls[4];
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) convt_geom(cls[py * 2 + px], N, H, W, Ci, ldx, Co, ldy, py, px);
  return launch_gg_classes(p, cls, 4, as_stream(s));
}

int lhg_conv_transpose2x2_backward_input(const float* gy, int N, int H, int W, int Co, int ldgy, const float* wp, int rows_pad,
                                         float* gx, int Ci, int ldgx, const float* gy_absmax, lhg_stream_t s) {
  return lhg_conv_transpose2x2_backward_input_amax(gy, N, H, W, Co, ldgy, wp, rows_pad, gx, Ci, ldgx, gy_absmax, nullptr, s);
}

int lhg_conv_transpose2x2_backward_input_amax(const float* gy, int N, int H, int W, int Co, int ldgy, const float* wp, int rows_pad,
                                              float* gx, int Ci, int ldgx, const float* gy_absmax, float* gx_absmax, lhg_stream_t s) {
  GGParams p{};
  p.a_amax = gy_absmax; p.w_amax = weight_amax(wp, 4, rows_pad, Co); p.out_amax = gx_absmax;
  Geom& g = p.g;
  g.N = N; g.Hi = 2 * H; g.Wi = 2 * W; g.Ci = Co; g.ldi = ldgy;
  g.Ho = H; g.Wo = W; g.Co = Ci; g.ldo = ldgx;
  g.gh = H; g.gw = W; g.oy0 = g.ox0 = 0; g.ostep = 1; g.istep = 2; g.T = 4;
  for (int t = 0; t < 4; ++t) { g.dy[t] = t >> 1; g.dx[t] = t & 1; g.ws[t] = t; }
  g.M = N * H * W;
  p.in = gy; p.wp = wp; p.out = gx; p.rows_pad = rows_pad; p.act = LHG_ACT_NONE;
  return launch_gg(p, as_stream(s));
}

int lhg_conv_transpose2x2_wgrad_splits(int N, int H, int W, int Ci, int Co) {
  return pick_splits((long long)N * H * W, pad64(Ci), pad64(Co), 1);
}

int lhg_conv_transpose2x2_backward_weight(const float* x, int N, int H, int W, int Ci, int ldx, const float* gy, int Co, int ldgy,
                                          float* slabs, int S, int ci_pad, int co_pad, const float* x_absmax, const float* gy_absmax,
                                          lhg_stream_t s) {
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      WGParams p{};
      p.in_amax = x_absmax; p.gout_amax = gy_absmax;
      convt_geom(p.g, N, H, W, Ci, ldx, Co, ldgy, py, px);
      p.in = x; p.gout = gy; p.slabs = slabs; p.m_pad = ci_pad; p.n_pad = co_pad; p.Tslabs = 4;
      int rc = launch_wg(p, S, as_stream(s));
      if (rc) return rc;
    }
  return LHG_OK;
}

// ---- weight gradient straight into the gradient tensor (ABI 6): tap-fused GEMM (wgrad6.hip) where it applies, the per-tap kernels
// + lhg_wgrad_reduce otherwise.  Workspace: [tickets, 256-byte aligned][partial slabs].
static bool wg6_mode() { return split_f16() && !act_is_bf16(); }

static void wg6_conv_problem(Wg6Problem& q, const float* x, int N, int H, int W, int Ci, int ldx, const float* gy, int Co, int ldgy, int KH, int KW,
                             int stride, const float* x_cmax, const float* gy_cmax) {
  q.strip = x; q.point = gy; q.strip_cmax = x_cmax; q.point_cmax = gy_cmax;
  q.N = N; q.Hs = H; q.Ws = W; q.Cm = Ci; q.lds = ldx;
  q.gh = (H + 2 * (KH / 2) - KH) / stride + 1; q.gw = (W + 2 * (KW / 2) - KW) / stride + 1; q.Cn = Co; q.ldp = ldgy;
  q.krows = KH; q.nt = KW; q.stride = stride; q.dy0 = -(KH / 2); q.dx0 = -(KW / 2);
  q.m_pad = pad64(Ci); q.n_pad = pad64(Co);
}

static void wg6_convt_problem(Wg6Problem& q, const float* x, int N, int H, int W, int Ci, int ldx, const float* gy, int Co, int ldgy,
                              const float* x_cmax, const float* gy_cmax) {
  // dW[ci][co][py][px] = sum_p x[p][ci] gy[2p + (py, px)][co]: the STRIP operand is gy (twice the extent, stride 2), the point operand x
  q.strip = gy; q.point = x; q.strip_cmax = gy_cmax; q.point_cmax = x_cmax;
  q.N = N; q.Hs = 2 * H; q.Ws = 2 * W; q.Cm = Co; q.lds = ldgy;
  q.gh = H; q.gw = W; q.Cn = Ci; q.ldp = ldx;
  q.krows = 2; q.nt = 2; q.stride = 2; q.dy0 = 0; q.dx0 = 0;
  q.m_pad = pad64(Co); q.n_pad = pad64(Ci);
}

static size_t wg6_workspace_bytes(const Wg6Problem& q) {
  const Wg6Plan plan = wg6_plan(q);
  if (plan.variant < 0) return 0;
  return 256 * (((size_t)wg6_tickets(q, plan) * 4 + 255) / 256) + wg6_slab_floats(q, plan) * 4;
}

static int wg6_run(const Wg6Problem& q, float* grad, int accumulate, void* ws, size_t ws_bytes, hipStream_t st, bool& handled) {
  handled = false;
  if (!wg6_mode()) return LHG_OK;
  const Wg6Plan plan = wg6_plan(q);
  if (plan.variant < 0) return LHG_OK;
  handled = true;
  const size_t tk_bytes = 256 * (((size_t)wg6_tickets(q, plan) * 4 + 255) / 256);
  const size_t need = tk_bytes + wg6_slab_floats(q, plan) * 4;
  if (need > ws_bytes) return fail(LHG_E_WORKSPACE, "backward_weight: workspace %zu < %zu bytes", ws_bytes, need);
  LHG_REQUIRE(need == 0 || (reinterpret_cast<uintptr_t>(ws) & 255) == 0, "backward_weight: workspace must be 256-byte aligned");
  unsigned* tickets = tk_bytes ? static_cast<unsigned*>(ws) : nullptr;
  float* slabs = reinterpret_cast<float*>(static_cast<char*>(ws) + tk_bytes);
  if (tk_bytes && hipMemsetAsync(tickets, 0, tk_bytes, st) != hipSuccess) return fail(LHG_E_LAUNCH, "backward_weight: hipMemsetAsync failed");
  const int T = q.krows * q.nt;
  const double flops = 2.0 * q.N * q.gh * (double)q.gw * q.m_pad * q.n_pad * T;
  {
    ScopedKernelTime timed(1, st, flops);
    Geom g{};
    g.M = q.N * q.gh * q.gw; g.Ci = q.Cm; g.Co = q.Cn; g.T = T; g.istep = q.stride; g.ostep = 1; g.Hi = q.Hs;
    timed.tag(g, q.m_pad, q.n_pad, 1000 + plan.variant + 100 * 1000 * plan.S + (plan.fused ? 100 * 1000 * 1000 : 0), flops);
    const int rc = wg6_launch(q, plan, slabs, tickets, grad, accumulate, st);
    if (rc) return rc;
  }
  if (!plan.fused) return wg6_reduce(slabs, plan.S, T, q.m_pad, q.n_pad, grad, q.Cn, q.Cm, accumulate, st);
  return LHG_OK;
}

size_t lhg_conv2d_backward_weight_workspace(int N, int H, int W, int Ci, int Co, int KH, int KW, int stride) {
  if (!conv_args_ok(KH, KW, stride)) return 0;
  const size_t per_tap = (size_t)lhg_conv2d_wgrad_splits(N, H, W, Ci, Co, KH, KW, stride) * KH * KW * pad64(Ci) * pad64(Co) * 4;
  Wg6Problem q{};
  wg6_conv_problem(q, nullptr, N, H, W, Ci, Ci, nullptr, Co, Co, KH, KW, stride, nullptr, nullptr);
  return std::max(per_tap, wg6_mode() ? wg6_workspace_bytes(q) : 0);
}

int lhg_conv2d_backward_weight_into(const float* x, int N, int H, int W, int Ci, int ldx, const float* gy, int Co, int ldgy, int KH, int KW,
                                    int stride, float* grad, int accumulate, void* ws, size_t ws_bytes, const float* x_absmax,
                                    const float* gy_absmax, lhg_stream_t s) {
  LHG_REQUIRE(conv_args_ok(KH, KW, stride), "conv2d_backward_weight: unsupported kernel %dx%d stride %d", KH, KW, stride);
  Wg6Problem q{};
  wg6_conv_problem(q, x, N, H, W, Ci, ldx, gy, Co, ldgy, KH, KW, stride, x_absmax, gy_absmax);
  bool handled = false;
  int rc = wg6_run(q, grad, accumulate, ws, ws_bytes, as_stream(s), handled);
  if (rc || handled) return rc;
  const int S = lhg_conv2d_wgrad_splits(N, H, W, Ci, Co, KH, KW, stride);
  const int ci_pad = pad64(Ci), co_pad = pad64(Co);
  const size_t need = (size_t)S * KH * KW * ci_pad * co_pad * 4;
  if (need > ws_bytes) return fail(LHG_E_WORKSPACE, "conv2d_backward_weight: workspace %zu < %zu bytes", ws_bytes, need);
  rc = lhg_conv2d_backward_weight(x, N, H, W, Ci, ldx, gy, Co, ldgy, KH, KW, stride, static_cast<float*>(ws), S, ci_pad, co_pad, x_absmax, gy_absmax, s);
  if (rc) return rc;
  return lhg_wgrad_reduce(static_cast<float*>(ws), S, KH * KW, ci_pad, co_pad, grad, Co, Ci, 1, accumulate, s);
}

size_t lhg_conv_transpose2x2_backward_weight_workspace(int N, int H, int W, int Ci, int Co) {
  const size_t per_tap = (size_t)lhg_conv_transpose2x2_wgrad_splits(N, H, W, Ci, Co) * 4 * pad64(Ci) * pad64(Co) * 4;
  Wg6Problem q{};
  wg6_convt_problem(q, nullptr, N, H, W, Ci, Ci, nullptr, Co, Co, nullptr, nullptr);
  return std::max(per_tap, wg6_mode() ? wg6_workspace_bytes(q) : 0);
}

int lhg_conv_transpose2x2_backward_weight_into(const float* x, int N, int H, int W, int Ci, int ldx, const float* gy, int Co, int ldgy,
                                               float* grad, int accumulate, void* ws, size_t ws_bytes, const float* x_absmax,
                                               const float* gy_absmax, lhg_stream_t s) {
  Wg6Problem q{};
  wg6_convt_problem(q, x, N, H, W, Ci, ldx, gy, Co, ldgy, x_absmax, gy_absmax);
  bool handled = false;
  int rc = wg6_run(q, grad, accumulate, ws, ws_bytes, as_stream(s), handled);
  if (rc || handled) return rc;
  const int S = lhg_conv_transpose2x2_wgrad_splits(N, H, W, Ci, Co);
  const int ci_pad = pad64(Ci), co_pad = pad64(Co);
  const size_t need = (size_t)S * 4 * ci_pad * co_pad * 4;
  if (need > ws_bytes) return fail(LHG_E_WORKSPACE, "conv_transpose2x2_backward_weight: workspace %zu < %zu bytes", ws_bytes, need);
  rc = lhg_conv_transpose2x2_backward_weight(x, N, H, W, Ci, ldx, gy, Co, ldgy, static_cast<float*>(ws), S, ci_pad, co_pad, x_absmax, gy_absmax, s);
  if (rc) return rc;
  return lhg_wgrad_reduce(static_cast<float*>(ws), S, 4, ci_pad, co_pad, grad, Ci, Co, 0, accumulate, s);
}

int lhg_absmax(const float* x, long long pixels, int C, int ld, float* out, lhg_stream_t s) {
  LHG_REQUIRE(pixels >= 0 && C > 0 && ld >= C, "absmax: bad extents (pixels %lld, C %d, ld %d)", pixels, C, ld);
  LHG_REQUIRE(!act_is_bf16(), "absmax: fp32 tensors only (the bf16 storage mode does not use it)");
  return launch_absmax(x, pixels, C, ld, out, as_stream(s), false);
}

int lhg_wgrad_reduce(const float* slabs, int S, int T, int m_pad, int n_pad, float* grad, int D0, int D1, int m_is_d1, int accumulate,
                     lhg_stream_t s) {
  const size_t total = (size_t)T * D0 * D1;
  if (total * 4 < (size_t)S * 64 || total < 65536) {  // few outputs, many slabs: spend the threads on the slab axis
    const int blocks = (int)std::min<size_t>((total + 15) / 16, 16384);
    hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(blocks), dim3(256), 0, as_stream(s), slabs, S, T, m_pad, n_pad, grad, D0, D1, m_is_d1, accumulate);
  } else {
    const int blocks = (int)std::min<size_t>((total + 63) / 64, 16384);
    hipLaunchKernelGGL(wgrad_reduce_kernel<64>, dim3(blocks), dim3(256), 0, as_stream(s), slabs, S, T, m_pad, n_pad, grad, D0, D1, m_is_d1, accumulate);
  }
  return check_launch("wgrad_reduce");
}

}  // extern "C"
