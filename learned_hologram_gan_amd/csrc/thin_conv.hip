// Thin convolutions for gfx950: 3x3 (pad 1) / 1x1, stride 1, where ONE side of the contraction has at most 8 channels:
//   RGBD -> 64 (UNet encoder1, neural_network_components.py:244), RGB -> 32 (critic block1, discriminator.py:16-19),
//   64 -> 6 (UNet final_layer, neural_network_components.py:288-291), 1024 -> 1 (critic head, discriminator.py:41).
// On the MFMA gather-GEMM these layers are > 90 % zero padding (K or N padded to 32 / 64) and take 60-390 us per launch at
// 384^2 x 4 although they move 20-170 MB; here they are direct fp32 FMA kernels bound by HBM:
//   fan-out : thin input (<= 8 ch)  -> wide output    y[p][cw] = sum_t sum_ct thin[p + s d_t][ct] * W(t, ct, cw)
//   fan-in  : wide input -> thin output (<= 8 ch)      y[p][ct] = sum_t sum_cw wide[p + s d_t][cw] * W(t, ct, cw)
//   wgrad   : dW(t, ct, cw) = sum_p wide[p][cw] * thin[p + s d_t][ct]
// with W(t, ct, cw) = w[ct*st + cw*sw + t] read straight from the OIHW checkpoint tensor, so the same three kernels serve the
// forward conv, its input gradient (s = -1, strides swapped) and both weight-gradient cases.
// Mapping: a lane owns CH float4 chunks of the WIDE channel axis (LPP = wide/(4 CH) lanes per pixel, 64/LPP pixels per wave per
// iteration) and keeps its T x CT x CH x 4 weights (or weight-gradient accumulators) in registers while it walks along an image
// row segment; wide tensors are touched with 16-byte lanes, 64 consecutive lanes = 1 KiB contiguous.  All reductions have a
// fixed order (xor-shuffle trees, then a slab reduce): results are run-to-run identical.
#include <type_traits>

#include "common.h"

namespace lhg {

struct ThinParams {
  const float* thin;   // (N,H,W,ld_t), ct_real channels used
  const float* wide;   // (N,H,W,ld_w), cw channels (multiple of 4)
  float* out;          // fan-out: wide tensor; fan-in: thin tensor; wgrad: partial slabs
  const float* w;      // OIHW weights (fan-out / fan-in)
  const float* bias;   // per OUTPUT channel or null
  const float* scale;  // fan-out epilogue: v = (acc + bias) * scale + shift (eval-mode BN folded in), or null
  const float* shift;
  int thin_vec;        // thin rows are 16-byte aligned with ld >= CT: read them with float4 loads
  int thin_planar;     // fan-out only: `thin` is (N, ct_real, H, W) — the reference's NCHW input read as it is (lhg_conv2d_thin_forward_nchw)
  int planar;          // fan-in: thin output written as (N, ct_real, H, W)
  int N, H, W;
  int ld_t, ld_w, ld_o;
  int ct_real, cw;
  int st, sw;          // W(t, ct, cw) = w[ct*st + cw*sw + t]
  int sign;            // +1: thin/wide operand sampled at p + d_t, -1: at p - d_t
  int lpp;             // lanes per pixel
  int segw;            // pixels of one row handled by one wave
  int act;
  float slope;
  int nwaves;          // wgrad: number of partial slabs
  int wpc;             // fan-out / wgrad: waves of a workgroup along the channel axis (1, 2 or 4)
  float* out_amax;     // fan-out (may be null): max|stored value| is max-accumulated here (LHG_ABSMAX_WORDS zero-filled floats), as the GEMM epilogues' y_absmax
};

template <int T>
__device__ __forceinline__ void tap_offset(int t, int sign, int& dy, int& dx) {
  if (T == 9) {
    dy = sign * (t / 3 - 1);
    dx = sign * (t % 3 - 1);
  } else {
    dy = dx = 0;
  }
}

// CT values of the thin tensor at pixel `pix` (zeros outside the image / beyond ct_real)
template <int CT>
__device__ __forceinline__ void load_thin(const float* __restrict__ base, long long pix, int ld, int ct_real, bool vec, bool ok, float (&v)[CT]) {
#pragma unroll
  for (int c = 0; c < CT; ++c) v[c] = 0.f;
  if (!ok) return;
  const float* q = base + pix * ld;
  if (CT >= 4 && vec) {
#pragma unroll
    for (int g = 0; g < CT / 4; ++g) {
      const f32x4 u = *reinterpret_cast<const f32x4*>(q + 4 * g);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[4 * g + e] = 4 * g + e < ct_real ? u[e] : 0.f;  // select: padding lanes may hold anything
    }
  } else {
#pragma unroll
    for (int c = 0; c < CT; ++c)
      if (c < ct_real) v[c] = q[c];
  }
}

// ------------------------------------------------------------------------------------------------ fan-out
// lane <-> one wide (output) channel.  A workgroup (4 waves) owns one image row segment: it stages the ROWS x (segment + halo)
// strip of the THIN operand in LDS once (zero-filled outside the image, so the inner loop has no boundary logic), then every
// wave walks its share of the segment TP pixels at a time.  The strip values are wave-uniform LDS broadcasts; a lane keeps only
// its T x CT weights and TP accumulators, so occupancy stays high and the only global traffic of the loop is the wide tensor.
// Waves of a workgroup split the work as wpc channel blocks x (4 / wpc) pixel ranges.
constexpr int TP = 8;  // pixels per iteration

extern __shared__ __attribute__((aligned(16))) float thin_lds[];

// LDS strip[r][j][ct] = thin[(y + r - HALO, x_begin - HALO + j)][ct], j < tw; zeros outside the image / beyond ct_real
template <int CT, int HALO>
__device__ __forceinline__ void stage_strip(const ThinParams& p, const float* __restrict__ thin, int row, int y, int x_begin, int tw) {
  constexpr int ROWS = 2 * HALO + 1;
  for (int i = threadIdx.x; i < ROWS * tw; i += 256) {
    const int r = i / tw, j = i - r * tw;
    const int dy = r - HALO, yy = y + dy, xx = x_begin - HALO + j;
    const bool ok = (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
    const float* q = thin + ((long long)(row + (ok ? dy : 0)) * p.W + (ok ? xx : x_begin)) * p.ld_t;  // always a valid pixel
    float v[CT];
    if (p.thin_planar) {  // NCHW: channel planes H W floats apart, consecutive strip pixels are consecutive floats of each plane
      const int n = row / p.H;
      const float* qp = thin + (((long long)n * p.ct_real) * p.H + (ok ? yy : y)) * p.W + (ok ? xx : x_begin);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float u = qp[(long long)(c < p.ct_real ? c : 0) * p.H * p.W];
        v[c] = (ok && c < p.ct_real) ? u : 0.f;
      }
    } else if (CT >= 4 && p.thin_vec) {
#pragma unroll
      for (int g = 0; g < CT / 4; ++g) {
        const f32x4 u = *reinterpret_cast<const f32x4*>(q + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * g + e] = (ok && 4 * g + e < p.ct_real) ? u[e] : 0.f;
      }
    } else {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float u = q[c < p.ct_real ? c : 0];
        v[c] = (ok && c < p.ct_real) ? u : 0.f;
      }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) thin_lds[(size_t)i * CT + c] = v[c];
  }
}

template <int T, int CT>
__global__ __launch_bounds__(256) void thin_fanout_kernel(const ThinParams p, const float* __restrict__ thin, float* __restrict__ out) {
  constexpr int HALO = T == 9 ? 1 : 0, ROWS = 2 * HALO + 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int segs = (p.W + p.segw - 1) / p.segw;
  const int row = blockIdx.x / segs, seg = blockIdx.x - row * segs;
  const int y = row % p.H;
  const int x_begin = seg * p.segw, x_cnt = min(p.segw, p.W - x_begin);
  const int tw = p.segw + 2 * HALO;  // segw is a multiple of TP: the strip covers whole iterations
  stage_strip<CT, HALO>(p, thin, row, y, x_begin, tw);
  __syncthreads();

  const int wc = wave % p.wpc, wp = wave / p.wpc, wpp = 4 / p.wpc;
  // at most 32 wide channels (RGB -> 32, the critic's first layer): the two halves of a wave take the same channels and different pixel
  // ranges — with a lane per channel half of the wave idled through every multiply-add (round 5: 118 -> 108 us at 384^2 x 8, 72 -> 62 at x 4)
  const int halves = p.cw <= 32 ? 2 : 1;
  const int c = halves == 2 ? (lane & 31) : (blockIdx.y * p.wpc + wc) * 64 + lane;
  const bool live = c < p.cw;
  const int cc = live ? c : p.cw - 1;
  float wr[T][CT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wr[t][ct] = ct < p.ct_real ? p.w[ct * p.st + cc * p.sw + (p.sign > 0 ? t : T - 1 - t)] : 0.f;
  const float bias = p.bias ? p.bias[cc] : 0.f, sc = p.scale ? p.scale[cc] : 1.f, sh = p.shift ? p.shift[cc] : 0.f;

  const int parts = wpp * halves;
  const int per = ((x_cnt + parts - 1) / parts + TP - 1) / TP * TP;
  const int xs = (wp * halves + (halves == 2 ? lane >> 5 : 0)) * per, xe = min(x_cnt, xs + per);
  float amax = 0.f;
  for (int x0 = xs; x0 < xe; x0 += TP) {
    float acc[TP];
#pragma unroll
    for (int j = 0; j < TP; ++j) acc[j] = bias;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
      float s[TP + 2 * HALO][CT];
      const float* src = thin_lds + ((size_t)r * tw + x0) * CT;
#pragma unroll
      for (int j = 0; j < TP + 2 * HALO; ++j)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) s[j][ct] = src[j * CT + ct];
#pragma unroll
      for (int kx = 0; kx < ROWS; ++kx)
#pragma unroll
        for (int j = 0; j < TP; ++j)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[j] = fmaf(s[j + kx][ct], wr[r * ROWS + kx][ct], acc[j]);
    }
    if (live) {
      // the activation and "all TP pixels inside the range" are decided ONCE per iteration (wave-uniform), so that the TP elements are
      // straight-line code: with the run-time activation switch and the bounds test per element the epilogue's instructions outnumbered
      // the 3x3 contraction's (round 5 timing ablations on a matrix-pipe variant of this kernel: no stores 98 us, no strip 87, one
      // multiply step instead of eighteen 75 — of 86; a straight-line epilogue: 69).  Same arithmetic, same bits.
      float* o = out + ((long long)row * p.W + x_begin + x0) * p.ld_o + c;
      const bool full = halves == 1 && x0 + TP <= xe;  // (two pixel ranges per wave: the test stays per lane)
      auto emit = [&](auto act_tag, auto full_tag) {
        constexpr int ACT = decltype(act_tag)::value;
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int j = 0; j < TP; ++j) {
          float v = acc[j] * sc + sh;
          if constexpr (ACT == LHG_ACT_RELU) v = v > 0.f ? v : 0.f;
          else if constexpr (ACT == LHG_ACT_LEAKY) v = v > 0.f ? v : v * p.slope;
          else if constexpr (ACT == LHG_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
          if (FULL || x0 + j < xe) {
            o[j * p.ld_o] = v;  // (32-bit offset: TP pixels x ld_o)
            amax = fmaxf(amax, fabsf(v));
          }
        }
      };
      auto by_full = [&](auto act_tag) {
        if (full) emit(act_tag, std::true_type{});
        else emit(act_tag, std::false_type{});
      };
      switch (p.act) {  // wave-uniform
        case LHG_ACT_RELU: by_full(std::integral_constant<int, LHG_ACT_RELU>{}); break;
        case LHG_ACT_LEAKY: by_full(std::integral_constant<int, LHG_ACT_LEAKY>{}); break;
        case LHG_ACT_SIGMOID: by_full(std::integral_constant<int, LHG_ACT_SIGMOID>{}); break;
        default: by_full(std::integral_constant<int, LHG_ACT_NONE>{}); break;
      }
    }
  }
  if (p.out_amax) {  // (uniform) the tensor scale of the GEMM that reads this output, without an lhg_absmax pass over it (round 5: 0.44 ms of a 4K frame)
    unsigned m = __float_as_uint(amax);
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o_, 64));
    unsigned* slot = reinterpret_cast<unsigned*>(p.out_amax);
    if (lane == 0 && m > *reinterpret_cast<volatile unsigned*>(slot)) atomicMax(slot, m);
  }
}

// ------------------------------------------------------------------------------------------------ fan-in
// the value of another lane of the same row of 16 through a DPP control (no LDS traffic)
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

template <int T, int CT, int CH>
__global__ __launch_bounds__(256) void thin_fanin_kernel(const ThinParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int segs = (p.W + p.segw - 1) / p.segw;
  const long long item = (long long)blockIdx.x * 4 + wave;
  if (item >= (long long)p.N * p.H * segs) return;
  const int row = (int)(item / segs), seg = (int)(item - (long long)row * segs);
  const int y = row % p.H;
  const int cl = lane % p.lpp, sub = lane / p.lpp, ppw = 64 / p.lpp;

  float wr[T][CT][CH][4];
#pragma unroll
  for (int ch = 0; ch < CH; ++ch) {
    const int c0 = (ch * p.lpp + cl) * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) wr[t][ct][ch][e] = ct < p.ct_real ? p.w[ct * p.st + (c0 + e) * p.sw + t] : 0.f;
  }

  const int x_end = min(p.W, (seg + 1) * p.segw);
  // every lane of the wave runs the same number of iterations (shuffles below need all lanes): the tail is masked, not skipped.
  // 1x1: the loads of U = 4 pixel groups go out before the first is used (one 16-byte load per lane and iteration left the memory latency
  // exposed: 120 us for 151 MB at 384^2); 3x3 keeps one group per iteration (nine loads each, and the weights fill the registers).
  constexpr int U = T == 1 ? 4 : 1;
  for (int xb0 = seg * p.segw; xb0 < x_end; xb0 += U * ppw) {
    f32x4 vin[U][T][CH];
    if constexpr (U > 1)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int x = xb0 + u * ppw + sub;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        int dy, dx;
        tap_offset<T>(t, p.sign, dy, dx);
        const int yy = y + dy, xx = x + dx;
        const bool ok = x < x_end && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
        const float* q = p.wide + ((long long)(row + dy) * p.W + xx) * p.ld_w;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (ok) v = *reinterpret_cast<const f32x4*>(q + (ch * p.lpp + cl) * 4);
          vin[u][t][ch] = v;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int xb = xb0 + u * ppw;
    if (xb >= x_end) break;  // wave-uniform
    const int x = xb + sub;
    const bool live = x < x_end;
    float acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      int dy, dx;
      tap_offset<T>(t, p.sign, dy, dx);
      const int yy = y + dy, xx = x + dx;
      const bool ok = live && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      const float* q = p.wide + ((long long)(row + dy) * p.W + xx) * p.ld_w;
#pragma unroll
      for (int ch = 0; ch < CH; ++ch) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (U > 1) v = vin[u][t][ch];
        else if (ok) v = *reinterpret_cast<const f32x4*>(q + (ch * p.lpp + cl) * 4);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[ct] = fmaf(v[e], wr[t][ct][ch][e], acc[ct]);
      }
    }
    // sum over the lpp lanes of this pixel: the xor tree over the low lane bits, in fixed order.  Inside a row of 16 lanes the partner
    // values come through DPP (quad permutes for distances 1 and 2; the half-row / row mirrors pair a lane with one of the OTHER group of
    // four / eight, whose lanes all hold that group's sum by then — the same pairs of sums as xor 4 / 8, same bits); only distances 16 and
    // 32 go through the LDS crossbar (ds_bpermute).  Measured: no change of the kernels' time — 32 permutes per four pixels were not what
    // bounds the 64 -> 6 head (120 us for 151 MB at 384^2) — kept because it takes that traffic off the LDS unit.
    if constexpr (CT == 8) {
      if (p.lpp >= 8) {  // (wave-uniform)
        // Reduce-SCATTER over the low three lane bits (round 5): at each stage a lane keeps half of its values — those whose output index
        // has the lane's bit — and hands the other half to its partner, so that after three stages lane (cl & 7) holds output cl & 7
        // summed over its eight lanes; the remaining stages are plain exchanges of that one value.  30 vector operations instead of 64,
        // and — what matters for the 64 -> 6 sigmoid head (1.56 ms of the 4K frame, vector-ALU bound) — every lane finishes ONE output
        // (bias, activation, store) instead of the wave issuing all eight for the pixel's first lane.
        const bool b0 = (cl & 1) != 0, b1 = (cl & 2) != 0, b2 = (cl & 4) != 0;
        float s4[4], s2[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) s4[i] = (b0 ? acc[2 * i + 1] : acc[2 * i]) + dpp_f32<0xB1>(b0 ? acc[2 * i] : acc[2 * i + 1]);  // partner: lane ^ 1
#pragma unroll
        for (int i = 0; i < 2; ++i) s2[i] = (b1 ? s4[2 * i + 1] : s4[2 * i]) + dpp_f32<0x4E>(b1 ? s4[2 * i] : s4[2 * i + 1]);    // partner: lane ^ 2
        float v = (b2 ? s2[1] : s2[0]) + __shfl_xor(b2 ? s2[0] : s2[1], 4, 64);
        for (int m = 8; m < p.lpp; m <<= 1) v += __shfl_xor(v, m, 64);
        const int ct = cl & 7;
        if (live && cl < 8 && ct < p.ct_real) {
          const int n = row / p.H;
          float* o = p.planar ? p.out + (((long long)n * p.ct_real + ct) * p.H + y) * p.W + x : p.out + ((long long)row * p.W + x) * p.ld_o + ct;
          *o = apply_act(v + (p.bias ? p.bias[ct] : 0.f), p.act, p.slope);
        }
        continue;
      }
    }
    if (p.lpp > 1) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] += dpp_f32<0xB1>(acc[ct]);    // quad_perm [1,0,3,2]
    }
    if (p.lpp > 2) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] += dpp_f32<0x4E>(acc[ct]);    // quad_perm [2,3,0,1]
    }
    if (p.lpp > 4) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] += dpp_f32<0x141>(acc[ct]);   // row_half_mirror
    }
    if (p.lpp > 8) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] += dpp_f32<0x140>(acc[ct]);   // row_mirror
    }
    for (int m = 16; m < p.lpp; m <<= 1)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] += __shfl_xor(acc[ct], m, 64);
    if (live && cl == 0) {
      const int n = row / p.H;
      float* o = p.planar ? p.out + (((long long)n * p.ct_real) * p.H + y) * p.W + x : p.out + ((long long)row * p.W + x) * p.ld_o;
      const long long cstride = p.planar ? (long long)p.H * p.W : 1;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        if (ct < p.ct_real) o[ct * cstride] = apply_act(acc[ct] + (p.bias ? p.bias[ct] : 0.f), p.act, p.slope);
    }
    }
  }
}

// ------------------------------------------------------------------------------------------------ wgrad
// Same mapping: lane <-> wide channel, thin strip staged in LDS; a lane accumulates its channel's T x CT gradients over the
// pixels its wave visits, so there is no cross-lane reduction.  Workgroups walk row segments item, item + gridDim.x, ...; the
// waves of a workgroup that share a channel block add their sums through LDS in wave order and the workgroup writes ONE partial
// slab: partial[workgroup][t][ct][cw], summed by thin_wgrad_reduce_kernel.  (Round 4: one slab per wave meant 256 workgroups for the
// 1024-slab budget — one wave per SIMD, every group of TP pixel loads waited for in full: 243 us for 151 MB at 384^2 x 4.  Now up to
// 1024 workgroups, and the next group's loads are in flight while the current one is multiplied.)
template <int T, int CT>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const ThinParams p, const float* __restrict__ thin, const float* __restrict__ wide,
                                                        float* __restrict__ partial) {
  constexpr int HALO = T == 9 ? 1 : 0, ROWS = 2 * HALO + 1;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int segs = (p.W + p.segw - 1) / p.segw;
  const int items = p.N * p.H * segs;
  const int tw = p.segw + 2 * HALO;
  const int wc = wave % p.wpc, wp = wave / p.wpc, wpp = 4 / p.wpc;
  const int c = (blockIdx.y * p.wpc + wc) * 64 + lane;
  const bool live = c < p.cw;
  const int cc = live ? c : p.cw - 1;

  float acc[T][CT];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[t][ct] = 0.f;

  for (int item = blockIdx.x; item < items; item += gridDim.x) {
    const int row = item / segs, seg = item - row * segs;
    const int y = row % p.H;
    const int x_begin = seg * p.segw, x_cnt = min(p.segw, p.W - x_begin);
    const int per = ((x_cnt + wpp - 1) / wpp + TP - 1) / TP * TP;
    const int xs = wp * per, xe = min(x_cnt, xs + per);
    const float* q0 = wide + ((long long)row * p.W + x_begin) * p.ld_w + cc;
    auto load_group = [&](int x0, float (&wv)[TP]) {
#pragma unroll
      for (int j = 0; j < TP; ++j) wv[j] = x0 + j < xe ? q0[(long long)(x0 + j) * p.ld_w] : 0.f;
    };
    float wv[TP], wn[TP];
    load_group(xs, wv);  // goes out before the strip is staged
    __syncthreads();  // previous strip fully consumed
    stage_strip<CT, HALO>(p, thin, row, y, x_begin, tw);
    __syncthreads();
    for (int x0 = xs; x0 < xe; x0 += TP) {
      load_group(x0 + TP, wn);
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        float s[TP + 2 * HALO][CT];
        const float* src = thin_lds + ((size_t)r * tw + x0) * CT;
#pragma unroll
        for (int j = 0; j < TP + 2 * HALO; ++j)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) s[j][ct] = src[j * CT + ct];
#pragma unroll
        for (int kx = 0; kx < ROWS; ++kx)
#pragma unroll
          for (int j = 0; j < TP; ++j)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) acc[r * ROWS + kx][ct] = fmaf(s[j + kx][ct], wv[j], acc[r * ROWS + kx][ct]);
      }
#pragma unroll
      for (int j = 0; j < TP; ++j) wv[j] = wn[j];
    }
  }
  // the wpp waves of a channel block: wave 0 stores, waves 1.. add in order (fixed summation order), then the block's slab is written
  __syncthreads();  // the last strip is no longer read
  float* red = thin_lds + (size_t)wc * T * CT * 64;
  for (int w = 0; w < wpp; ++w) {
    if (wp == w) {
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          float* r = red + (t * CT + ct) * 64 + lane;
          *r = w == 0 ? acc[t][ct] : *r + acc[t][ct];
        }
    }
    __syncthreads();
  }
  if (live && wp == 0) {
    // sign < 0: the strip was sampled at p + d_t' while the gradient tap is t = T-1-t' (d_{T-1-t} = -d_t)
    float* slab = partial + (size_t)blockIdx.x * T * CT * p.cw + c;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) slab[((size_t)(p.sign > 0 ? t : T - 1 - t) * CT + ct) * p.cw] = red[(t * CT + ct) * 64 + lane];
  }
}

// gw[ct*st + cw*sw + t] = sum_slab partial[slab][t][ct][cw] (double accumulation, fixed order).  grid ceil(J/32), block 32 x 32.
__global__ __launch_bounds__(1024) void thin_wgrad_reduce_kernel(const float* __restrict__ partial, int nwaves, int T, int CT, int cw,
                                                                   int ct_real, int st, int sw, float* __restrict__ gw) {
  __shared__ double red[32][32];
  const int jl = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int J = T * CT * cw;
  const int j = blockIdx.x * 32 + jl;
  double s = 0;
  if (j < J)
    for (int b = slice; b < nwaves; b += 32) s += partial[(size_t)b * J + j];
  red[slice][jl] = s;
  __syncthreads();
  if (slice == 0 && j < J) {
    double a = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) a += red[i][jl];
    const int t = j / (CT * cw), r = j - t * CT * cw;
    const int ct = r / cw, c = r - ct * cw;
    if (ct < ct_real) gw[ct * st + c * sw + t] = (float)a;
  }
}

// ------------------------------------------------------------------------------------------------ host side
enum { K_FANOUT = 0, K_FANIN = 1, K_WGRAD = 2 };

static int pick_segw(int rows, int W, int step) {
  // enough waves to cover the chip (>= ~4096) but at least a few iterations each to amortise the register-resident weights
  int segw = W;
  while (segw > 4 * step && (long long)rows * ((W + segw - 1) / segw) < 4096) segw = (segw + 1) / 2;
  return (segw + step - 1) / step * step;
}

template <int T, int CT>
static int launch_lane_kernel(int kind, ThinParams& p, hipStream_t st) {
  constexpr int HALO = T == 9 ? 1 : 0;
  const int rows = p.N * p.H;
  const int cblocks = (p.cw + 63) / 64;
  p.wpc = cblocks >= 4 ? 4 : (cblocks >= 2 ? 2 : 1);
  // one workgroup per row segment; segments as long as LDS comfortably allows (<= 512 pixels), a multiple of TP
  int segw = (p.W + TP - 1) / TP * TP;
  if (segw > 512) segw = 512;
  while (segw > 4 * TP && (long long)rows * ((p.W + segw - 1) / segw) * ((cblocks + p.wpc - 1) / p.wpc) < 1024) segw = (segw / 2 + TP - 1) / TP * TP;
  p.segw = segw;
  const int items = rows * ((p.W + segw - 1) / segw);
  const unsigned gy = (unsigned)((cblocks + p.wpc - 1) / p.wpc);
  const size_t lds = (size_t)(2 * HALO + 1) * (segw + 2 * HALO) * CT * sizeof(float);
  if (kind == K_WGRAD) {
    int blocks = p.nwaves;  // nwaves = slab budget; one slab per workgroup
    if (blocks > items) blocks = items;
    if (blocks < 1) blocks = 1;
    p.nwaves = blocks;
    const size_t lds_red = (size_t)p.wpc * T * CT * 64 * sizeof(float);  // the waves' sums meet in the strip's LDS
    hipLaunchKernelGGL((thin_wgrad_kernel<T, CT>), dim3(blocks, gy), dim3(256), std::max(lds, lds_red), st, p, p.thin, p.wide, p.out);
    return check_launch("thin_wgrad");
  }
  hipLaunchKernelGGL((thin_fanout_kernel<T, CT>), dim3(items, gy), dim3(256), lds, st, p, p.thin, p.out);
  return check_launch("thin_fanout");
}

template <int T, int CT, int CH>
static int launch_fanin(ThinParams& p, hipStream_t st) {
  const int ppw = 64 / p.lpp;
  p.segw = pick_segw(p.N * p.H, p.W, ppw);
  const long long items = (long long)p.N * p.H * ((p.W + p.segw - 1) / p.segw);
  hipLaunchKernelGGL((thin_fanin_kernel<T, CT, CH>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, p);
  return check_launch("thin_fanin");
}

// thin side rounded up to 1 / 4 / 8 channels.  fan-out / wgrad: one lane per wide channel.  fan-in: the wide side as 16-byte
// chunks, <= 64 chunks -> one per lane, 256 chunks -> 4 per lane.
static int dispatch_thin(int kind, int k, ThinParams& p, hipStream_t st) {
  const int T = k * k;
  LHG_REQUIRE(k == 1 || k == 3, "thin conv: kernel %dx%d unsupported (1x1 and 3x3 only)", k, k);
  LHG_REQUIRE(p.ct_real >= 1 && p.ct_real <= 8, "thin conv: thin side has %d channels (1..8)", p.ct_real);
  const int CT = p.ct_real == 1 ? 1 : (p.ct_real <= 4 ? 4 : 8);
  if (kind != K_FANIN) {
    p.thin_vec = CT >= 4 && p.ld_t % 4 == 0 && p.ld_t >= CT && (reinterpret_cast<uintptr_t>(p.thin) & 15) == 0;
#define LHG_THIN_CASE(t, ct) \
  if (T == t && CT == ct) return launch_lane_kernel<t, ct>(kind, p, st)
    LHG_THIN_CASE(9, 4);
    LHG_THIN_CASE(1, 4);
    LHG_THIN_CASE(9, 1);
    LHG_THIN_CASE(1, 1);
    LHG_THIN_CASE(1, 8);
#undef LHG_THIN_CASE
    return fail(LHG_E_ARG, "thin conv: no kernel for %dx%d taps with %d thin channels", k, k, p.ct_real);
  }
  LHG_REQUIRE(p.cw % 4 == 0 && p.ld_w % 4 == 0 && (reinterpret_cast<uintptr_t>(p.wide) & 15) == 0, "thin conv: wide tensor must be 16-byte aligned with C %% 4 == 0 (C %d, ld %d)", p.cw, p.ld_w);
  const int chunks = p.cw / 4;
  int CH = 1;
  if (chunks > 64) {
    LHG_REQUIRE(chunks == 256, "thin conv: wide side of %d channels unsupported (<= 256 or 1024)", p.cw);
    CH = 4;
    p.lpp = 64;
  } else {
    LHG_REQUIRE((chunks & (chunks - 1)) == 0, "thin conv: wide side must have 4 * 2^n channels (got %d)", p.cw);
    p.lpp = chunks;
  }
#define LHG_THIN_CASE(t, ct, ch) \
  if (T == t && CT == ct && CH == ch) return launch_fanin<t, ct, ch>(p, st)
  LHG_THIN_CASE(9, 4, 1);
  LHG_THIN_CASE(1, 4, 1);
  LHG_THIN_CASE(9, 1, 1);
  LHG_THIN_CASE(1, 1, 1);
  LHG_THIN_CASE(9, 1, 4);
  LHG_THIN_CASE(1, 8, 1);
#undef LHG_THIN_CASE
  return fail(LHG_E_ARG, "thin conv: no kernel for %dx%d taps, %d thin / %d wide channels", k, k, p.ct_real, p.cw);
}

// upper bound of the number of partial slabs (the launcher lowers it to the number of row segments): 1024 waves, <= 16 MiB
static int wgrad_waves(int J) {
  long long nw = 1024;
  const long long cap = (4ll << 20) / J;
  if (nw > cap) nw = cap;
  return (int)(nw < 1 ? 1 : nw);
}

}  // namespace lhg

using namespace lhg;

extern "C" {

int lhg_conv2d_thin_supported(int Ci, int Co, int k, int stride) {
  if (stride != 1 || (k != 1 && k != 3)) return 0;
  auto wide_ok = [](int c) {
    if (c % 4) return false;
    const int chunks = c / 4;
    return chunks == 256 || (chunks <= 64 && (chunks & (chunks - 1)) == 0);
  };
  auto combo_ok = [&](int thin, int wide) {
    if (thin < 1 || thin > 8 || !wide_ok(wide)) return false;
    const int CT = thin == 1 ? 1 : (thin <= 4 ? 4 : 8);
    const int CH = wide / 4 > 64 ? 4 : 1;
    if (CH == 4) return k == 3 && CT == 1;
    if (CT == 8) return k == 1;
    return true;
  };
  if (Ci <= 4 && combo_ok(Ci, Co)) return 1;   // thin input
  if (Co <= 8 && combo_ok(Co, Ci)) return 2;   // thin output
  return 0;
}

int lhg_conv2d_thin_forward(const float* x, int N, int H, int W, int Ci, int ldx, const float* w, int Co, int k, float* y, int ldy,
                            const float* bias, const float* scale, const float* shift, int act, float slope, int planar_out,
                            lhg_stream_t s) {
  return lhg_conv2d_thin_forward_amax(x, N, H, W, Ci, ldx, w, Co, k, y, ldy, bias, scale, shift, act, slope, planar_out, nullptr, s);
}

int lhg_conv2d_thin_forward_amax(const float* x, int N, int H, int W, int Ci, int ldx, const float* w, int Co, int k, float* y, int ldy,
                                 const float* bias, const float* scale, const float* shift, int act, float slope, int planar_out,
                                 float* y_absmax, lhg_stream_t s) {
  const int mode = lhg_conv2d_thin_supported(Ci, Co, k, 1);
  LHG_REQUIRE(y_absmax == nullptr || mode == 1, "conv2d_thin_forward: max|y| is measured by the thin-INPUT kernels only (a wide output that feeds a GEMM)");
  LHG_REQUIRE(!act_is_bf16(), "conv2d_thin_forward: thin convolutions take fp32 tensors (bf16 storage routes every layer to the MFMA path)");
  LHG_REQUIRE(mode != 0, "conv2d_thin_forward: %d -> %d channels, %dx%d is not a thin convolution", Ci, Co, k, k);
  LHG_REQUIRE(N > 0 && H > 0 && W > 0 && ldx >= Ci && (planar_out || ldy >= Co), "conv2d_thin_forward: bad extents");
  LHG_REQUIRE(!(planar_out && mode == 1), "conv2d_thin_forward: planar output only for thin outputs");
  LHG_REQUIRE(!((scale || shift) && mode == 2), "conv2d_thin_forward: scale/shift epilogue only for thin inputs");
  ThinParams p{};
  p.N = N; p.H = H; p.W = W; p.w = w; p.bias = bias; p.scale = scale; p.shift = shift; p.act = act; p.slope = slope; p.sign = 1;
  p.planar = planar_out;
  const int T = k * k;
  if (mode == 1) {  // thin = x (ci), wide = y (co): W(t, ci, co) = w[co*Ci*T + ci*T + t]
    p.thin = x; p.ld_t = ldx; p.ct_real = Ci; p.out = y; p.ld_o = ldy; p.cw = Co; p.wide = y; p.ld_w = ldy;
    p.st = T; p.sw = Ci * T; p.out_amax = y_absmax;
    return dispatch_thin(K_FANOUT, k, p, as_stream(s));
  }
  // thin = y (co), wide = x (ci): W(t, co, ci) = w[co*Ci*T + ci*T + t]
  p.wide = x; p.ld_w = ldx; p.cw = Ci; p.out = y; p.ld_o = ldy; p.ct_real = Co; p.thin = nullptr; p.ld_t = 0;
  p.st = Ci * T; p.sw = T;
  return dispatch_thin(K_FANIN, k, p, as_stream(s));
}

int lhg_conv2d_thin_forward_nchw(const float* x_nchw, int N, int H, int W, int Ci, const float* w, int Co, int k, float* y, int ldy,
                                 const float* bias, const float* scale, const float* shift, int act, float slope, float* y_absmax, lhg_stream_t s) {
  LHG_REQUIRE(lhg_conv2d_thin_supported(Ci, Co, k, 1) == 1, "conv2d_thin_forward_nchw: %d -> %d channels, %dx%d is not a thin-INPUT convolution", Ci, Co, k, k);
  LHG_REQUIRE(!act_is_bf16(), "conv2d_thin_forward_nchw: thin convolutions take fp32 tensors");
  LHG_REQUIRE(N > 0 && H > 0 && W > 0 && ldy >= Co, "conv2d_thin_forward_nchw: bad extents");
  ThinParams p{};
  p.N = N; p.H = H; p.W = W; p.w = w; p.bias = bias; p.scale = scale; p.shift = shift; p.act = act; p.slope = slope; p.sign = 1;
  const int T = k * k;
  p.thin = x_nchw; p.ld_t = Ci; p.thin_planar = 1; p.ct_real = Ci; p.out = y; p.ld_o = ldy; p.cw = Co; p.wide = y; p.ld_w = ldy;
  p.st = T; p.sw = Ci * T; p.out_amax = y_absmax;
  return dispatch_thin(K_FANOUT, k, p, as_stream(s));
}

int lhg_conv2d_thin_backward_input(const float* gy, int N, int H, int W, int Co, int ldg, const float* w, int Ci, int k, float* gx,
                                   int ldgx, lhg_stream_t s) {
  const int mode = lhg_conv2d_thin_supported(Ci, Co, k, 1);
  LHG_REQUIRE(!act_is_bf16(), "conv2d_thin_backward_input: thin convolutions take fp32 tensors (bf16 storage routes every layer to the MFMA path)");
  LHG_REQUIRE(mode != 0, "conv2d_thin_backward_input: %d -> %d channels, %dx%d is not a thin convolution", Ci, Co, k, k);
  LHG_REQUIRE(N > 0 && H > 0 && W > 0 && ldg >= Co && ldgx >= Ci, "conv2d_thin_backward_input: bad extents");
  ThinParams p{};
  p.N = N; p.H = H; p.W = W; p.w = w; p.bias = nullptr; p.act = LHG_ACT_NONE; p.sign = -1;
  const int T = k * k;
  if (mode == 1) {  // output gx thin (ci), input gy wide (co): gx[q][ci] = sum gy[q - d_t][co] w[co][ci][t]
    p.wide = gy; p.ld_w = ldg; p.cw = Co; p.out = gx; p.ld_o = ldgx; p.ct_real = Ci;
    p.st = T; p.sw = Ci * T;
    return dispatch_thin(K_FANIN, k, p, as_stream(s));
  }
  // input gy thin (co), output gx wide (ci)
  p.thin = gy; p.ld_t = ldg; p.ct_real = Co; p.out = gx; p.ld_o = ldgx; p.cw = Ci; p.wide = gx; p.ld_w = ldgx;
  p.st = Ci * T; p.sw = T;
  return dispatch_thin(K_FANOUT, k, p, as_stream(s));
}

size_t lhg_conv2d_thin_wgrad_workspace(int N, int H, int W, int Ci, int Co, int k) {
  const int mode = lhg_conv2d_thin_supported(Ci, Co, k, 1);
  if (!mode) return 0;
  const int thin = mode == 1 ? Ci : Co, wide = mode == 1 ? Co : Ci;
  const int CT = thin == 1 ? 1 : (thin <= 4 ? 4 : 8);
  const int J = k * k * CT * wide;
  (void)N; (void)H; (void)W;
  return (size_t)wgrad_waves(J) * J * sizeof(float);
}

int lhg_conv2d_thin_backward_weight(const float* x, int N, int H, int W, int Ci, int ldx, const float* gy, int Co, int ldg, int k,
                                    float* gw, float* ws, size_t ws_bytes, lhg_stream_t s) {
  const int mode = lhg_conv2d_thin_supported(Ci, Co, k, 1);
  LHG_REQUIRE(!act_is_bf16(), "conv2d_thin_backward_weight: thin convolutions take fp32 tensors (bf16 storage routes every layer to the MFMA path)");
  LHG_REQUIRE(mode != 0, "conv2d_thin_backward_weight: %d -> %d channels, %dx%d is not a thin convolution", Ci, Co, k, k);
  LHG_REQUIRE(N > 0 && H > 0 && W > 0 && ldx >= Ci && ldg >= Co, "conv2d_thin_backward_weight: bad extents");
  const size_t need = lhg_conv2d_thin_wgrad_workspace(N, H, W, Ci, Co, k);
  if (ws_bytes < need) return fail(LHG_E_WORKSPACE, "conv2d_thin_backward_weight: workspace %zu < %zu", ws_bytes, need);
  ThinParams p{};
  p.N = N; p.H = H; p.W = W; p.out = ws;
  const int T = k * k;
  if (mode == 1) {  // thin = x at p + d_t, wide = gy at p: gw[co][ci][t]
    p.thin = x; p.ld_t = ldx; p.ct_real = Ci; p.wide = gy; p.ld_w = ldg; p.cw = Co; p.sign = 1;
    p.st = T; p.sw = Ci * T;
  } else {          // wide = x at q, thin = gy at q - d_t
    p.thin = gy; p.ld_t = ldg; p.ct_real = Co; p.wide = x; p.ld_w = ldx; p.cw = Ci; p.sign = -1;
    p.st = Ci * T; p.sw = T;
  }
  const int CT = p.ct_real == 1 ? 1 : (p.ct_real <= 4 ? 4 : 8);
  const int J = T * CT * p.cw;
  p.nwaves = wgrad_waves(J);
  int rc = dispatch_thin(K_WGRAD, k, p, as_stream(s));
  if (rc) return rc;
  hipLaunchKernelGGL(thin_wgrad_reduce_kernel, dim3((J + 31) / 32), dim3(1024), 0, as_stream(s), ws, p.nwaves, T, CT, p.cw, p.ct_real,
                     p.st, p.sw, gw);
  return check_launch("thin_wgrad_reduce");
}

}  // extern "C"
