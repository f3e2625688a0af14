// Tap-fused weight-gradient GEMM of the fp16-split mode (wg6_kernel.inc) — its own translation unit: instantiation table, the
// deterministic (variant, split count) plan, the launcher and the slab reduction.  Called from conv_engine.hip (wg6_api.h).
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>

#include "wg6_api.h"
#include "wg6_support.h"

namespace lhg {

#include "wg6_kernel.inc"

// grad[n][m][tap] (+)= sum_s slabs[s][tap][m][n], s ascending.  One thread per (tap, m, four consecutive n): 16-byte slab loads, EIGHT
// splits in flight per thread (the pass is latency-bound: a launch reads S x dW bytes that the GEMM has just written, mostly from
// L2 / Infinity Cache), four 4-byte stores (tap-strided rows of the OIHW / IOHW gradient: L2 merges them).
__global__ __launch_bounds__(256) void wg6_reduce_kernel(const float* __restrict__ slabs, int S, int T, int m_pad, int n_pad, float* __restrict__ grad,
                                                         int Cn, int Cm, int accumulate) {
  const int n4 = n_pad / 4;
  const long long total = (long long)T * Cm * n4;
  const size_t slab_stride = (size_t)T * m_pad * n_pad;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % n4);
    const int m = (int)((i / n4) % Cm);
    const int tap = (int)(i / ((long long)n4 * Cm));
    const float* src = slabs + ((size_t)tap * m_pad + m) * n_pad + c4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(src);
    int s = 1;
    for (; s + 8 <= S; s += 8) {
      f32x4 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(s + u) * slab_stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) v += q[u];   // split order: the same sum as the in-launch reduction
    }
    for (; s < S; ++s) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * slab_stride);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = c4 * 4 + e;
      if (n < Cn) {
        float* dst = grad + ((size_t)n * Cm + m) * T + tap;
        *dst = accumulate ? *dst + v[e] : v[e];
      }
    }
  }
}

// The same sum for many splits and few outputs (64 -> 64 @ 384^2: S = 256 slabs of 9 x 64 x 64, i.e. 36 workgroups of the kernel above walking
// 256 loads each): the four waves of a workgroup take the four contiguous quarters of the split axis for the SAME 64 outputs and wave 0 adds
// the quarters in order, ((q0 + q1) + q2) + q3 — a fixed order, used for S >= 32 only (below that the kernel above, whose order the in-launch
// reduction shares).
__global__ __launch_bounds__(256) void wg6_reduce4_kernel(const float* __restrict__ slabs, int S, int T, int m_pad, int n_pad, float* __restrict__ grad,
                                                          int Cn, int Cm, int accumulate) {
  __shared__ f32x4 part[3][64];
  const int n4 = n_pad / 4;
  const long long total = (long long)T * Cm * n4;
  const size_t slab_stride = (size_t)T * m_pad * n_pad;
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + o;
  const bool live = i < total;
  const long long ii = live ? i : total - 1;
  const int c4 = (int)(ii % n4);
  const int m = (int)((ii / n4) % Cm);
  const int tap = (int)(ii / ((long long)n4 * Cm));
  const int chunk = (S + 3) / 4, s0 = q * chunk, s1 = min(S, s0 + chunk);
  const float* src = slabs + ((size_t)tap * m_pad + m) * n_pad + c4 * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (s0 < s1) {
    v = *reinterpret_cast<const f32x4*>(src + (size_t)s0 * slab_stride);
    int s = s0 + 1;
    for (; s + 8 <= s1; s += 8) {
      f32x4 r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(s + u) * slab_stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) v += r[u];
    }
    for (; s < s1; ++s) v += *reinterpret_cast<const f32x4*>(src + (size_t)s * slab_stride);
  }
  if (q > 0) part[q - 1][o] = v;
  __syncthreads();
  if (q == 0 && live) {
    const int filled = (S + chunk - 1) / chunk;  // quarters that hold splits (S >= 32: all four)
    for (int k = 1; k < filled; ++k) v += part[k - 1][o];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = c4 * 4 + e;
      if (n < Cn) {
        float* dst = grad + ((size_t)n * Cm + m) * T + tap;
        *dst = accumulate ? *dst + v[e] : v[e];
      }
    }
  }
}

namespace {

struct Variant {
  const char* name;
  int BM, BN, CWM, CWN, NT, NTY, STR;
  int occ;      // workgroups a CU holds (registers / LDS)
  float eff;    // matrix-pipe efficiency of the model (measured per layer with tools/wg6_sweep.py)
  void (*kernel)(const WG6Params);
};

#define LHG_WG6(BM, BN, CWM, CWN, NT, NTY, STR, OCC, EFF) \
  {#BM "x" #BN " c" #CWM "x" #CWN " nt" #NT " ny" #NTY " s" #STR, BM, BN, CWM, CWN, NT, NTY, STR, OCC, EFF, wg6_kernel<BM, BN, CWM, CWN, NT, NTY, STR>}

// occ / eff: resident workgroups per CU and whole-launch matrix-pipe efficiency (2.1 GHz, reduce launch included) measured per layer of
// the 384^2 batch-4 step on MI355X with tools/wg6_sweep.py (profiles/r04_wg6_sweep.jsonl); the plan below only RANKS variants with them.
const Variant kVariants[] = {
    // 3x3 stride 1
    LHG_WG6(128, 128, 4, 2, 3, 1, 1, 1, 0.56f),
    LHG_WG6(128, 64, 4, 1, 3, 1, 1, 1, 0.50f),
    LHG_WG6(64, 128, 2, 2, 3, 1, 1, 1, 0.50f),
    LHG_WG6(64, 64, 2, 2, 3, 1, 1, 2, 0.44f),
    LHG_WG6(64, 64, 2, 2, 3, 3, 1, 1, 0.47f),
    // 3x3 stride 2
    LHG_WG6(128, 128, 4, 2, 3, 1, 2, 1, 0.47f),
    LHG_WG6(128, 64, 4, 1, 3, 1, 2, 1, 0.41f),
    LHG_WG6(64, 128, 2, 2, 3, 1, 2, 1, 0.45f),
    LHG_WG6(64, 64, 2, 2, 3, 1, 2, 2, 0.36f),
    // 1x1 (HBM-bound shapes: two workgroups per CU where the tile allows it)
    LHG_WG6(128, 128, 2, 2, 1, 1, 1, 1, 0.22f),
    LHG_WG6(128, 64, 2, 2, 1, 1, 1, 2, 0.23f),
    LHG_WG6(64, 128, 2, 2, 1, 1, 1, 2, 0.23f),
    LHG_WG6(64, 64, 2, 2, 1, 1, 1, 2, 0.20f),
    // 2x2 transposed conv (strip = gy at twice the extent)
    LHG_WG6(128, 128, 4, 2, 2, 1, 2, 1, 0.45f),
    LHG_WG6(128, 64, 4, 1, 2, 1, 2, 1, 0.42f),
    LHG_WG6(64, 128, 2, 2, 2, 1, 2, 1, 0.42f),
    LHG_WG6(64, 64, 2, 2, 2, 1, 2, 2, 0.36f),
};
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

int g_force_variant = -1, g_force_splits = -1, g_force_fused = -1;
int g_last_plan[3] = {-1, 0, 0};  // (variant, splits, fused) of the last launch: lhg_wg6_last_plan

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

bool variant_fits(const Variant& v, const Wg6Problem& q) {
  return v.NT == q.nt && v.STR == q.stride && q.krows % v.NTY == 0 && q.m_pad % v.BM == 0 && q.n_pad % v.BN == 0;
}

long long steps_total(const Wg6Problem& q) { return ((long long)q.N * q.gh * (q.gw + 2) + 31) / 32; }

// one round of the chip per split where the tiles allow it; the model prices a workgroup-step by its MFMAs per SIMD
int splits_for(const Variant& v, const Wg6Problem& q) {
  const long long tiles = (long long)(q.m_pad / v.BM) * (q.n_pad / v.BN) * (q.krows / v.NTY);
  const long long slots = 256ll * v.occ;
  const long long steps = steps_total(q);
  long long S = tiles >= slots ? 1 : slots / tiles;
  S = std::min(S, std::max(1ll, steps / 8));  // at least eight steps per workgroup
  return (int)std::max(1ll, std::min(S, 4096ll));
}

double model_cycles(const Variant& v, const Wg6Problem& q, int S) {
  const long long tiles = (long long)(q.m_pad / v.BM) * (q.n_pad / v.BN) * (q.krows / v.NTY);
  const long long slots = 256ll * v.occ;
  const long long rounds = (tiles * S + slots - 1) / slots;
  const long long steps = (steps_total(q) + S - 1) / S;
  const int TM = v.BM / v.CWM / 32, TN = v.BN / v.CWN / 32;
  const double mfma_per_wave_step = 2.0 * v.NT * v.NTY * TM * TN * 3;
  const double cyc_step = mfma_per_wave_step * 32.0 * (v.CWM * v.CWN / 4.0) * v.occ / v.eff;  // all resident workgroups of a CU advance one step
  const double slab_bytes = (double)S * q.krows * q.nt * q.m_pad * q.n_pad * 4.0;
  const double reduce_cycles = S > 1 ? 2.0 * slab_bytes / 1500.0 + 6000.0 : 0.0;  // written + read back at ~3 TB/s of a 2 GHz clock, plus a launch
  const double per_wg = 14000.0;  // prologue (scales, first tiles) + epilogue (slab stores) of one workgroup round
  return (double)rounds * (steps * cyc_step + per_wg) + 4000.0 + reduce_cycles;
}

unsigned magic_for(int d) { return (unsigned)(((1ull << 32) + (unsigned)d - 1) / (unsigned)d); }

}  // namespace

int wg6_variant_count() { return kNumVariants; }
const char* wg6_variant_name(int v) { return v >= 0 && v < kNumVariants ? kVariants[v].name : "?"; }

Wg6Plan wg6_plan(const Wg6Problem& q) {
  Wg6Plan plan{-1, 1, 0};
  static const int enabled = env_int("LHG_WG6", 1);
  if (!enabled) return plan;
  // extents the kernel's 32-bit arithmetic covers
  const long long total = (long long)q.N * q.gh * (q.gw + 2);
  const unsigned long long s_bytes = (((unsigned long long)q.N * q.Hs * q.Ws - 1) * q.lds + q.Cm) * 4ull;
  const unsigned long long p_bytes = (((unsigned long long)q.N * q.gh * q.gw - 1) * q.ldp + q.Cn) * 4ull;
  const unsigned long long lim = (1ull << 32) - (1ull << 20);
  if (total <= 0 || (total + 64) * (q.gw + 1) >= (1ll << 32) || ((total + 64) / (q.gw + 2) + 2) * (long long)q.gh >= (1ll << 32)) return plan;
  if (s_bytes >= lim || p_bytes >= lim) return plan;
  if (q.lds % 4 || q.ldp % 4 || q.Cm % 4 || q.Cn % 4 || q.m_pad % 64 || q.n_pad % 64 || q.m_pad < q.Cm || q.n_pad < q.Cn) return plan;
  if ((reinterpret_cast<uintptr_t>(q.strip) & 15) || (reinterpret_cast<uintptr_t>(q.point) & 15)) return plan;
  static const int env_variant = env_int("LHG_WG6_VARIANT", -1), env_splits = env_int("LHG_WG6_SPLITS", -1), env_fused = env_int("LHG_WG6_FUSED", -1);
  const int fv = g_force_variant >= 0 ? g_force_variant : env_variant;
  const int fs = g_force_splits >= 1 ? g_force_splits : env_splits;
  const int ff = g_force_fused >= 0 ? g_force_fused : env_fused;
  double best = 1e300;
  for (int v = 0; v < kNumVariants; ++v) {
    if (!variant_fits(kVariants[v], q)) continue;
    if (fv >= 0 && fv != v) continue;
    const int S = fs >= 1 ? fs : splits_for(kVariants[v], q);
    const double c = model_cycles(kVariants[v], q, S);
    if (c < best) { best = c; plan.variant = v; plan.S = S; }
  }
  if (plan.variant < 0) return plan;
  const long long steps = steps_total(q);
  if (plan.S > steps) plan.S = (int)steps;
  // Reduction form.  Unsplit: the workgroup's slab is final and it moves it into the gradient's layout itself (no second launch).  Split:
  // the in-launch form (the last workgroup of a tile to arrive sums the S slabs) measured SLOWER than a reduce launch on every layer of
  // the step (tools/wg6_sweep.py, profiles/r04_wg6_sweep.jsonl: 512 -> 1024 stride 2 at S = 2: 334 against 314 us; 32 -> 64 stride 2 at
  // S = 42: 154 against 104 us) — one workgroup per tile reads what the whole chip reads in the separate launch — so it is kept for
  // A/B runs only (LHG_WG6_FUSED=1).
  plan.fused = ff >= 0 ? ff : (plan.S == 1);
  return plan;
}

size_t wg6_slab_floats(const Wg6Problem& q, const Wg6Plan& plan) {
  return (size_t)plan.S * q.krows * q.nt * q.m_pad * q.n_pad;  // (S = 1: the one slab is the staging buffer of the gradient's layout change)
}

int wg6_tickets(const Wg6Problem& q, const Wg6Plan& plan) {
  if (!plan.fused || plan.S <= 1 || plan.variant < 0) return 0;
  const Variant& v = kVariants[plan.variant];
  return (q.m_pad / v.BM) * (q.n_pad / v.BN) * (q.krows / v.NTY);
}

int wg6_launch(const Wg6Problem& q, const Wg6Plan& plan, float* slabs, unsigned* tickets, float* grad, int accumulate, hipStream_t st) {
  LHG_REQUIRE(plan.variant >= 0 && plan.variant < kNumVariants, "wg6: no variant");
  const Variant& v = kVariants[plan.variant];
  LHG_REQUIRE(variant_fits(v, q), "wg6: variant %s does not fit the geometry", v.name);
  LHG_REQUIRE(plan.S >= 1 && plan.S <= 65535, "wg6: bad split count %d", plan.S);
  LHG_REQUIRE(q.strip_cmax != nullptr && q.point_cmax != nullptr, "wgrad (fp32_split_f16 mode): the operands' per-channel absmax vectors are missing (lhg_channel_absmax)");
  LHG_REQUIRE(slabs != nullptr, "wg6: slab workspace missing");
  LHG_REQUIRE(!plan.fused || grad != nullptr, "wg6: fused reduction needs the gradient");
  LHG_REQUIRE(!(plan.fused && plan.S > 1) || tickets != nullptr, "wg6: fused reduction needs zeroed tickets");
  WG6Params p{};
  p.sp = q.strip; p.pp = q.point; p.slabs = slabs; p.s_amax = q.strip_cmax; p.p_amax = q.point_cmax;
  p.Cm = q.Cm; p.Cn = q.Cn; p.lds = q.lds; p.ldp = q.ldp;
  p.Hs = q.Hs; p.Ws = q.Ws; p.gh = q.gh; p.gw = q.gw; p.gwp = q.gw + 2;
  p.dy0 = q.dy0; p.dx0 = q.dx0;
  p.m_pad = q.m_pad; p.n_pad = q.n_pad; p.T = q.krows * q.nt;
  p.total = q.N * q.gh * p.gwp;
  const long long steps = steps_total(q);
  p.kchunk = (int)((steps + plan.S - 1) / plan.S) * 32;
  p.magic_gwp = magic_for(p.gwp); p.magic_gh = magic_for(q.gh);
  static const int xcd = env_int("LHG_XCD", 1);
  p.xcd = xcd;
  p.s_bytes = (unsigned)((((unsigned long long)q.N * q.Hs * q.Ws - 1) * q.lds + q.Cm) * 4ull);
  p.p_bytes = (unsigned)((((unsigned long long)q.N * q.gh * q.gw - 1) * q.ldp + q.Cn) * 4ull);
  p.tickets = (plan.fused && plan.S > 1) ? tickets : nullptr;
  p.grad = plan.fused ? grad : nullptr;
  p.D1 = q.Cm; p.accumulate = accumulate; p.S = plan.S;
  g_last_plan[0] = plan.variant; g_last_plan[1] = plan.S; g_last_plan[2] = plan.fused;
  const dim3 grid((q.m_pad / v.BM) * (q.n_pad / v.BN), q.krows / v.NTY, plan.S);
  hipLaunchKernelGGL(v.kernel, grid, dim3(64 * (v.CWM * v.CWN + 4)), 0, st, p);
  return check_launch("wg6_kernel");
}

int wg6_reduce(const float* slabs, int S, int T, int m_pad, int n_pad, float* grad, int Cn, int Cm, int accumulate, hipStream_t st) {
  const long long total = (long long)T * Cm * (n_pad / 4);
  static const int split4 = [] { const char* e = getenv("LHG_WG6_REDUCE4"); return e ? atoi(e) : 1; }();  // 0: the one-thread-per-output kernel for every S (A/B)
  if (split4 && S >= 32 && total < (1ll << 20)) {  // many splits, few outputs: the split axis over the workgroup's four waves
    hipLaunchKernelGGL(wg6_reduce4_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, slabs, S, T, m_pad, n_pad, grad, Cn, Cm, accumulate);
    return check_launch("wg6_reduce4");
  }
  const int blocks = (int)std::min<long long>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(wg6_reduce_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, st, slabs, S, T, m_pad, n_pad, grad, Cn, Cm, accumulate);
  return check_launch("wg6_reduce");
}

}  // namespace lhg

extern "C" int lhg_wg6_force(int variant, int splits, int fused) {
  lhg::g_force_variant = variant;
  lhg::g_force_splits = splits;
  lhg::g_force_fused = fused;
  return LHG_OK;
}

extern "C" int lhg_wg6_last_plan(int* variant, int* splits, int* fused) {
  if (variant) *variant = lhg::g_last_plan[0];
  if (splits) *splits = lhg::g_last_plan[1];
  if (fused) *fused = lhg::g_last_plan[2];
  return LHG_OK;
}

extern "C" int lhg_wg6_variants(void) { return lhg::kNumVariants; }
extern "C" const char* lhg_wg6_variant_name(int v) { return lhg::wg6_variant_name(v); }
