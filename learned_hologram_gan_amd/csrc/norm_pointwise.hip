// Batch-norm (train statistics, apply, backward, double backward), pooling, layout and
// pointwise kernels of the hot path — all HBM-bound streaming kernels over NHWC fp32:
// 16-byte lanes along the channel axis, one pass per reduction, partials reduced in double.
#include <algorithm>

#include "common.h"

namespace lhg {

// thread (tx, ty): tx indexes a float4 column group, ty a pixel row inside the pass.  At most 16 lanes (64 channels) along the
// channel axis per workgroup; wider tensors spread their channel groups over blockIdx.y, so that the deep layers (24 x 24 x 1024:
// 2304 pixels) still launch hundreds of workgroups instead of a few dozen that walk all channels serially.
struct ColMap {
  int lanes_c, rows;  // lanes_c * rows == 256
  int gy;             // workgroups along the channel axis
};
static inline ColMap col_map(int C) {
  int c4 = C / 4;
  int lanes = 1;
  while (lanes < c4 && lanes < 16) lanes <<= 1;
  return {lanes, 256 / lanes, (c4 + lanes - 1) / lanes};
}

// ------------------------------------------------------------------ generic column reductions
// Streaming loops keep PIXEL_UNROLL pixels of loads in flight per thread: the kernels are bound by memory latency, not by issue
// (with one pixel per iteration a CU holds ~50-100 KB in flight, below what 8 TB/s x the loaded latency needs, and the bf16 storage
// mode — half the bytes per load — ran no faster than fp32).  load(q) only loads, use(q, v) only computes/stores.
constexpr int PIXEL_UNROLL = 4;
template <class L, class F>
__device__ __forceinline__ void pixel_loop(long long q0, long long q1, long long step, L load, F use) {
  long long q = q0;
  for (; q + (PIXEL_UNROLL - 1) * step < q1; q += PIXEL_UNROLL * step) {
    decltype(load(q)) v[PIXEL_UNROLL];
#pragma unroll
    for (int u = 0; u < PIXEL_UNROLL; ++u) v[u] = load(q + u * step);
#pragma unroll
    for (int u = 0; u < PIXEL_UNROLL; ++u) use(q + u * step, v[u]);
  }
  for (; q < q1; q += step) use(q, load(q));
}

// agent-scope relaxed accesses (sc1) for rows that cross workgroups: see draw_last below
template <class V> __device__ __forceinline__ void st_agent(V* p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <class V> __device__ __forceinline__ V ld_agent(const V* p) { return __hip_atomic_load(const_cast<V*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// NSUM partial sums per channel.  pre(cb) -> per-column constants, load(q, cb) -> loaded values of one pixel,
// use(values, constants, acc) adds the NSUM float4 contributions.
template <int NSUM, class P, class L, class F>
__device__ __forceinline__ void column_reduce(long long pixels, int C, int lanes_c, int rows, float* __restrict__ partial, P pre, L load,
                                              F use) {
  __shared__ float red[NSUM][256 * 4];
  const int tx = threadIdx.x % lanes_c, ty = threadIdx.x / lanes_c;
  const long long chunk = (pixels + gridDim.x - 1) / gridDim.x;
  const long long p0 = blockIdx.x * chunk, p1 = min(pixels, p0 + chunk);
  for (int cb0 = blockIdx.y * lanes_c * 4; cb0 < C; cb0 += gridDim.y * lanes_c * 4) {  // uniform trip count: barriers inside
    const int cb = cb0 + tx * 4;
    const bool live = cb < C;
    f32x4 acc[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live) {
      const auto ctx = pre(cb);
      pixel_loop(p0 + ty, p1, rows, [&](long long q) { return load(q, cb); }, [&](long long, const auto& v) { use(v, ctx, acc); });
    }
#pragma unroll
    for (int k = 0; k < NSUM; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[k][(ty * lanes_c + tx) * 4 + e] = acc[k][e];
    __syncthreads();
    if (ty == 0 && live) {
#pragma unroll
      for (int k = 0; k < NSUM; ++k) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rows; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] += red[k][(r * lanes_c + tx) * 4 + e];
        float* row = partial + ((size_t)blockIdx.x * NSUM + k) * C + cb;  // (agent-scope stores: another workgroup may fold this row, column_finish)
#pragma unroll
        for (int e = 0; e < 4; ++e) st_agent(row + e, s[e]);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ in-launch finish of the two-stage reductions (round 5)
// A column reduction used to be two launches: per-workgroup partial rows, then a small kernel that sums the rows (reduce_partials,
// bn_stats_reduce_final, channel_absmax_reduce: 290 launches of 8 - 15 us per train step, each also a host dispatch).  With `Fin` the
// launch that writes the rows finishes them itself: partial rows are grouped by FIN_GROUP; the LAST workgroup of a group to publish its
// row (release, ticket fetch_add, acquire — the hand-off of wg6_kernel.inc: nobody waits for anybody, any dispatch order) folds the
// group's rows into one first-level row, and the last GROUP to do so folds the first-level rows and writes the result.  Rows and groups
// are summed in index order in double whoever arrives last: the result does not depend on the arrival order (bit-repeatable).  Two levels
// because one workgroup walking 2048 rows is latency-bound (~100 us); 32 rows, then <= 64 groups, are two short hops.  The tickets
// are zero before the launch and the last arrivers put the zero back (one persistent buffer per stream: hip_ops._fused_scratch).
constexpr int FIN_GROUP = 32, FIN_MAXG = 64;  // rows per group; groups per channel block (2048 partial rows at most)
struct Fin {
  unsigned* tickets;  // [gridDim.y][1 + FIN_MAXG]; nullptr: no in-launch finish (the rows are all this launch leaves)
  void* level1;       // first-level rows: doubles [groups][NSUM][C] for sums, unsigned [NB][groups][C] for maxima
};

// Rows that cross workgroups (partial rows, first-level rows) are written and read with AGENT-SCOPE relaxed atomic accesses (sc1: they
// write through / bypass the per-XCD L2 and the per-CU vector cache) instead of plain accesses bracketed by release / acquire fences: on
// this chip an agent-scope release is `buffer_wbl2` — a write-back of the XCD's whole L2 — and an acquire `buffer_inv`, executed here by
// every one of ~2000 workgroups per launch (first version of this code: the train step went from 32 to 45 ms).  The rows are a few KB.

// true in every thread of the workgroup that draws ticket `total - 1`.  Every thread's earlier st_agent stores have completed (vmcnt)
// before the ticket is drawn, so whoever draws a later ticket reads them with ld_agent.  Called by all 256 threads.
__device__ __forceinline__ bool draw_last(unsigned* ticket, unsigned total) {
  __shared__ unsigned last_flag;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = old == total - 1u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the next launch finds the ticket at zero again
    last_flag = last ? 1u : 0u;
  }
  __syncthreads();
  return last_flag != 0u;
}

// Sums: partial[row][k][c] floats (row = blockIdx.x of the writer) -> final(c, double sums[NSUM]) called by one thread per channel of
// this workgroup's channel block.  Called by all threads after the workgroup's own row has been stored.
template <int NSUM, class F>
__device__ __forceinline__ void column_finish(const float* __restrict__ partial, int C, int lanes_c, const Fin& fin, F final_fn) {
  __shared__ double fsh[NSUM * 256];
  const int nblk = gridDim.x;
  const int ch = lanes_c * 4, nsl = 256 / ch;
  const int cl = threadIdx.x % ch, slice = threadIdx.x / ch;
  const int c = blockIdx.y * ch + cl;
  const bool live = c < C;
  const int g = blockIdx.x / FIN_GROUP, ngroups = (nblk + FIN_GROUP - 1) / FIN_GROUP;
  const int r0 = g * FIN_GROUP, nr = min(FIN_GROUP, nblk - r0);
  unsigned* tk = fin.tickets + (size_t)blockIdx.y * (1 + FIN_MAXG);
  double* level1 = static_cast<double*>(fin.level1);
  if (!draw_last(tk + 1 + g, (unsigned)nr)) return;
  double acc[NSUM];
#pragma unroll
  for (int k = 0; k < NSUM; ++k) acc[k] = 0.0;
  if (live) {
#pragma unroll 4
    for (int r = slice; r < nr; r += nsl)
#pragma unroll
      for (int k = 0; k < NSUM; ++k) acc[k] += (double)ld_agent(partial + ((size_t)(r0 + r) * NSUM + k) * C + c);
  }
#pragma unroll
  for (int k = 0; k < NSUM; ++k) fsh[k * 256 + slice * ch + cl] = acc[k];
  __syncthreads();
  if (slice == 0 && live) {
#pragma unroll
    for (int k = 0; k < NSUM; ++k) {
      double t = 0.0;
      for (int q = 0; q < nsl; ++q) t += fsh[k * 256 + q * ch + cl];
      st_agent(level1 + ((size_t)g * NSUM + k) * C + c, t);
    }
  }
  if (!draw_last(tk, (unsigned)ngroups)) return;
#pragma unroll
  for (int k = 0; k < NSUM; ++k) acc[k] = 0.0;
  if (live) {
#pragma unroll 4
    for (int q = slice; q < ngroups; q += nsl)
#pragma unroll
      for (int k = 0; k < NSUM; ++k) acc[k] += ld_agent(level1 + ((size_t)q * NSUM + k) * C + c);
  }
#pragma unroll
  for (int k = 0; k < NSUM; ++k) fsh[k * 256 + slice * ch + cl] = acc[k];
  __syncthreads();
  if (slice == 0 && live) {
    double sums[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) {
      double t = 0.0;
      for (int q = 0; q < nsl; ++q) t += fsh[k * 256 + q * ch + cl];
      sums[k] = t;
    }
    final_fn(c, sums);
  }
}

// Maxima: NB partial buffers [row][c] of magnitude bits -> out[b][c].  Buffers that are nullptr are skipped (uniformly).
template <int NB>
__device__ __forceinline__ void channel_max_finish(unsigned* const (&partial)[NB], unsigned* const (&out)[NB], int C, int lanes_c, const Fin& fin) {
  __shared__ unsigned msh[NB * 256];
  const int nblk = gridDim.x;
  const int ch = lanes_c * 4, nsl = 256 / ch;
  const int cl = threadIdx.x % ch, slice = threadIdx.x / ch;
  const int c = blockIdx.y * ch + cl;
  const bool live = c < C;
  const int g = blockIdx.x / FIN_GROUP, ngroups = (nblk + FIN_GROUP - 1) / FIN_GROUP;
  const int r0 = g * FIN_GROUP, nr = min(FIN_GROUP, nblk - r0);
  unsigned* tk = fin.tickets + (size_t)blockIdx.y * (1 + FIN_MAXG);
  unsigned* level1 = static_cast<unsigned*>(fin.level1);  // [NB][groups][C]
  if (!draw_last(tk + 1 + g, (unsigned)nr)) return;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    unsigned m = 0u;
    if (live && partial[b]) {
#pragma unroll 8
      for (int r = slice; r < nr; r += nsl) m = max(m, ld_agent(partial[b] + (size_t)(r0 + r) * C + c));
    }
    msh[b * 256 + slice * ch + cl] = m;
  }
  __syncthreads();
  if (slice == 0 && live) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (!partial[b]) continue;
      unsigned m = 0u;
      for (int q = 0; q < nsl; ++q) m = max(m, msh[b * 256 + q * ch + cl]);
      st_agent(level1 + ((size_t)b * ngroups + g) * C + c, m);
    }
  }
  if (!draw_last(tk, (unsigned)ngroups)) return;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    unsigned m = 0u;
    if (live && partial[b]) {
#pragma unroll 8
      for (int q = slice; q < ngroups; q += nsl) m = max(m, ld_agent(level1 + ((size_t)b * ngroups + q) * C + c));
    }
    msh[b * 256 + slice * ch + cl] = m;
  }
  __syncthreads();
  if (slice == 0 && live) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (!partial[b]) continue;
      unsigned m = 0u;
      for (int q = 0; q < nsl; ++q) m = max(m, msh[b * 256 + q * ch + cl]);
      out[b][c] = m;
    }
  }
}

// Activation tensors are fp32 or — in the bf16 storage mode (lhg_set_activation_dtype) — bf16; every kernel computes in fp32.
// Four channels per lane either way: 16-byte or 8-byte accesses.
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 ld4(const __bf16* p) { return __builtin_convertvector(*reinterpret_cast<const bf16x4_t*>(p), f32x4); }
__device__ __forceinline__ void st4(__bf16* p, f32x4 v) { *reinterpret_cast<bf16x4_t*>(p) = __builtin_convertvector(v, bf16x4_t); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }

struct V1 { f32x4 a; };
struct V2 { f32x4 a, b; };
struct V3 { f32x4 a, b, c; };
struct V4 { f32x4 a, b, c, d; };
__device__ __forceinline__ f32x4 act_grad4(f32x4 yy, int act, float slope) {
  f32x4 m;
#pragma unroll
  for (int e = 0; e < 4; ++e) m[e] = act_grad_from_output(yy[e], act, slope);
  return m;
}

// Running max|v| of what a streaming kernel writes — the tensor scale the fp16-split GEMMs need of their operands (lhg_absmax):
// folded into the producer it costs no pass of its own.  Magnitude bits compare as unsigned (NaN > inf > finite).
__device__ __forceinline__ unsigned amax4(unsigned m, f32x4 v) {
  // v_max_f32 with |.| source modifiers: one instruction per element.  (A NaN element does not raise the maximum here — it still turns
  // into a NaN fp16 term in the GEMM that reads it, so the result is NaN either way; an infinity does and leaves the tensor unscaled.)
  float f = __uint_as_float(m);
#pragma unroll
  for (int e = 0; e < 4; ++e) f = fmaxf(f, fabsf(v[e]));
  return __float_as_uint(f);
}
// `seen`: the slot's value loaded when the workgroup started (its latency hides behind the streaming loop); only a workgroup that beat
// it goes back to memory at all, and then re-reads before the atomic: a few dozen atomics per launch instead of one per workgroup.
__device__ __forceinline__ unsigned amax_peek(const float* out) { return out ? *reinterpret_cast<const volatile unsigned*>(out) : 0u; }
__device__ __forceinline__ void amax_commit(unsigned m, float* out, unsigned seen) {  // every thread of the (256-thread) workgroup calls this
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  __shared__ unsigned red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(red[0], red[1]), max(red[2], red[3]));
    unsigned* o = reinterpret_cast<unsigned*>(out);
    if (m > seen && m > *reinterpret_cast<volatile unsigned*>(o)) atomicMax(o, m);
  }
}

// ------------------------------------------------------------------ BN statistics
// mean / invstd / running statistics of one channel from its two shifted sums (the arithmetic of bn_stats_reduce_final)
__device__ __forceinline__ void bn_stats_final_shifted(int c, double s0, double s1, double shift, long long pixels, int C, float* __restrict__ stats,
                                                       float* running_mean, float* running_var, float momentum, float eps);
template <class T>
__device__ __forceinline__ void bn_stats_final(int c, double s0, double s1, const T* __restrict__ x, long long pixels, int C, float* __restrict__ stats,
                                               float* running_mean, float* running_var, float momentum, float eps) {
  bn_stats_final_shifted(c, s0, s1, (double)ld1(x + c), pixels, C, stats, running_mean, running_var, momentum, eps);
}
__device__ __forceinline__ void bn_stats_final_shifted(int c, double s0, double s1, double shift, long long pixels, int C, float* __restrict__ stats,
                                                       float* running_mean, float* running_var, float momentum, float eps) {
  const double n = (double)pixels;
  const double dm = s0 / n;
  const double mean = shift + dm;
  double var = s1 / n - dm * dm;
  if (var < 0) var = 0;
  stats[c] = (float)mean;
  stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) {
    const double unbiased = n > 1 ? var * n / (n - 1) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

template <class T>
__global__ __launch_bounds__(256) void bn_stats_partial(const T* __restrict__ x, long long pixels, int C, int ld,
                                                        int lanes_c, int rows, float* __restrict__ partial, Fin fin = Fin{nullptr, nullptr},
                                                        float* __restrict__ stats = nullptr, float* running_mean = nullptr,
                                                        float* running_var = nullptr, float momentum = 0.f, float eps = 0.f) {
  // shifted sums around the first pixel of each channel (stable one-pass variance)
  column_reduce<2>(
      pixels, C, lanes_c, rows, partial, [&](int cb) { return ld4(x + cb); },
      [&](long long q, int cb) { return V1{ld4(x + (size_t)q * ld + cb)}; },
      [&](const V1& v, const f32x4& k, f32x4* acc) {
        const f32x4 d = v.a - k;
        acc[0] += d;
        acc[1] += d * d;
      });
  if (fin.tickets)  // (uniform) the launch finishes its own rows: mean / invstd / running statistics without a second launch
    column_finish<2>(partial, C, lanes_c, fin, [&](int c, const double* sm) {
      bn_stats_final(c, sm[0], sm[1], x, pixels, C, stats, running_mean, running_var, momentum, eps);
    });
}

// v = x * a + b with ONE rounding per element (fma), a = invstd * gamma, b = beta - mean * a: the pre-activation of bn_apply.  The backward
// kernels evaluate the same expression when they derive the activation mask from x instead of reading the forward output y
// (no residual, ReLU / LeakyReLU: sign(y) == sign(v) exactly, one tensor read less per pass).
__device__ __forceinline__ f32x4 bn_affine(f32x4 x, f32x4 a, f32x4 b) {
  f32x4 v;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(x[e], a[e], b[e]);
  return v;
}

// Workgroup stage of a per-channel maximum: the row lanes (ty) of each channel lane (tx) are combined through LDS and lane ty == 0
// stores the four channels' maxima.  Called by all 256 threads (two barriers).
__device__ __forceinline__ void channel_max_commit(u32x4* red, const u32x4& mine, int tx, int ty, int lanes_c, int rows, bool live, unsigned* dst) {
  red[ty * lanes_c + tx] = mine;
  __syncthreads();
  if (ty == 0 && live) {
    u32x4 m = mine;
    for (int r = 1; r < rows; ++r) {
      const u32x4 o = red[r * lanes_c + tx];
#pragma unroll
      for (int e = 0; e < 4; ++e) m[e] = max(m[e], o[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) st_agent(dst + e, m[e]);  // (agent scope: channel_max_finish may fold this row from another workgroup)
  }
  __syncthreads();
}

// ------------------------------------------------------------------ BN apply (+residual, +activation)
// cmax_partial (may be null): partial[blockIdx.x][c] = max over this workgroup's pixels of |y[.][c]| (magnitude bits) — the first stage of
// lhg_channel_absmax taken on the way out (the per-channel scales of the weight-gradient GEMM that reads y), finished by
// channel_absmax_reduce over gridDim.x rows.
template <class T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, int ldx, long long pixels, int C,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const T* __restrict__ res, int ldres,
                                                       int act, float slope, T* __restrict__ y, int ldy, int lanes_c, int rows,
                                                       float* __restrict__ y_amax, unsigned* __restrict__ cmax_partial,
                                                       Fin fin = Fin{nullptr, nullptr}, unsigned* __restrict__ cmax_out = nullptr) {
  __shared__ u32x4 cred[256];
  const int tx = threadIdx.x % lanes_c, ty = threadIdx.x / lanes_c;
  unsigned am = 0;
  const unsigned seen = amax_peek(y_amax);
  for (int cb0 = blockIdx.y * lanes_c * 4; cb0 < C; cb0 += gridDim.y * lanes_c * 4) {  // uniform trip count: barriers inside
    const int cb = cb0 + tx * 4;
    const bool live = cb < C;
    f32x4 cm = {0.f, 0.f, 0.f, 0.f};  // per-channel running max|.| (v_max_f32 with |.| modifier: one instruction per element, as amax4)
    if (live) {
      const f32x4 mean = ld4(stats + cb), inv = ld4(stats + C + cb);
      const f32x4 a = inv * ld4(gamma + cb);
      const f32x4 b = ld4(beta + cb) - mean * a;
      pixel_loop(
          (long long)blockIdx.x * rows + ty, pixels, (long long)gridDim.x * rows,
          [&](long long q) {
            V2 v;
            v.a = ld4(x + (size_t)q * ldx + cb);
            if (res) v.b = ld4(res + (size_t)q * ldres + cb);
            return v;
          },
          [&](long long q, const V2& l) {
            f32x4 v = bn_affine(l.a, a, b);
            if (res) v += l.b;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act, slope);
            st4(y + (size_t)q * ldy + cb, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) cm[e] = fmaxf(cm[e], fabsf(v[e]));
          });
      am = max(am, __float_as_uint(fmaxf(fmaxf(cm[0], cm[1]), fmaxf(cm[2], cm[3]))));
    }
    if (cmax_partial) channel_max_commit(cred, __builtin_bit_cast(u32x4, cm), tx, ty, lanes_c, rows, live, cmax_partial + (size_t)blockIdx.x * C + cb);
  }
  if (y_amax) amax_commit(am, y_amax, seen);
  if (fin.tickets && cmax_partial) {  // (uniform) per-channel maxima finished here: cmax_out[c], no channel_absmax_reduce launch later
    unsigned* const pp[1] = {cmax_partial};
    unsigned* const oo[1] = {cmax_out};
    channel_max_finish<1>(pp, oo, C, lanes_c, fin);
  }
}

// ------------------------------------------------------------------ BN backward
// y == nullptr: the activation mask is recomputed from x (gamma / beta non-null; forward had no residual and a ReLU / LeakyReLU)
template <class T>
__global__ __launch_bounds__(256) void bn_bwd_partial(const T* __restrict__ gy, int ldgy, const T* __restrict__ x, int ldx,
                                                      const T* __restrict__ y, int ldy, long long pixels, int C,
                                                      const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int act, float slope, int lanes_c, int rows,
                                                      float* __restrict__ partial, Fin fin = Fin{nullptr, nullptr},
                                                      float* __restrict__ sums = nullptr) {
  const bool recompute = y == nullptr && act != LHG_ACT_NONE;
  column_reduce<2>(
      pixels, C, lanes_c, rows, partial,
      [&](int cb) {
        V4 c;
        c.a = ld4(stats + cb);
        c.b = ld4(stats + C + cb);
        if (recompute) {
          c.c = c.b * ld4(gamma + cb);           // a = invstd * gamma     (as bn_apply_kernel)
          c.d = ld4(beta + cb) - c.a * c.c;      // b = beta - mean * a
        }
        return c;
      },
      [&](long long q, int cb) {
        V3 v;
        v.a = ld4(gy + (size_t)q * ldgy + cb);
        v.b = ld4(x + (size_t)q * ldx + cb);
        if (act != LHG_ACT_NONE && !recompute) v.c = ld4(y + (size_t)q * ldy + cb);
        return v;
      },
      [&](const V3& v, const V4& st, f32x4* acc) {
        f32x4 g = v.a;
        if (act != LHG_ACT_NONE) g *= act_grad4(recompute ? bn_affine(v.b, st.c, st.d) : v.c, act, slope);
        const f32x4 xh = (v.b - st.a) * st.b;
        acc[0] += g;
        acc[1] += g * xh;
      });
  if (fin.tickets)  // (uniform) sums[k][c] for bn_bwd_apply without a reduce launch in between
    column_finish<2>(partial, C, lanes_c, fin, [&](int c, const double* sm) {
      sums[c] = (float)sm[0];
      sums[C + c] = (float)sm[1];
    });
}

// sums[k][c] = sum_b partial[b][k][c] in double.  grid (ceil(C / RP_CH), NSUM), block RP_CH channels x RP_SLICES slices of the partial rows.
// The kernel is pure latency: a handful of workgroups on the whole chip walk <= 2048 rows, and it sits on the chain the step waits for
// (one per BatchNorm call: ~100 per train step).  Round 5: 16 channels x 64 slices with SIXTEEN rows in flight per thread (32 x 32 with four
// in flight before: sixteen dependent round trips for 2048 rows, now two) — 10.4 -> ~6 us per call.
constexpr int RP_SLICES = 64, RP_CH = 16, RP_UNROLL = 16;
// NS columns (k_stride apart) of the rows slice, slice + SL, ... summed in double.  Every batch of RP_UNROLL rows is requested before any
// of it is used, rows past the end clamped and zeroed: ONE memory round trip per RP_UNROLL * SL rows for all NS columns.
template <int NS, int SL, int UN = RP_UNROLL>
__device__ __forceinline__ void reduce_rows_n(const float* __restrict__ src, size_t k_stride, size_t stride, int nblk, int slice, double (&out)[NS]) {
  double acc[NS][4];
#pragma unroll
  for (int k = 0; k < NS; ++k) acc[k][0] = acc[k][1] = acc[k][2] = acc[k][3] = 0.0;
  for (int b = slice; b < nblk; b += UN * SL) {
    float v[NS][UN];
    // rows past the end: the load is CLAMPED to the last row and its value replaced by zero afterwards — a conditional load is a branch,
    // and the requests would wait for each other (the empty asm keeps the compiler from sinking the load under the select again)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const size_t r = (size_t)min(b + u * SL, nblk - 1) * stride;
#pragma unroll
      for (int k = 0; k < NS; ++k) v[k][u] = src[r + k * k_stride];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        asm volatile("" : "+v"(v[k][u]));
        acc[k][u & 3] += (double)(b + u * SL < nblk ? v[k][u] : 0.f);
      }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) out[k] = (acc[k][0] + acc[k][1]) + (acc[k][2] + acc[k][3]);
}
// One column: the round-4 loop (whole batches unpredicated, the rest one row per iteration).  The predicated one-trip form above was
// measured against it for this kernel family (round 5, interleaved A/B of the train step on one box): 30.90 against 30.86 ms, the
// launches 13.9 against 12.9 us under the profiler — no gain, so the single-column reductions keep the loop they had.
__device__ __forceinline__ double reduce_rows(const float* __restrict__ src, size_t stride, int nblk, int slice) {
  double acc[4] = {0, 0, 0, 0};
  int b = slice;
  for (; b + (RP_UNROLL - 1) * RP_SLICES < nblk; b += RP_UNROLL * RP_SLICES) {
    float v[RP_UNROLL];
#pragma unroll
    for (int u = 0; u < RP_UNROLL; ++u) v[u] = src[(size_t)(b + u * RP_SLICES) * stride];
#pragma unroll
    for (int u = 0; u < RP_UNROLL; ++u) acc[u & 3] += (double)v[u];
  }
  for (; b < nblk; b += RP_SLICES) acc[0] += (double)src[(size_t)b * stride];
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

template <int NSUM>
__global__ __launch_bounds__(1024) void reduce_partials(const float* __restrict__ partial, int nblk, int C, float* __restrict__ sums,
                                                        int accumulate = 0) {
  __shared__ double red[RP_SLICES][RP_CH];
  const int lane_c = threadIdx.x % RP_CH, slice = threadIdx.x / RP_CH, k = blockIdx.y;
  const int c = blockIdx.x * RP_CH + lane_c;
  red[slice][lane_c] = c < C ? reduce_rows(partial + (size_t)k * C + c, (size_t)NSUM * C, nblk, slice) : 0.0;
  __syncthreads();
  if (slice == 0 && c < C) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < RP_SLICES; ++i) s += red[i][lane_c];
    float* dst = sums + (size_t)k * C + c;
    *dst = accumulate ? *dst + (float)s : (float)s;
  }
}

// reduce_partials<2> and the statistics' finish in one launch: the channels of a block sum their two shifted-sum columns over the partial
// rows and slice 0 writes mean / invstd and the running statistics (bn_stats_final).
template <class T>
__global__ __launch_bounds__(1024) void bn_stats_reduce_final(const float* __restrict__ partial, int nblk, const T* __restrict__ x,
                                                              long long pixels, int C, float* __restrict__ stats, float* running_mean,
                                                              float* running_var, float momentum, float eps) {
  __shared__ double red[2][RP_SLICES][RP_CH];
  const int lane_c = threadIdx.x % RP_CH, slice = threadIdx.x / RP_CH;
  const int c = blockIdx.x * RP_CH + lane_c;
  {
    double two[2] = {0.0, 0.0};
    if (c < C) reduce_rows_n<2, RP_SLICES>(partial + c, (size_t)C, (size_t)2 * C, nblk, slice, two);  // both columns' rows in one round trip
    red[0][slice][lane_c] = two[0];
    red[1][slice][lane_c] = two[1];
  }
  __syncthreads();
  if (slice != 0 || c >= C) return;
  double sum[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < RP_SLICES; ++i) s += red[k][i][lane_c];
    sum[k] = (double)(float)s;
  }
  bn_stats_final(c, sum[0], sum[1], x, pixels, C, stats, running_mean, running_var, momentum, eps);
}

// The statistics of a tensor whose producer — a conv GEMM's epilogue (gg_epilogue.inc, STATS; lhg_conv2d_forward_stats) — left the first
// stage behind: partial[row][2][C] floats, sums of (x - shift[c]) and their squares over the row's pixels (shift = the conv's bias, null:
// zero).  CH channels x SL row slices per workgroup, SL chosen so that the rows are ONE round trip where possible (16 SL rows in flight
// per channel: the kernel is pure latency, on the chain the step waits for): a 384^2 layer leaves ~9000 rows.
template <int CH, int SL>
__global__ __launch_bounds__(CH * SL) void bn_stats_finish_kernel(const float* __restrict__ partial, int nblk, const float* __restrict__ shift,
                                                                  long long pixels, int C, float* __restrict__ stats, float* running_mean,
                                                                  float* running_var, float momentum, float eps) {
  __shared__ double red[2][SL][CH];
  const int lane_c = threadIdx.x % CH, slice = threadIdx.x / CH;
  const int c = blockIdx.x * CH + lane_c;
  {
    double two[2] = {0.0, 0.0};
    if (c < C) reduce_rows_n<2, SL>(partial + c, (size_t)C, (size_t)2 * C, nblk, slice, two);
    red[0][slice][lane_c] = two[0];
    red[1][slice][lane_c] = two[1];
  }
  __syncthreads();
  // the slices folded pairwise (fixed tree: the same bits whatever the timing), log2(SL) steps instead of one thread walking SL values
#pragma unroll
  for (int h = SL / 2; h > 0; h >>= 1) {
    if (slice < h) {
      red[0][slice][lane_c] += red[0][slice + h][lane_c];
      red[1][slice][lane_c] += red[1][slice + h][lane_c];
    }
    __syncthreads();
  }
  if (slice != 0 || c >= C) return;
  bn_stats_final_shifted(c, red[0][0][lane_c], red[1][0][lane_c], shift ? (double)shift[c] : 0.0, pixels, C, stats, running_mean, running_var, momentum, eps);
}

template <class T>
__global__ __launch_bounds__(256) void bn_bwd_apply(const T* __restrict__ gy, int ldgy, const T* __restrict__ x, int ldx,
                                                    const T* __restrict__ y, int ldy, long long pixels, int C,
                                                    const float* __restrict__ stats, const float* __restrict__ gamma,
                                                    const float* __restrict__ sums, int act, float slope,
                                                    T* __restrict__ gx, int ldgx, T* __restrict__ gres, int ldgres,
                                                    float* __restrict__ ggamma, float* __restrict__ gbeta, int accumulate, int lanes_c,
                                                    int rows, float* __restrict__ gx_amax, const float* __restrict__ beta, float invn,
                                                    unsigned* __restrict__ cmax_partial,        // as bn_apply_kernel's, of gx
                                                    unsigned* __restrict__ cmax_partial_res,    // the same of gres (null without gres)
                                                    float* __restrict__ gres_amax = nullptr,    // max|gres| (null: not measured)
                                                    Fin fin = Fin{nullptr, nullptr},             // in-launch finish of the two partial buffers ->
                                                    unsigned* __restrict__ cmax_out = nullptr, unsigned* __restrict__ cmax_out_res = nullptr) {
  __shared__ u32x4 cred[256];
  const int tx = threadIdx.x % lanes_c, ty = threadIdx.x / lanes_c;
  unsigned am = 0, amr = 0;
  const unsigned seen = amax_peek(gx_amax), seen_r = amax_peek(gres_amax);
  for (int cb0 = blockIdx.y * lanes_c * 4; cb0 < C; cb0 += gridDim.y * lanes_c * 4) {  // uniform trip count: barriers inside
    const int cb = cb0 + tx * 4;
    const bool live = cb < C;
    f32x4 cm = {0.f, 0.f, 0.f, 0.f}, cr = {0.f, 0.f, 0.f, 0.f};
    if (live) {
      const f32x4 mean = ld4(stats + cb), inv = ld4(stats + C + cb), gam = ld4(gamma + cb);
      const f32x4 sg = ld4(sums + cb), sgx = ld4(sums + C + cb);
      if (blockIdx.x == 0 && ty == 0) {  // one writer per channel: accumulation into a gradient slot is race free and ordered by the stream
        if (ggamma) st4(ggamma + cb, accumulate ? ld4(ggamma + cb) + sgx : sgx);
        if (gbeta) st4(gbeta + cb, accumulate ? ld4(gbeta + cb) + sg : sg);
      }
      const f32x4 k = gam * inv, mg = sg * invn, mgx = sgx * invn;
      const bool recompute = y == nullptr && act != LHG_ACT_NONE;  // mask from x: v = x * a + b as bn_apply_kernel evaluated it
      f32x4 fa = inv, fb = inv;
      if (recompute) {
        fa = inv * gam;
        fb = ld4(beta + cb) - mean * fa;
      }
      pixel_loop(
          (long long)blockIdx.x * rows + ty, pixels, (long long)gridDim.x * rows,
          [&](long long q) {
            V3 v;
            v.a = ld4(gy + (size_t)q * ldgy + cb);
            v.b = ld4(x + (size_t)q * ldx + cb);
            if (act != LHG_ACT_NONE && !recompute) v.c = ld4(y + (size_t)q * ldy + cb);
            return v;
          },
          [&](long long q, const V3& v) {
            f32x4 g = v.a;
            if (act != LHG_ACT_NONE) g *= act_grad4(recompute ? bn_affine(v.b, fa, fb) : v.c, act, slope);
            if (gres) {
              st4(gres + (size_t)q * ldgres + cb, g);
              if (cmax_partial_res || gres_amax) {
#pragma unroll
                for (int e = 0; e < 4; ++e) cr[e] = fmaxf(cr[e], fabsf(g[e]));
              }
            }
            const f32x4 xh = (v.b - mean) * inv;
            const f32x4 o = k * (g - mg - xh * mgx);
            st4(gx + (size_t)q * ldgx + cb, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) cm[e] = fmaxf(cm[e], fabsf(o[e]));
          });
      am = max(am, __float_as_uint(fmaxf(fmaxf(cm[0], cm[1]), fmaxf(cm[2], cm[3]))));
      amr = max(amr, __float_as_uint(fmaxf(fmaxf(cr[0], cr[1]), fmaxf(cr[2], cr[3]))));
    }
    if (cmax_partial) channel_max_commit(cred, __builtin_bit_cast(u32x4, cm), tx, ty, lanes_c, rows, live, cmax_partial + (size_t)blockIdx.x * C + cb);
    if (cmax_partial_res) channel_max_commit(cred, __builtin_bit_cast(u32x4, cr), tx, ty, lanes_c, rows, live, cmax_partial_res + (size_t)blockIdx.x * C + cb);
  }
  if (gx_amax) amax_commit(am, gx_amax, seen);
  if (gres_amax) {  // (uniform)
    __syncthreads();  // amax_commit's four-word LDS scratch is reused
    amax_commit(amr, gres_amax, seen_r);
  }
  if (fin.tickets && cmax_partial) {  // (uniform)
    unsigned* const pp[2] = {cmax_partial, cmax_partial_res};
    unsigned* const oo[2] = {cmax_out, cmax_out_res};
    channel_max_finish<2>(pp, oo, C, lanes_c, fin);
  }
}

// ------------------------------------------------------------------ BN double backward (WGAN-GP)
// Notation (xc = x - mean, r = invstd, M = pixels, g = gy * act'(y), q = ggx):
//   ggy = act'(y) * gamma*r * (q - mean(q) - xc r^2 mean(q xc))
//   gx2 = gamma * r^3/M * [ xc*(S_q S_g / M - S_gq + 3 r^2 S_gx S_qx / M) + S_qx (S_g/M - g) + S_gx (S_q/M - q) ]
//   ggamma2 = r * (S_gq - S_q S_g / M - r^2 S_gx S_qx / M)
template <class T>
__global__ __launch_bounds__(256) void bn_bwd2_partial(const T* __restrict__ ggx, const T* __restrict__ gy,
                                                       const T* __restrict__ x, const T* __restrict__ y, long long pixels, int C,
                                                       const float* __restrict__ stats, int act, float slope, int lanes_c, int rows,
                                                       float* __restrict__ partial, Fin fin = Fin{nullptr, nullptr},
                                                       float* __restrict__ sums = nullptr) {
  column_reduce<5>(
      pixels, C, lanes_c, rows, partial, [&](int cb) { return ld4(stats + cb); },
      [&](long long qi, int cb) {
        const size_t o = (size_t)qi * C + cb;
        V4 v;
        v.a = ld4(gy + o);
        v.b = ld4(ggx + o);
        v.c = ld4(x + o);
        if (act != LHG_ACT_NONE) v.d = ld4(y + o);
        return v;
      },
      [&](const V4& v, const f32x4& mean, f32x4* acc) {
        f32x4 g = v.a;
        if (act != LHG_ACT_NONE) g *= act_grad4(v.d, act, slope);
        const f32x4 q = v.b, xc = v.c - mean;
        acc[0] += g;
        acc[1] += g * xc;
        acc[2] += q;
        acc[3] += q * xc;
        acc[4] += g * q;
      });
  if (fin.tickets)
    column_finish<5>(partial, C, lanes_c, fin, [&](int c, const double* sm) {
#pragma unroll
      for (int k = 0; k < 5; ++k) sums[(size_t)k * C + c] = (float)sm[k];
    });
}

template <class T>
__global__ __launch_bounds__(256) void bn_bwd2_apply(const T* __restrict__ ggx, const T* __restrict__ gy,
                                                     const T* __restrict__ x, const T* __restrict__ y, long long pixels, int C,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     const float* __restrict__ sums, int act, float slope,
                                                     T* __restrict__ ggy, T* __restrict__ gx2, float* __restrict__ ggamma2,
                                                     int lanes_c, int rows, float invM) {
  const int tx = threadIdx.x % lanes_c, ty = threadIdx.x / lanes_c;
  for (int cb = (blockIdx.y * lanes_c + tx) * 4; cb < C; cb += gridDim.y * lanes_c * 4) {
    const f32x4 mean = ld4(stats + cb), r = ld4(stats + C + cb), gam = ld4(gamma + cb);
    const f32x4 Sg = ld4(sums + cb), Sgx = ld4(sums + C + cb), Sq = ld4(sums + 2 * C + cb), Sqx = ld4(sums + 3 * C + cb),
                Sgq = ld4(sums + 4 * C + cb);
    const f32x4 r2 = r * r, r3 = r2 * r;
    const f32x4 all_sub = Sq * Sg * invM - Sgq + r2 * Sgx * Sqx * (3.f * invM);
    if (blockIdx.x == 0 && ty == 0 && ggamma2) st4(ggamma2 + cb, r * (Sgq - Sq * Sg * invM - r2 * Sgx * Sqx * invM));
    const f32x4 kI = gam * r3 * invM, kO = gam * r;
    pixel_loop(
        (long long)blockIdx.x * rows + ty, pixels, (long long)gridDim.x * rows,
        [&](long long qi) {
          const size_t o = (size_t)qi * C + cb;
          V4 v;
          v.a = ld4(gy + o);
          v.b = ld4(ggx + o);
          v.c = ld4(x + o);
          if (act != LHG_ACT_NONE) v.d = ld4(y + o);
          return v;
        },
        [&](long long qi, const V4& v) {
          const size_t o = (size_t)qi * C + cb;
          f32x4 g = v.a, m = {1.f, 1.f, 1.f, 1.f};
          if (act != LHG_ACT_NONE) {
            m = act_grad4(v.d, act, slope);
            g *= m;
          }
          const f32x4 q = v.b, xc = v.c - mean;
          st4(gx2 + o, kI * (xc * all_sub + Sqx * (Sg * invM - g) + Sgx * (Sq * invM - q)));
          st4(ggy + o, m * kO * (q - Sq * invM - xc * r2 * Sqx * invM));
        });
  }
}

// ------------------------------------------------------------------ channel sums (bias gradients)
template <class T>
__global__ __launch_bounds__(256) void channel_sum_partial(const T* __restrict__ x, long long pixels, int C, int ld, int lanes_c, int rows,
                                                           float* __restrict__ partial, Fin fin = Fin{nullptr, nullptr},
                                                           float* __restrict__ out = nullptr, int accumulate = 0) {
  column_reduce<1>(
      pixels, C, lanes_c, rows, partial, [&](int) { return 0; }, [&](long long q, int cb) { return V1{ld4(x + (size_t)q * ld + cb)}; },
      [&](const V1& v, int, f32x4* acc) { acc[0] += v.a; });
  if (fin.tickets)
    column_finish<1>(partial, C, lanes_c, fin, [&](int c, const double* sm) { out[c] = accumulate ? out[c] + (float)sm[0] : (float)sm[0]; });
}

// ------------------------------------------------------------------ per-channel max|x| (scales of the fp16-split weight-gradient GEMMs)
// Two stages like the column sums, and for the same reason: one atomicMax per channel and workgroup (2048 x C atomics on C addresses)
// was measured at 127 us per tensor, 12 ms per train step.  partial[b][c] = max over block b's pixels; magnitude bits compare as unsigned.
__global__ __launch_bounds__(256) void channel_absmax_partial(const float* __restrict__ x, long long pixels, int C, int ld, int lanes_c, int rows,
                                                              unsigned* __restrict__ partial, Fin fin = Fin{nullptr, nullptr},
                                                              unsigned* __restrict__ out = nullptr) {
  __shared__ u32x4 red[256];
  const int tx = threadIdx.x % lanes_c, ty = threadIdx.x / lanes_c;
  const long long chunk = (pixels + gridDim.x - 1) / gridDim.x;
  const long long p0 = blockIdx.x * chunk, p1 = min(pixels, p0 + chunk);
  for (int cb0 = blockIdx.y * lanes_c * 4; cb0 < C; cb0 += gridDim.y * lanes_c * 4) {  // uniform trip count: barriers inside
    const int cb = cb0 + tx * 4;
    const bool live = cb < C;
    u32x4 m = {0u, 0u, 0u, 0u};
    if (live)
      pixel_loop(p0 + ty, p1, rows, [&](long long q) { return *reinterpret_cast<const u32x4*>(x + (size_t)q * ld + cb); },
                 [&](long long, const u32x4& v) {
#pragma unroll
                   for (int e = 0; e < 4; ++e) m[e] = max(m[e], v[e] & 0x7fffffffu);
                 });
    red[ty * lanes_c + tx] = m;
    __syncthreads();
    if (ty == 0 && live) {
      for (int r = 1; r < rows; ++r) {
        const u32x4 o = red[r * lanes_c + tx];
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = max(m[e], o[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) st_agent(partial + (size_t)blockIdx.x * C + cb + e, m[e]);
    }
    __syncthreads();
  }
  if (fin.tickets) {
    unsigned* const pp[1] = {partial};
    unsigned* const oo[1] = {out};
    channel_max_finish<1>(pp, oo, C, lanes_c, fin);
  }
}

// out[c] = max_b partial[b][c]; block = 32 channels x 32 slices of the partial rows
constexpr int CM_SLICES = 32;  // 32 channels x 32 slices per block (the sum reductions above use their own split)
__global__ __launch_bounds__(1024) void channel_absmax_reduce(const unsigned* __restrict__ partial, int nblk, int C, unsigned* __restrict__ out) {
  __shared__ unsigned red[CM_SLICES][32];
  const int lane_c = threadIdx.x & 31, slice = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + lane_c;
  unsigned m = 0;
  if (c < C) {
    // eight rows in flight per thread: the loop is latency-bound (two to eight workgroups on the whole chip, 2048 rows: 17.5 us per
    // call with one load per iteration, 94 calls per train step)
    int b = slice;
    for (; b + 7 * CM_SLICES < nblk; b += 8 * CM_SLICES) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b + u * CM_SLICES) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) m = max(m, v[u]);
    }
    for (; b < nblk; b += CM_SLICES) m = max(m, partial[(size_t)b * C + c]);
  }
  red[slice][lane_c] = m;
  __syncthreads();
  if (slice == 0 && c < C) {
#pragma unroll
    for (int i = 1; i < CM_SLICES; ++i) m = max(m, red[i][lane_c]);
    out[c] = m;
  }
}

// ------------------------------------------------------------------ layout
template <class T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int HW, int ld) {
  // one thread per (pixel, channel of ld); reads are strided by HW across channels but each
  // channel plane is swept contiguously by consecutive pixels of consecutive blocks.
  const size_t total = (size_t)N * HW * ld;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % ld);
    const size_t pix = i / ld;
    const size_t n = pix / HW, hw = pix - n * HW;
    st1(dst + i, c < C ? src[(n * C + c) * HW + hw] : 0.f);
  }
}

// Narrow tensors (the step's inputs: 3, 4 or 6 planes into 4 .. 32 channels): one thread per PIXEL — every plane is read along hw by
// consecutive lanes (whole lines), the pixel's ld channels leave as 16-byte stores; no integer division per element.
template <int LD4>  // ld / 4
__global__ __launch_bounds__(256) void nchw_to_nhwc_pixel_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW) {
  const size_t pixels = (size_t)N * HW;
  for (size_t pix = blockIdx.x * (size_t)blockDim.x + threadIdx.x; pix < pixels; pix += (size_t)gridDim.x * blockDim.x) {
    const size_t n = pix / HW, hw = pix - n * HW;
    const float* s0 = src + n * (size_t)C * HW + hw;
    f32x4 v[LD4];
#pragma unroll
    for (int q = 0; q < LD4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[q][e] = (4 * q + e) < C ? s0[(size_t)(4 * q + e) * HW] : 0.f;
#pragma unroll
    for (int q = 0; q < LD4; ++q) *reinterpret_cast<f32x4*>(dst + pix * (4 * LD4) + 4 * q) = v[q];
  }
}

template <class T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, int ld, float* __restrict__ dst, int N, int C, int HW) {
  const size_t total = (size_t)N * C * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t hw = i % HW;
    const size_t nc = i / HW;
    const size_t n = nc / C, c = nc - n * C;
    dst[i] = ld1(src + (n * HW + hw) * ld + c);
  }
}

// ------------------------------------------------------------------ max pool 2x2
template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, int N, int H, int W, int C, int ldx,
                                                          T* __restrict__ y, int ldy) {
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const size_t op = i / C4;
    const int ox = (int)(op % Wo);
    const int oy = (int)((op / Wo) % Ho);
    const size_t n = op / ((size_t)Wo * Ho);
    const T* b = x + ((n * H + 2 * oy) * W + 2 * ox) * (size_t)ldx + c;
    f32x4 v = ld4(b);
    const f32x4 v1 = ld4(b + ldx), v2 = ld4(b + (size_t)W * ldx), v3 = ld4(b + (size_t)(W + 1) * ldx);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaxf(v[e], v1[e]), fmaxf(v2[e], v3[e]));
    st4(y + op * ldy + c, v);
  }
}

template <class T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ x, int ldx, const T* __restrict__ gy, int ldgy,
                                                          int N, int H, int W, int C, T* __restrict__ gx, int ldgx,
                                                          const T* __restrict__ gadd = nullptr, int ldga = 0) {  // gadd: added to gx (same pixels)
  const int Ho = H / 2, Wo = W / 2, C4 = C / 4;
  const size_t total = (size_t)N * Ho * Wo * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const size_t op = i / C4;
    const int ox = (int)(op % Wo);
    const int oy = (int)((op / Wo) % Ho);
    const size_t n = op / ((size_t)Wo * Ho);
    const size_t base = ((n * H + 2 * oy) * W + 2 * ox);
    const size_t offs[4] = {base, base + 1, base + W, base + W + 1};
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ld4(x + offs[k] * ldx + c);
    const f32x4 g = ld4(gy + op * ldgy + c);
    f32x4 o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int best = 0;
      float bv = v[0][e];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][e] > bv) { bv = v[k][e]; best = k; }  // first maximum wins (row-major window scan)
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k][e] = k == best ? g[e] : 0.f;
    }
    if (gadd) {  // the gradient that reached x through its other consumer (a skip connection): one pass instead of autograd's strided add
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] += ld4(gadd + offs[k] * ldga + c);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) st4(gx + offs[k] * ldgx + c, o[k]);
  }
}

template <class T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ g, int ldg, const T* __restrict__ y, int ldy,
                                                      long long pixels, int C, int act, float slope, T* __restrict__ out, int ldo) {
  const int C4 = C / 4;
  const size_t total = (size_t)pixels * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    const size_t q = i / C4;
    f32x4 v = ld4(g + q * ldg + c);
    const f32x4 yy = ld4(y + q * ldy + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= act_grad_from_output(yy[e], act, slope);
    st4(out + q * ldo + c, v);
  }
}

// ------------------------------------------------------------------ Adam
// consts (optional, device): {1 - b1^t, sqrt(1 - b2^t), gradient scale, lr} read by the kernel instead of the by-value arguments, so that a
// captured launch (hipGraph) follows the step count and a schedule's learning rate without being re-recorded.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt, float gscale, const float* __restrict__ consts) {
  // torch.optim.Adam single-tensor path: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
  // denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m/denom
  if (consts) { bc1 = consts[0]; bc2_sqrt = consts[1]; gscale = consts[2]; lr = consts[3]; }
  const float step_size = lr / bc1;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;  // gscale = 1 / world: the data-parallel mean of the all-reduced sum (1.0 is exact: single process)
    const float mi = m[i] + (gi - m[i]) * (1.f - b1);  // lerp form used by torch
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= step_size * (mi / denom);
  }
}

static inline int grid_for(size_t work_items, int per_block = 256, int cap = 4096) {
  size_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > (size_t)cap) b = cap;
  return (int)b;
}

// Column reductions are latency-bound streaming kernels: ~8 resident workgroups per CU (2048 blocks) keep enough
// loads in flight to approach the HBM rate; 512 blocks measured only ~1.6 TB/s.
static inline int partial_blocks(long long pixels, int gy = 1) {
  return (int)std::min<long long>(std::max(1, 2048 / gy), std::max<long long>(1, pixels / 64));
}

// workgroups along the pixel axis of the apply kernels (bn_apply_kernel, bn_bwd_apply) = rows of their per-channel partial maxima
static inline int apply_blocks(long long pixels, const ColMap& cm) { return grid_for((size_t)pixels, cm.rows * 4, std::max(1, 2048 / cm.gy)); }

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace lhg

using namespace lhg;

#define LHG_NHWC_OK(ptr, C, ld, what)                                                                         \
  LHG_REQUIRE((C) % 4 == 0 && (ld) % 4 == 0 && (ld) >= (C) && aligned16(ptr), "%s: NHWC tensor needs C%%4==0, ld%%4==0, 16-byte base (C=%d ld=%d)", what, (int)(C), (int)(ld))

extern "C" {

int lhg_abi_version(void) { return LHG_ABI_VERSION; }
const char* lhg_last_error(void) { return err_buf(); }

}  // extern "C"

// ---- host side, generic over the activation element type T (float, or __bf16 in the bf16 storage mode).  The C entry points take
// `float*` for every tensor (the ABI of the fp32 path); in the bf16 mode the same pointers address bf16 NHWC tensors.
#define LHG_ACT_CALL(fn, ...) (lhg::act_is_bf16() ? fn<__bf16>(__VA_ARGS__) : fn<float>(__VA_ARGS__))
template <class T> static inline const T* as_act(const float* p) { return reinterpret_cast<const T*>(p); }
template <class T> static inline T* as_act(float* p) { return reinterpret_cast<T*>(p); }

template <class T>
static int nchw_to_nhwc_impl(const float* src, float* dst, int N, int C, int H, int W, int ld, lhg_stream_t s) {
  LHG_REQUIRE(ld >= C, "nchw_to_nhwc: ld %d < C %d", ld, C);
  const size_t total = (size_t)N * H * W * ld;
  if (sizeof(T) == 4 && ld % 4 == 0 && ld <= 32 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
    const int g = grid_for((size_t)N * H * W, 256, 16384);
    float* d = reinterpret_cast<float*>(dst);
    switch (ld / 4) {
      case 1: hipLaunchKernelGGL(nchw_to_nhwc_pixel_kernel<1>, dim3(g), dim3(256), 0, as_stream(s), src, d, N, C, H * W); break;
      case 2: hipLaunchKernelGGL(nchw_to_nhwc_pixel_kernel<2>, dim3(g), dim3(256), 0, as_stream(s), src, d, N, C, H * W); break;
      case 4: hipLaunchKernelGGL(nchw_to_nhwc_pixel_kernel<4>, dim3(g), dim3(256), 0, as_stream(s), src, d, N, C, H * W); break;
      case 8: hipLaunchKernelGGL(nchw_to_nhwc_pixel_kernel<8>, dim3(g), dim3(256), 0, as_stream(s), src, d, N, C, H * W); break;
      default: hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(total, 256, 16384)), dim3(256), 0, as_stream(s), src, as_act<T>(dst), N, C, H * W, ld);
    }
    return check_launch("nchw_to_nhwc");
  }
  hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for(total, 256, 16384)), dim3(256), 0, as_stream(s), src, as_act<T>(dst), N, C, H * W, ld);
  return check_launch("nchw_to_nhwc");
}

template <class T>
static int nhwc_to_nchw_impl(const float* src, int ld, float* dst, int N, int C, int H, int W, lhg_stream_t s) {
  LHG_REQUIRE(ld >= C, "nhwc_to_nchw: ld %d < C %d", ld, C);
  const size_t total = (size_t)N * C * H * W;
  hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3(grid_for(total, 256, 16384)), dim3(256), 0, as_stream(s), as_act<T>(src), ld, dst, N, C, H * W);
  return check_launch("nhwc_to_nchw");
}

template <class T>
static int channel_sum_impl(const float* x, long long pixels, int C, int ld, float* out, int accumulate, float* ws, lhg_stream_t s) {
  LHG_NHWC_OK(x, C, ld, "channel_sum");
  const ColMap cm = col_map(C);
  const int nblk = partial_blocks(pixels, cm.gy);
  hipLaunchKernelGGL((channel_sum_partial<T>), dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(x), pixels, C, ld, cm.lanes_c, cm.rows, ws);
  hipLaunchKernelGGL(reduce_partials<1>, dim3((C + RP_CH - 1) / RP_CH, 1), dim3(1024), 0, as_stream(s), ws, nblk, C, out, accumulate);
  return check_launch("channel_sum");
}

template <class T>
static int bn_stats_impl(const float* x, long long pixels, int C, int ld, float* stats, float* running_mean, float* running_var,
                         float momentum, float eps, float* ws, lhg_stream_t s) {
  LHG_NHWC_OK(x, C, ld, "bn_stats");
  LHG_REQUIRE(pixels > 0, "bn_stats: empty tensor");
  const ColMap cm = col_map(C);
  const int nblk = partial_blocks(pixels, cm.gy);
  hipLaunchKernelGGL((bn_stats_partial<T>), dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(x), pixels, C, ld, cm.lanes_c, cm.rows, ws);
  hipLaunchKernelGGL((bn_stats_reduce_final<T>), dim3((C + RP_CH - 1) / RP_CH), dim3(1024), 0, as_stream(s), ws, nblk, as_act<T>(x), pixels, C, stats,
                     running_mean, running_var, momentum, eps);
  return check_launch("bn_stats");
}

template <class T>
static int bn_apply_impl(const float* x, int ldx, long long pixels, int C, const float* stats, const float* gamma, const float* beta,
                         const float* res, int ldres, int act, float slope, float* y, int ldy, float* y_absmax, lhg_stream_t s,
                         float* cmax_partial = nullptr) {
  LHG_NHWC_OK(x, C, ldx, "bn_apply(x)");
  LHG_NHWC_OK(y, C, ldy, "bn_apply(y)");
  if (res) LHG_NHWC_OK(res, C, ldres, "bn_apply(res)");
  const ColMap cm = col_map(C);
  const int nblk = apply_blocks(pixels, cm);
  hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(x), ldx, pixels, C, stats, gamma, beta, as_act<T>(res),
                     ldres, act, slope, as_act<T>(y), ldy, cm.lanes_c, cm.rows, y_absmax, reinterpret_cast<unsigned*>(cmax_partial));
  return check_launch("bn_apply");
}

// phase 0: sums + apply (one call);  1: sums only -> `sums` (2C floats, caller's);  2: apply only from `sums`, statistics over 1/inv_count
// samples (the GLOBAL batch when the sums were all-reduced: synchronised batch statistics)
template <class T>
static int bn_backward_impl(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy, long long pixels, int C,
                            const float* stats, const float* gamma, int act, float slope, float* gx, int ldgx, float* gres, int ldgres,
                            float* ggamma, float* gbeta, int accumulate, float* ws, float* gx_absmax, const float* beta, lhg_stream_t s,
                            int phase = 0, float* sums_io = nullptr, float inv_count = 0.f, float* cmax_partial = nullptr,
                            float* cmax_partial_res = nullptr, float* gres_absmax = nullptr) {
  LHG_NHWC_OK(gy, C, ldgy, "bn_backward(gy)");
  LHG_NHWC_OK(x, C, ldx, "bn_backward(x)");
  if (phase != 1) LHG_NHWC_OK(gx, C, ldgx, "bn_backward(gx)");
  if (act != LHG_ACT_NONE && y) LHG_NHWC_OK(y, C, ldy, "bn_backward(y)");
  if (act != LHG_ACT_NONE && !y)
    LHG_REQUIRE(beta != nullptr && gres == nullptr && (act == LHG_ACT_RELU || act == LHG_ACT_LEAKY),
                "bn_backward: without y the mask is recomputed from x: needs beta, no residual, ReLU / LeakyReLU");
  if (gres) LHG_NHWC_OK(gres, C, ldgres, "bn_backward(gres)");
  LHG_REQUIRE(phase == 0 || sums_io != nullptr, "bn_backward: the split phases need the caller's sums buffer (2*C floats)");
  const ColMap cm = col_map(C);
  const int nblk = partial_blocks(pixels, cm.gy);
  float* sums = phase == 0 ? ws + (size_t)nblk * 2 * C : sums_io;
  if (phase != 2) {
    hipLaunchKernelGGL((bn_bwd_partial<T>), dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(gy), ldgy, as_act<T>(x), ldx, as_act<T>(y), ldy, pixels, C,
                       stats, gamma, beta, act, slope, cm.lanes_c, cm.rows, ws);
    hipLaunchKernelGGL(reduce_partials<2>, dim3((C + RP_CH - 1) / RP_CH, 2), dim3(1024), 0, as_stream(s), ws, nblk, C, sums, 0);
  }
  if (phase != 1) {
    const int nb2 = apply_blocks(pixels, cm);
    const float invn = phase == 2 ? inv_count : 1.f / (float)pixels;
    hipLaunchKernelGGL((bn_bwd_apply<T>), dim3(nb2, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(gy), ldgy, as_act<T>(x), ldx, as_act<T>(y), ldy, pixels, C, stats,
                       gamma, sums, act, slope, as_act<T>(gx), ldgx, as_act<T>(gres), ldgres, ggamma, gbeta, accumulate, cm.lanes_c, cm.rows, gx_absmax, beta, invn,
                       reinterpret_cast<unsigned*>(cmax_partial), reinterpret_cast<unsigned*>(gres ? cmax_partial_res : nullptr), gres ? gres_absmax : nullptr);
  }
  return check_launch("bn_backward");
}

template <class T>
static int bn_backward_backward_impl(const float* ggx, const float* gy, const float* x, const float* y, long long pixels, int C,
                                     const float* stats, const float* gamma, int act, float slope, float* ggy, float* gx2, float* ggamma2,
                                     float* ws, lhg_stream_t s, int phase = 0, float* sums_io = nullptr, float inv_count = 0.f) {
  LHG_NHWC_OK(ggx, C, C, "bn_backward_backward(ggx)");
  LHG_REQUIRE(aligned16(gy) && aligned16(x) && (phase == 1 || (aligned16(ggy) && aligned16(gx2))), "bn_backward_backward: unaligned tensor");
  LHG_REQUIRE(phase == 0 || sums_io != nullptr, "bn_backward_backward: the split phases need the caller's sums buffer (5*C floats)");
  const ColMap cm = col_map(C);
  const int nblk = partial_blocks(pixels, cm.gy);
  float* sums = phase == 0 ? ws + (size_t)nblk * 5 * C : sums_io;
  if (phase != 2) {
    hipLaunchKernelGGL((bn_bwd2_partial<T>), dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(ggx), as_act<T>(gy), as_act<T>(x), as_act<T>(y), pixels, C,
                       stats, act, slope, cm.lanes_c, cm.rows, ws);
    hipLaunchKernelGGL(reduce_partials<5>, dim3((C + RP_CH - 1) / RP_CH, 5), dim3(1024), 0, as_stream(s), ws, nblk, C, sums, 0);
  }
  if (phase != 1) {
    const int nb2 = grid_for((size_t)pixels, cm.rows * 4, std::max(1, 4096 / cm.gy));
    const float invM = phase == 2 ? inv_count : 1.f / (float)pixels;
    hipLaunchKernelGGL((bn_bwd2_apply<T>), dim3(nb2, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(ggx), as_act<T>(gy), as_act<T>(x), as_act<T>(y), pixels, C, stats,
                       gamma, sums, act, slope, as_act<T>(ggy), as_act<T>(gx2), ggamma2, cm.lanes_c, cm.rows, invM);
  }
  return check_launch("bn_backward_backward");
}

template <class T>
static int maxpool_fwd_impl(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, lhg_stream_t s) {
  LHG_NHWC_OK(x, C, ldx, "maxpool(x)");
  LHG_NHWC_OK(y, C, ldy, "maxpool(y)");
  LHG_REQUIRE(H % 2 == 0 && W % 2 == 0, "maxpool2x2: odd extent %dx%d", H, W);
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(grid_for(total, 256, 8192)), dim3(256), 0, as_stream(s), as_act<T>(x), N, H, W, C, ldx, as_act<T>(y), ldy);
  return check_launch("maxpool_fwd");
}

template <class T>
static int maxpool_bwd_impl(const float* x, int ldx, const float* gy, int ldgy, int N, int H, int W, int C, float* gx, int ldgx, lhg_stream_t s,
                            const float* gadd = nullptr, int ldga = 0) {
  LHG_NHWC_OK(x, C, ldx, "maxpool_bwd(x)");
  LHG_NHWC_OK(gy, C, ldgy, "maxpool_bwd(gy)");
  LHG_NHWC_OK(gx, C, ldgx, "maxpool_bwd(gx)");
  if (gadd) LHG_NHWC_OK(gadd, C, ldga, "maxpool_bwd(gadd)");
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid_for(total, 256, 8192)), dim3(256), 0, as_stream(s), as_act<T>(x), ldx, as_act<T>(gy), ldgy, N, H, W,
                     C, as_act<T>(gx), ldgx, as_act<T>(gadd), ldga);
  return check_launch("maxpool_bwd");
}

template <class T>
static int act_backward_impl(const float* g, int ldg, const float* y, int ldy, long long pixels, int C, int act, float slope, float* out,
                             int ldo, lhg_stream_t s) {
  LHG_NHWC_OK(g, C, ldg, "act_backward(g)");
  LHG_NHWC_OK(y, C, ldy, "act_backward(y)");
  LHG_NHWC_OK(out, C, ldo, "act_backward(out)");
  const size_t total = (size_t)pixels * (C / 4);
  hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(grid_for(total, 256, 8192)), dim3(256), 0, as_stream(s), as_act<T>(g), ldg, as_act<T>(y), ldy, pixels, C, act,
                     slope, as_act<T>(out), ldo);
  return check_launch("act_backward");
}

// ---- fused calls (ABI 9): every two-stage reduction finishes inside the launch that writes its partial rows (Fin), a BatchNorm forward
// or backward is ONE entry point.  Scratch: `ws` (LHG_FUSED_WS_FLOATS floats) and `tickets` (LHG_FUSED_TICKETS zero-initialised unsigned,
// self-cleaning) — one persistent pair per stream, because launches of one stream never overlap and everything in them is consumed by
// the same call.
struct FusedWs {
  float* partial;      // [rows][NSUM][C] floats, rows <= 2048 / gy
  double* level1;      // [groups][NSUM][C] doubles
  float* sums;         // [NSUM][C] floats
  unsigned* cpart[2];  // per-channel partial maxima of the apply kernels' one or two outputs
  unsigned* clevel1;   // [2][groups][C]
};
static_assert(LHG_FUSED_WS_FLOATS >= 655360 + 81920 + 20480 + 2 * 131072 + 16384, "fused workspace layout");
static inline FusedWs fused_ws(float* ws) {
  FusedWs w;
  w.partial = ws;
  w.level1 = reinterpret_cast<double*>(ws + 655360);
  w.sums = ws + 655360 + 81920;
  w.cpart[0] = reinterpret_cast<unsigned*>(ws + 655360 + 81920 + 20480);
  w.cpart[1] = w.cpart[0] + 131072;
  w.clevel1 = w.cpart[1] + 131072;
  return w;
}
#define LHG_FUSED_OK(ws, tickets, C, what)                                                                                        \
  LHG_REQUIRE((ws) != nullptr && (tickets) != nullptr && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 && (C) <= 4096,              \
              "%s: needs the fused scratch (LHG_FUSED_WS_FLOATS floats, 16-byte aligned; LHG_FUSED_TICKETS zeroed unsigned) and C <= 4096 (C=%d)", what, (int)(C))

template <class T>
static int bn_forward_train_impl(const float* x, int ldx, long long pixels, int C, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float momentum, float eps, const float* res, int ldres, int act, float slope, float* y, int ldy,
                                 float* stats, float* y_absmax, float* y_chanmax, int chanmax_finish, float* ws, unsigned* tickets, lhg_stream_t s) {
  LHG_NHWC_OK(x, C, ldx, "bn_forward_train(x)");
  LHG_NHWC_OK(y, C, ldy, "bn_forward_train(y)");
  if (res) LHG_NHWC_OK(res, C, ldres, "bn_forward_train(res)");
  LHG_REQUIRE(pixels > 0 && stats != nullptr && gamma != nullptr && beta != nullptr, "bn_forward_train: empty tensor or missing statistics / affine parameters");
  LHG_FUSED_OK(ws, tickets, C, "bn_forward_train");
  LHG_REQUIRE(y_chanmax == nullptr || sizeof(T) == 4, "bn_forward_train: per-channel maxima are measured of fp32 tensors only");
  const ColMap cm = col_map(C);
  const FusedWs w = fused_ws(ws);
  const int nblk = partial_blocks(pixels, cm.gy);
  hipLaunchKernelGGL((bn_stats_partial<T>), dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(x), pixels, C, ldx, cm.lanes_c, cm.rows, w.partial,
                     Fin{tickets, w.level1}, stats, running_mean, running_var, momentum, eps);
  const int nb2 = apply_blocks(pixels, cm);
  // chanmax_finish != 0: y_chanmax receives the C finished maxima (the apply launch folds its own partial rows: ~11 us longer);
  // 0: y_chanmax IS the partial-row buffer (lhg_chanmax_partial_rows x C floats), finished later by lhg_channel_absmax_finish on whatever stream asks
  unsigned* rows_out = y_chanmax ? (chanmax_finish ? w.cpart[0] : reinterpret_cast<unsigned*>(y_chanmax)) : nullptr;
  hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(nb2, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(x), ldx, pixels, C, stats, gamma, beta, as_act<T>(res),
                     ldres, act, slope, as_act<T>(y), ldy, cm.lanes_c, cm.rows, y_absmax, rows_out,
                     Fin{(y_chanmax && chanmax_finish) ? tickets : nullptr, w.clevel1}, reinterpret_cast<unsigned*>(y_chanmax));
  return check_launch("bn_forward_train");
}

template <class T>
static int bn_backward_fused_impl(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy, long long pixels, int C,
                                  const float* stats, const float* gamma, const float* beta, int act, float slope, float* gx, int ldgx, float* gres,
                                  int ldgres, float* ggamma, float* gbeta, int accumulate, float* gx_absmax, float* gres_absmax, float* gx_chanmax,
                                  float* gres_chanmax, int chanmax_finish, float* ws, unsigned* tickets, lhg_stream_t s) {
  LHG_NHWC_OK(gy, C, ldgy, "bn_backward_fused(gy)");
  LHG_NHWC_OK(x, C, ldx, "bn_backward_fused(x)");
  LHG_NHWC_OK(gx, C, ldgx, "bn_backward_fused(gx)");
  if (act != LHG_ACT_NONE && y) LHG_NHWC_OK(y, C, ldy, "bn_backward_fused(y)");
  if (act != LHG_ACT_NONE && !y)
    LHG_REQUIRE(beta != nullptr && gres == nullptr && (act == LHG_ACT_RELU || act == LHG_ACT_LEAKY),
                "bn_backward_fused: without y the mask is recomputed from x: needs beta, no residual, ReLU / LeakyReLU");
  if (gres) LHG_NHWC_OK(gres, C, ldgres, "bn_backward_fused(gres)");
  LHG_FUSED_OK(ws, tickets, C, "bn_backward_fused");
  LHG_REQUIRE((gx_chanmax == nullptr && gres_chanmax == nullptr) || sizeof(T) == 4, "bn_backward_fused: per-channel maxima are measured of fp32 tensors only");
  LHG_REQUIRE(gres_chanmax == nullptr || (gres != nullptr && gx_chanmax != nullptr), "bn_backward_fused: gres_chanmax needs gres and gx_chanmax");
  const ColMap cm = col_map(C);
  const FusedWs w = fused_ws(ws);
  const int nblk = partial_blocks(pixels, cm.gy);
  hipLaunchKernelGGL((bn_bwd_partial<T>), dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(gy), ldgy, as_act<T>(x), ldx, as_act<T>(y), ldy, pixels, C,
                     stats, gamma, beta, act, slope, cm.lanes_c, cm.rows, w.partial, Fin{tickets, w.level1}, w.sums);
  const int nb2 = apply_blocks(pixels, cm);
  hipLaunchKernelGGL((bn_bwd_apply<T>), dim3(nb2, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(gy), ldgy, as_act<T>(x), ldx, as_act<T>(y), ldy, pixels, C, stats,
                     gamma, w.sums, act, slope, as_act<T>(gx), ldgx, as_act<T>(gres), ldgres, ggamma, gbeta, accumulate, cm.lanes_c, cm.rows, gx_absmax, beta,
                     1.f / (float)pixels, gx_chanmax ? (chanmax_finish ? w.cpart[0] : reinterpret_cast<unsigned*>(gx_chanmax)) : nullptr,
                     gres_chanmax ? (chanmax_finish ? w.cpart[1] : reinterpret_cast<unsigned*>(gres_chanmax)) : nullptr, gres ? gres_absmax : nullptr,
                     Fin{(gx_chanmax && chanmax_finish) ? tickets : nullptr, w.clevel1}, reinterpret_cast<unsigned*>(gx_chanmax),
                     reinterpret_cast<unsigned*>(gres_chanmax));
  return check_launch("bn_backward_fused");
}

template <class T>
static int bn_backward_backward_fused_impl(const float* ggx, const float* gy, const float* x, const float* y, long long pixels, int C,
                                           const float* stats, const float* gamma, int act, float slope, float* ggy, float* gx2, float* ggamma2,
                                           float* ws, unsigned* tickets, lhg_stream_t s) {
  LHG_NHWC_OK(ggx, C, C, "bn_backward_backward_fused(ggx)");
  LHG_REQUIRE(aligned16(gy) && aligned16(x) && aligned16(ggy) && aligned16(gx2), "bn_backward_backward_fused: unaligned tensor");
  LHG_FUSED_OK(ws, tickets, C, "bn_backward_backward_fused");
  const ColMap cm = col_map(C);
  const FusedWs w = fused_ws(ws);
  const int nblk = partial_blocks(pixels, cm.gy);
  hipLaunchKernelGGL((bn_bwd2_partial<T>), dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(ggx), as_act<T>(gy), as_act<T>(x), as_act<T>(y), pixels, C,
                     stats, act, slope, cm.lanes_c, cm.rows, w.partial, Fin{tickets, w.level1}, w.sums);
  const int nb2 = grid_for((size_t)pixels, cm.rows * 4, std::max(1, 4096 / cm.gy));
  hipLaunchKernelGGL((bn_bwd2_apply<T>), dim3(nb2, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(ggx), as_act<T>(gy), as_act<T>(x), as_act<T>(y), pixels, C, stats,
                     gamma, w.sums, act, slope, as_act<T>(ggy), as_act<T>(gx2), ggamma2, cm.lanes_c, cm.rows, 1.f / (float)pixels);
  return check_launch("bn_backward_backward_fused");
}

template <class T>
static int channel_sum_fused_impl(const float* x, long long pixels, int C, int ld, float* out, int accumulate, float* ws, unsigned* tickets, lhg_stream_t s) {
  LHG_NHWC_OK(x, C, ld, "channel_sum_fused");
  LHG_FUSED_OK(ws, tickets, C, "channel_sum_fused");
  LHG_REQUIRE(pixels > 0 && out != nullptr, "channel_sum_fused: empty tensor or missing output");
  const ColMap cm = col_map(C);
  const FusedWs w = fused_ws(ws);
  const int nblk = partial_blocks(pixels, cm.gy);
  hipLaunchKernelGGL((channel_sum_partial<T>), dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), as_act<T>(x), pixels, C, ld, cm.lanes_c, cm.rows, w.partial,
                     Fin{tickets, w.level1}, out, accumulate);
  return check_launch("channel_sum_fused");
}

extern "C" {

int lhg_set_activation_dtype(int dtype) {
  LHG_REQUIRE(dtype == LHG_DTYPE_F32 || dtype == LHG_DTYPE_BF16, "set_activation_dtype: unknown dtype %d", dtype);
  lhg::act_dtype() = dtype;
  return LHG_OK;
}
int lhg_get_activation_dtype(void) { return lhg::act_dtype(); }

int lhg_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int ld, lhg_stream_t s) {
  return LHG_ACT_CALL(nchw_to_nhwc_impl, src, dst, N, C, H, W, ld, s);
}
int lhg_nhwc_to_nchw(const float* src, int ld, float* dst, int N, int C, int H, int W, lhg_stream_t s) {
  return LHG_ACT_CALL(nhwc_to_nchw_impl, src, ld, dst, N, C, H, W, s);
}
int lhg_channel_sum(const float* x, long long pixels, int C, int ld, float* out, int accumulate, float* ws, lhg_stream_t s) {
  return LHG_ACT_CALL(channel_sum_impl, x, pixels, C, ld, out, accumulate, ws, s);
}
int lhg_channel_absmax(const float* x, long long pixels, int C, int ld, float* out, float* ws, lhg_stream_t s) {
  LHG_NHWC_OK(x, C, ld, "channel_absmax");
  LHG_REQUIRE(pixels > 0 && out != nullptr && ws != nullptr, "channel_absmax: empty tensor or missing output / workspace");
  LHG_REQUIRE(!act_is_bf16(), "channel_absmax: fp32 tensors only (the bf16 storage mode does not use it)");
  const ColMap cm = col_map(C);
  const int nblk = partial_blocks(pixels, cm.gy);
  hipLaunchKernelGGL(channel_absmax_partial, dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), x, pixels, C, ld, cm.lanes_c, cm.rows, reinterpret_cast<unsigned*>(ws));
  hipLaunchKernelGGL(channel_absmax_reduce, dim3((C + 31) / 32), dim3(1024), 0, as_stream(s), reinterpret_cast<const unsigned*>(ws), nblk, C,
                     reinterpret_cast<unsigned*>(out));
  return check_launch("channel_absmax");
}
long long lhg_chanmax_partial_rows(long long pixels, int C) { return pixels > 0 && C > 0 ? apply_blocks(pixels, col_map(C)) : 0; }
int lhg_channel_absmax_finish(const float* partial, long long pixels, int C, float* out, lhg_stream_t s) {
  LHG_REQUIRE(partial != nullptr && out != nullptr && pixels > 0 && C > 0 && C % 4 == 0, "channel_absmax_finish: missing buffer or bad shape");
  hipLaunchKernelGGL(channel_absmax_reduce, dim3((C + 31) / 32), dim3(1024), 0, as_stream(s), reinterpret_cast<const unsigned*>(partial),
                     apply_blocks(pixels, col_map(C)), C, reinterpret_cast<unsigned*>(out));
  return check_launch("channel_absmax_finish");
}
int lhg_channel_absmax_finish_rows(const float* partial, int rows, int C, float* out, lhg_stream_t s) {
  LHG_REQUIRE(partial != nullptr && out != nullptr && rows > 0 && C > 0 && C % 4 == 0, "channel_absmax_finish_rows: missing buffer or bad shape");
  hipLaunchKernelGGL(channel_absmax_reduce, dim3((C + 31) / 32), dim3(1024), 0, as_stream(s), reinterpret_cast<const unsigned*>(partial), rows, C,
                     reinterpret_cast<unsigned*>(out));
  return check_launch("channel_absmax_finish_rows");
}
int lhg_bn_apply_chanmax(const float* x, int ldx, long long pixels, int C, const float* stats, const float* gamma, const float* beta,
                         const float* res, int ldres, int act, float slope, float* y, int ldy, float* y_absmax, float* y_chanmax_partial,
                         lhg_stream_t s) {
  LHG_REQUIRE(y_chanmax_partial != nullptr && !act_is_bf16(), "bn_apply_chanmax: fp32 tensors and a partial buffer (lhg_chanmax_partial_rows x C floats)");
  return bn_apply_impl<float>(x, ldx, pixels, C, stats, gamma, beta, res, ldres, act, slope, y, ldy, y_absmax, s, y_chanmax_partial);
}
int lhg_bn_backward_chanmax(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy, long long pixels, int C,
                            const float* stats, const float* gamma, int act, float slope, float* gx, int ldgx, float* gres, int ldgres,
                            float* ggamma, float* gbeta, int accumulate, float* ws, float* gx_absmax, const float* beta,
                            float* gx_chanmax_partial, float* gres_chanmax_partial, float* gres_absmax, lhg_stream_t s) {
  LHG_REQUIRE(gx_chanmax_partial != nullptr && !act_is_bf16(), "bn_backward_chanmax: fp32 tensors and a partial buffer (lhg_chanmax_partial_rows x C floats)");
  return bn_backward_impl<float>(gy, ldgy, x, ldx, y, ldy, pixels, C, stats, gamma, act, slope, gx, ldgx, gres, ldgres, ggamma, gbeta, accumulate, ws,
                                 gx_absmax, beta, s, 0, nullptr, 0.f, gx_chanmax_partial, gres_chanmax_partial, gres_absmax);
}
int lhg_bn_stats(const float* x, long long pixels, int C, int ld, float* stats, float* running_mean, float* running_var,
                 float momentum, float eps, float* ws, lhg_stream_t s) {
  return LHG_ACT_CALL(bn_stats_impl, x, pixels, C, ld, stats, running_mean, running_var, momentum, eps, ws, s);
}
int lhg_bn_stats_finish(const float* partial, int rows, const float* shift, long long pixels, int C, float* stats, float* running_mean,
                        float* running_var, float momentum, float eps, lhg_stream_t s) {
  LHG_REQUIRE(partial != nullptr && rows > 0 && pixels > 0 && C > 0, "bn_stats_finish: empty input (rows %d, pixels %lld, C %d)", rows, pixels, C);
#define LHG_FINISH(CH, SL) hipLaunchKernelGGL((bn_stats_finish_kernel<CH, SL>), dim3((C + CH - 1) / CH), dim3(CH * SL), 0, as_stream(s), partial, rows, shift, \
                                              pixels, C, stats, running_mean, running_var, momentum, eps)
  // (tools/time_bn_finish.py: 9216 rows x 64 channels 14.6 us on 8 x 128 against 21.8 on 1 x 1024, 18500 rows 47 against 39; thirty-two rows
  //  in flight per thread instead of sixteen: twice as slow in every form)
  if (rows > 12288) LHG_FINISH(1, 1024);
  else if (rows > 4096) LHG_FINISH(8, 128);
  else if (rows > 1024) LHG_FINISH(4, 256);
  else LHG_FINISH(16, 64);
#undef LHG_FINISH
  return check_launch("bn_stats_finish");
}
int lhg_bn_apply(const float* x, int ldx, long long pixels, int C, const float* stats, const float* gamma, const float* beta,
                 const float* res, int ldres, int act, float slope, float* y, int ldy, float* y_absmax, lhg_stream_t s) {
  return LHG_ACT_CALL(bn_apply_impl, x, ldx, pixels, C, stats, gamma, beta, res, ldres, act, slope, y, ldy, y_absmax, s);
}
int lhg_bn_backward(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy, long long pixels, int C,
                    const float* stats, const float* gamma, int act, float slope, float* gx, int ldgx, float* gres, int ldgres,
                    float* ggamma, float* gbeta, int accumulate, float* ws, float* gx_absmax, const float* beta, lhg_stream_t s) {
  return LHG_ACT_CALL(bn_backward_impl, gy, ldgy, x, ldx, y, ldy, pixels, C, stats, gamma, act, slope, gx, ldgx, gres, ldgres, ggamma, gbeta,
                      accumulate, ws, gx_absmax, beta, s);
}
int lhg_bn_backward_backward(const float* ggx, const float* gy, const float* x, const float* y, long long pixels, int C,
                             const float* stats, const float* gamma, int act, float slope, float* ggy, float* gx2, float* ggamma2,
                             float* ws, lhg_stream_t s) {
  return LHG_ACT_CALL(bn_backward_backward_impl, ggx, gy, x, y, pixels, C, stats, gamma, act, slope, ggy, gx2, ggamma2, ws, s);
}
// ---- synchronised batch statistics (data-parallel replicas normalising over the GLOBAL batch): the two halves of the calls above, with the
// per-channel sums in the caller's hands in between (all-reduce them, then apply with inv_count = 1 / global sample count)
int lhg_bn_backward_sums(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy, long long pixels, int C,
                         const float* stats, const float* gamma, int act, float slope, float* sums, float* ws, const float* beta, lhg_stream_t s) {
  return LHG_ACT_CALL(bn_backward_impl, gy, ldgy, x, ldx, y, ldy, pixels, C, stats, gamma, act, slope, nullptr, C, nullptr, C, nullptr, nullptr, 0, ws,
                      nullptr, beta, s, 1, sums, 0.f);
}
int lhg_bn_backward_apply(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy, long long pixels, int C,
                          const float* stats, const float* gamma, const float* sums, float inv_count, int act, float slope, float* gx, int ldgx,
                          float* gres, int ldgres, float* gx_absmax, const float* beta, lhg_stream_t s) {
  LHG_REQUIRE(inv_count > 0.f, "bn_backward_apply: inv_count must be 1 / (samples behind the sums)");
  return LHG_ACT_CALL(bn_backward_impl, gy, ldgy, x, ldx, y, ldy, pixels, C, stats, gamma, act, slope, gx, ldgx, gres, ldgres, nullptr, nullptr, 0, nullptr,
                      gx_absmax, beta, s, 2, const_cast<float*>(sums), inv_count);
}
int lhg_bn_backward_backward_sums(const float* ggx, const float* gy, const float* x, const float* y, long long pixels, int C,
                                  const float* stats, const float* gamma, int act, float slope, float* sums, float* ws, lhg_stream_t s) {
  return LHG_ACT_CALL(bn_backward_backward_impl, ggx, gy, x, y, pixels, C, stats, gamma, act, slope, nullptr, nullptr, nullptr, ws, s, 1, sums, 0.f);
}
int lhg_bn_backward_backward_apply(const float* ggx, const float* gy, const float* x, const float* y, long long pixels, int C,
                                   const float* stats, const float* gamma, const float* sums, float inv_count, int act, float slope,
                                   float* ggy, float* gx2, float* ggamma2, lhg_stream_t s) {
  LHG_REQUIRE(inv_count > 0.f, "bn_backward_backward_apply: inv_count must be 1 / (samples behind the sums)");
  return LHG_ACT_CALL(bn_backward_backward_impl, ggx, gy, x, y, pixels, C, stats, gamma, act, slope, ggy, gx2, ggamma2, nullptr, s, 2,
                      const_cast<float*>(sums), inv_count);
}
// ---- ABI 9: fused calls (see bn_forward_train_impl)
long long lhg_fused_workspace_floats(void) { return LHG_FUSED_WS_FLOATS; }
int lhg_fused_ticket_count(void) { return LHG_FUSED_TICKETS; }
int lhg_bn_forward_train(const float* x, int ldx, long long pixels, int C, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float momentum, float eps, const float* res, int ldres, int act, float slope, float* y, int ldy,
                         float* stats, float* y_absmax, float* y_chanmax, int chanmax_finish, float* ws, unsigned* tickets, lhg_stream_t s) {
  return LHG_ACT_CALL(bn_forward_train_impl, x, ldx, pixels, C, gamma, beta, running_mean, running_var, momentum, eps, res, ldres, act, slope, y, ldy, stats,
                      y_absmax, y_chanmax, chanmax_finish, ws, tickets, s);
}
int lhg_bn_backward_fused(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy, long long pixels, int C, const float* stats,
                          const float* gamma, const float* beta, int act, float slope, float* gx, int ldgx, float* gres, int ldgres, float* ggamma,
                          float* gbeta, int accumulate, float* gx_absmax, float* gres_absmax, float* gx_chanmax, float* gres_chanmax, int chanmax_finish,
                          float* ws, unsigned* tickets, lhg_stream_t s) {
  return LHG_ACT_CALL(bn_backward_fused_impl, gy, ldgy, x, ldx, y, ldy, pixels, C, stats, gamma, beta, act, slope, gx, ldgx, gres, ldgres, ggamma, gbeta,
                      accumulate, gx_absmax, gres_absmax, gx_chanmax, gres_chanmax, chanmax_finish, ws, tickets, s);
}
int lhg_bn_backward_backward_fused(const float* ggx, const float* gy, const float* x, const float* y, long long pixels, int C, const float* stats,
                                   const float* gamma, int act, float slope, float* ggy, float* gx2, float* ggamma2, float* ws, unsigned* tickets,
                                   lhg_stream_t s) {
  return LHG_ACT_CALL(bn_backward_backward_fused_impl, ggx, gy, x, y, pixels, C, stats, gamma, act, slope, ggy, gx2, ggamma2, ws, tickets, s);
}
int lhg_channel_sum_fused(const float* x, long long pixels, int C, int ld, float* out, int accumulate, float* ws, unsigned* tickets, lhg_stream_t s) {
  return LHG_ACT_CALL(channel_sum_fused_impl, x, pixels, C, ld, out, accumulate, ws, tickets, s);
}
int lhg_channel_absmax_fused(const float* x, long long pixels, int C, int ld, float* out, float* ws, unsigned* tickets, lhg_stream_t s) {
  LHG_NHWC_OK(x, C, ld, "channel_absmax_fused");
  LHG_FUSED_OK(ws, tickets, C, "channel_absmax_fused");
  LHG_REQUIRE(pixels > 0 && out != nullptr, "channel_absmax_fused: empty tensor or missing output");
  LHG_REQUIRE(!act_is_bf16(), "channel_absmax_fused: fp32 tensors only (the bf16 storage mode does not use it)");
  const ColMap cm = col_map(C);
  const FusedWs w = fused_ws(ws);
  const int nblk = partial_blocks(pixels, cm.gy);
  hipLaunchKernelGGL(channel_absmax_partial, dim3(nblk, cm.gy), dim3(256), 0, as_stream(s), x, pixels, C, ld, cm.lanes_c, cm.rows, w.cpart[0],
                     Fin{tickets, w.clevel1}, reinterpret_cast<unsigned*>(out));
  return check_launch("channel_absmax_fused");
}

int lhg_maxpool2x2_forward(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, lhg_stream_t s) {
  return LHG_ACT_CALL(maxpool_fwd_impl, x, N, H, W, C, ldx, y, ldy, s);
}
int lhg_maxpool2x2_backward(const float* x, int ldx, const float* gy, int ldgy, int N, int H, int W, int C, float* gx, int ldgx,
                            lhg_stream_t s) {
  return LHG_ACT_CALL(maxpool_bwd_impl, x, ldx, gy, ldgy, N, H, W, C, gx, ldgx, s);
}
int lhg_maxpool2x2_backward_add(const float* x, int ldx, const float* gy, int ldgy, int N, int H, int W, int C, const float* gadd, int ldgadd,
                                float* gx, int ldgx, lhg_stream_t s) {
  LHG_REQUIRE(gadd != nullptr, "maxpool2x2_backward_add: the tensor to add is missing");
  return LHG_ACT_CALL(maxpool_bwd_impl, x, ldx, gy, ldgy, N, H, W, C, gx, ldgx, s, gadd, ldgadd);
}
int lhg_act_backward(const float* g, int ldg, const float* y, int ldy, long long pixels, int C, int act, float slope, float* out,
                     int ldo, lhg_stream_t s) {
  return LHG_ACT_CALL(act_backward_impl, g, ldg, y, ldy, pixels, C, act, slope, out, ldo, s);
}

int lhg_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps, int step,
                  lhg_stream_t s) {
  return lhg_adam_step_scaled(p, g, m, v, n, lr, beta1, beta2, eps, step, 1.0f, nullptr, s);
}

int lhg_adam_step_scaled(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps, int step,
                         float grad_scale, const float* consts4, lhg_stream_t s) {
  LHG_REQUIRE(consts4 != nullptr || step >= 1, "adam_step: step must be >= 1");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((size_t)n, 256, 4096)), dim3(256), 0, as_stream(s), p, g, m, v, n, lr, beta1, beta2, eps,
                     (float)bc1, (float)sqrt(bc2), grad_scale, consts4);
  return check_launch("adam_step");
}

}  // extern "C"
