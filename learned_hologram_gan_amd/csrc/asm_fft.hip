// Fused angular-spectrum operator for gfx950:
//     out = crop( IFFT2( F1 (.) F2 (.) FFT2( pad( in ) ) ) )
// on `planes` independent rows0 x cols0 fields (the reference issues
// F.pad, torch.fft.fft2, complex multiplies, torch.fft.ifft2 and a slice as separate ATen ops on
// full rows x cols planes, angular_spectrum_method.py:374-392, 503-552).
//
// HBM-bound, so the design goal is bytes: the zero padding is never written or read.
//   pass 1  rows_forward : polar->complex, zero-extend a row in LDS, 1-D FFT along columns,
//                          only the rows0 non-zero rows exist            -> T1[plane][rows0][cols]
//   pass 2  cols_filter  : 16 adjacent columns per workgroup (128-byte segments), zero-extend to
//                          `rows` in LDS, FFT, multiply by the transfer-function slab(s),
//                          inverse FFT, keep the rows0 cropped rows      -> T2[plane][rows0][cols]
//   pass 3  rows_inverse : 1-D inverse FFT, crop to cols0, |z| / angle(z) / complex epilogue.
// Per plane-pair of 2-D transforms this moves ~11 MB at 384^2/1024^2 instead of the 67 MB of two
// dense fft2 calls (DESIGN.md has the accounting).  Transform lengths 256 and 1024 (the 192^2 / 384^2 frames with the
// reference's pads) run on register-resident two-step transforms (asm_cols_reg.inc: one LDS round trip per 1-D transform);
// every other product of the primes up to 13 (<= 4096: 2304 x 4096 for 4K, 832 = 2^6 13, 2800 = 2^4 5^2 7) uses the Stockham
// autosort radix-4 / 2 / 3 / 5 / 7 / 11 / 13 stages below (in LDS, host-exact twiddle table computed in double, in place with
// register staging: read-all / barrier / write-all); any other length up to 8192 (4976 = 2^4 311) runs as a Bluestein convolution
// on the same stages (length: bluestein_len — a power of two up to 4096, the shortest 2^a 3^b length above).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace lhg {

enum { IN_POLAR = 0, IN_PHASE = 1, IN_COMPLEX = 2 };
enum { OUT_COMPLEX = 0, OUT_ABS_ANGLE = 1, OUT_ABS = 2 };
enum { F_NONE = 0, F_MUL = 1, F_MUL_CONJ = 2, F_DIV = 3, F_DIV_CONJ = 4 };

constexpr int MAX_IT = 4;  // butterflies per thread per stage

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// nf transforms of length n at buf + f*stride.  tw: n entries exp(-2 pi i k / n) (LDS).
// All threads of the workgroup must call this (barriers inside).
// ---- power-of-two lengths: every stage parameter is a compile-time constant, so butterfly -> (transform, index, twiddle) is
// shifts and masks and the four element offsets are immediates.  The passes are bound by the butterflies' instruction stream
// (the generic code below spends two thirds of it on index arithmetic), not by LDS or HBM.
template <int N, int P, bool INV>
__device__ __forceinline__ void stage4_pow2(float2* buf, int nf, int stride, const float2* tw) {
  constexpr int T = N / 4, TWSTEP = N / (P * 4);
  constexpr int LT = __builtin_ctz(T);
  const int nthreads = blockDim.x, tid = threadIdx.x;
  const int total = nf * T;
  float2 u[MAX_IT][4];
  int off[MAX_IT];
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it) {
    const int b = tid + it * nthreads;
    if (b < total) {
      const int f = b >> LT, i = b & (T - 1), k = i & (P - 1);
      const float2* x = buf + f * stride + i;
      float2 u0 = x[0], u1 = x[T], u2 = x[2 * T], u3 = x[3 * T];
      if (P > 1) {
        float2 w1 = tw[k * TWSTEP], w2 = tw[2 * k * TWSTEP], w3 = tw[3 * k * TWSTEP];
        if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
        u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
      }
      const float2 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3), d = csub(u1, u3);
      const float2 v3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
      u[it][0] = cadd(v0, v2); u[it][1] = cadd(v1, v3); u[it][2] = csub(v0, v2); u[it][3] = csub(v1, v3);
      off[it] = f * stride + ((i - k) << 2) + k;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it) {
    if (tid + it * nthreads < total) {
      float2* y = buf + off[it];
      y[0] = u[it][0]; y[P] = u[it][1]; y[2 * P] = u[it][2]; y[3 * P] = u[it][3];
    }
  }
  __syncthreads();
}

template <int N, int P, bool INV>
__device__ __forceinline__ void stage2_pow2(float2* buf, int nf, int stride, const float2* tw) {
  constexpr int T = N / 2, TWSTEP = N / (P * 2);
  constexpr int LT = __builtin_ctz(T);
  const int nthreads = blockDim.x, tid = threadIdx.x;
  const int total = nf * T;
  float2 u[2 * MAX_IT][2];
  int off[2 * MAX_IT];
#pragma unroll
  for (int it = 0; it < 2 * MAX_IT; ++it) {
    const int b = tid + it * nthreads;
    if (b < total) {
      const int f = b >> LT, i = b & (T - 1), k = i & (P - 1);
      const float2* x = buf + f * stride + i;
      float2 u0 = x[0], u1 = x[T];
      float2 w = tw[k * TWSTEP];
      if (INV) w.y = -w.y;
      u1 = cmul(u1, w);
      u[it][0] = cadd(u0, u1);
      u[it][1] = csub(u0, u1);
      off[it] = f * stride + ((i - k) << 1) + k;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2 * MAX_IT; ++it) {
    if (tid + it * nthreads < total) {
      float2* y = buf + off[it];
      y[0] = u[it][0];
      y[P] = u[it][1];
    }
  }
  __syncthreads();
}

template <int N, int P, bool INV>
__device__ __forceinline__ void stages_pow2(float2* buf, int nf, int stride, const float2* tw) {
  if constexpr (N / P >= 4) {
    stage4_pow2<N, P, INV>(buf, nf, stride, tw);
    stages_pow2<N, P * 4, INV>(buf, nf, stride, tw);
  } else if constexpr (N / P == 2) {
    stage2_pow2<N, P, INV>(buf, nf, stride, tw);
  }
}

// ---- compile-time stages for a length that is not a power of two (2304 = 4^4 3^2, the row count of the 4K geometry): the same
// butterflies as the generic loop below with every stage parameter a constant, so transform / index / twiddle position come from
// multiply-shift divisions by constants instead of runtime integer divisions (the column pass of a 4K frame is bound by exactly that).
template <int N, int P, int R, bool INV>
__device__ __forceinline__ void stage_c(float2* buf, int nf, int stride, const float2* tw) {
  static_assert(R == 4 || R == 3, "radix");
  constexpr int T = N / R, TWSTEP = N / (P * R);
  const int nthreads = blockDim.x, tid = threadIdx.x;
  const int total = nf * T;
  float2 u[MAX_IT][R];
  int off[MAX_IT];
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it) {
    const int b = tid + it * nthreads;
    if (b < total) {
      const int f = b / T, i = b - f * T, k = i % P;
      const float2* x = buf + f * stride + i;
      float2 v[R];
#pragma unroll
      for (int j = 0; j < R; ++j) v[j] = x[j * T];
      if (P > 1) {
#pragma unroll
        for (int j = 1; j < R; ++j) {
          float2 w = tw[j * k * TWSTEP];
          if (INV) w.y = -w.y;
          v[j] = cmul(v[j], w);
        }
      }
      if constexpr (R == 4) {
        const float2 v0 = cadd(v[0], v[2]), v1 = csub(v[0], v[2]), v2 = cadd(v[1], v[3]), d = csub(v[1], v[3]);
        const float2 v3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
        u[it][0] = cadd(v0, v2); u[it][1] = cadd(v1, v3); u[it][2] = csub(v0, v2); u[it][3] = csub(v1, v3);
      } else {
        const float c3 = 0.8660254037844386f;
        const float2 t1 = cadd(v[1], v[2]);
        const float2 t2 = make_float2(v[0].x - 0.5f * t1.x, v[0].y - 0.5f * t1.y);
        const float2 d = csub(v[1], v[2]);
        const float2 t3 = INV ? make_float2(-c3 * d.y, c3 * d.x) : make_float2(c3 * d.y, -c3 * d.x);
        u[it][0] = cadd(v[0], t1); u[it][1] = cadd(t2, t3); u[it][2] = csub(t2, t3);
      }
      off[it] = f * stride + (i - k) * R + k;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < MAX_IT; ++it) {
    if (tid + it * nthreads < total) {
      float2* y = buf + off[it];
#pragma unroll
      for (int j = 0; j < R; ++j) y[j * P] = u[it][j];
    }
  }
  __syncthreads();
}

template <bool INV>
__device__ __forceinline__ void stages_2304(float2* buf, int nf, int stride, const float2* tw) {
  stage_c<2304, 1, 4, INV>(buf, nf, stride, tw);
  stage_c<2304, 4, 4, INV>(buf, nf, stride, tw);
  stage_c<2304, 16, 4, INV>(buf, nf, stride, tw);
  stage_c<2304, 64, 4, INV>(buf, nf, stride, tw);
  stage_c<2304, 256, 3, INV>(buf, nf, stride, tw);
  stage_c<2304, 768, 3, INV>(buf, nf, stride, tw);
}

// ---- radix-R stage for the odd primes 5, 7, 11, 13 (832 = 2^6 13, 2800 = 2^4 5^2 7: the padded extents of the CLI's default pad):
// the same Stockham step as the radix-4 / 2 / 3 stages with the R-point DFT written out (R^2 complex multiply-adds, the powers of
// w_R taken from the twiddle table at compile-time indices).  16 / R butterflies per thread, read-all / barrier / write-all.
template <int R, bool INV>
__device__ __forceinline__ void stage_prime(float2* buf, int n, int p, int nf, int stride, const float2* tw) {
  constexpr int ITS = 16 / R;
  const int nthreads = blockDim.x, tid = threadIdx.x;
  const int T = n / R, total = nf * T, twstep = n / (p * R), wstep = n / R;
  float2 w[R];
#pragma unroll
  for (int m = 0; m < R; ++m) {
    w[m] = tw[m * wstep];
    if (INV) w[m].y = -w[m].y;
  }
  float2 out[ITS][R];
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int b = tid + it * nthreads;
    if (b < total) {
      const int f = b / T, i = b - f * T, k = i % p;
      const float2* x = buf + f * stride + i;
      float2 u[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        u[j] = x[j * T];
        if (j > 0 && p > 1) {
          float2 t = tw[j * k * twstep];
          if (INV) t.y = -t.y;
          u[j] = cmul(u[j], t);
        }
      }
#pragma unroll
      for (int jp = 0; jp < R; ++jp) {
        float2 acc = u[0];
#pragma unroll
        for (int j = 1; j < R; ++j) acc = cadd(acc, cmul(u[j], w[(j * jp) % R]));
        out[it][jp] = acc;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int b = tid + it * nthreads;
    if (b < total) {
      const int f = b / T, i = b - f * T, k = i % p;
      float2* y = buf + f * stride + (i - k) * R + k;
#pragma unroll
      for (int jp = 0; jp < R; ++jp) y[jp * p] = out[it][jp];
    }
  }
  __syncthreads();
}

// PRIMES: the length may contain the factors 5 / 7 / 11 / 13.  A separate instantiation: with the prime stages compiled into it the
// kernel of the 2^a 3^b lengths (the 4K geometry 2304 x 4096) ran 15 % slower (register pressure of the 13-point butterfly).
template <bool INV, bool PRIMES = false>
__device__ __forceinline__ void lds_fft(float2* buf, int n, int nf, int stride, const float2* tw) {
  switch (n) {  // wave-uniform
    case 64: return stages_pow2<64, 1, INV>(buf, nf, stride, tw);
    case 128: return stages_pow2<128, 1, INV>(buf, nf, stride, tw);
    case 256: return stages_pow2<256, 1, INV>(buf, nf, stride, tw);
    case 512: return stages_pow2<512, 1, INV>(buf, nf, stride, tw);
    case 1024: return stages_pow2<1024, 1, INV>(buf, nf, stride, tw);
    case 2048: return stages_pow2<2048, 1, INV>(buf, nf, stride, tw);
    case 4096: return stages_pow2<4096, 1, INV>(buf, nf, stride, tw);
    case 2304: return stages_2304<INV>(buf, nf, stride, tw);
    case 8192: return stages_pow2<8192, 1, INV>(buf, nf, stride, tw);    // Bluestein convolution lengths of the large non-smooth extents
    case 16384: return stages_pow2<16384, 1, INV>(buf, nf, stride, tw);  // (512 / 1024 threads, twiddles read from global memory)
    default: break;
  }
  const int nthreads = blockDim.x, tid = threadIdx.x;
  for (int p = 1; p < n;) {
    const int rem = n / p;                                   // radix 4 while possible, then 2, then 3, then the odd primes up to 13
    const int R = (rem % 4 == 0) ? 4 : (rem % 2 == 0) ? 2 : (!PRIMES || rem % 3 == 0) ? 3 : (rem % 5 == 0) ? 5 : (rem % 7 == 0) ? 7 : (rem % 11 == 0) ? 11 : 13;
    if (PRIMES && R > 4) {  // wave-uniform
      switch (R) {
        case 5: stage_prime<5, INV>(buf, n, p, nf, stride, tw); break;
        case 7: stage_prime<7, INV>(buf, n, p, nf, stride, tw); break;
        case 11: stage_prime<11, INV>(buf, n, p, nf, stride, tw); break;
        default: stage_prime<13, INV>(buf, n, p, nf, stride, tw); break;
      }
      p *= R;
      continue;
    }
    const bool p_pow2 = (p & (p - 1)) == 0;
    const int T = n / R;
    const int total = nf * T;
    const int twstep = n / (p * R);
    const bool t_pow2 = (T & (T - 1)) == 0;  // wave-uniform: butterfly -> (transform, index) by shift instead of an integer division
    const int lt = __ffs(T) - 1;
    // butterfly b -> (line f, index i): nothing to divide for the one-line workgroups of the long (Bluestein) lines
    auto line_of = [&](int b_, int& f_, int& i_) {
      if (nf == 1) { f_ = 0; i_ = b_; }
      else { f_ = t_pow2 ? b_ >> lt : b_ / T; i_ = b_ - f_ * T; }
    };
    // i mod p without the integer division when p is not a power of two (radix-3 stages): float reciprocal and one correction (i < 2^24)
    const float inv_p = 1.f / (float)p;
    auto mod_p = [&](int i_) {
      if (p_pow2) return i_ & (p - 1);
      int k_ = i_ - (int)((float)i_ * inv_p) * p;
      if (k_ < 0) k_ += p;
      else if (k_ >= p) k_ -= p;
      return k_;
    };
    float2 u[MAX_IT][4];
    if (R == 4) {
#pragma unroll
      for (int it = 0; it < MAX_IT; ++it) {
        const int b = tid + it * nthreads;
        if (b < total) {
          int f, i;
          line_of(b, f, i);
          const int k = mod_p(i);
          const float2* x = buf + f * stride + i;
          float2 u0 = x[0], u1 = x[T], u2 = x[2 * T], u3 = x[3 * T];
          if (p > 1) {
            float2 w1 = tw[k * twstep], w2 = tw[2 * k * twstep], w3 = tw[3 * k * twstep];
            if (INV) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
            u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
          }
          const float2 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3), d = csub(u1, u3);
          const float2 v3 = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
          u[it][0] = cadd(v0, v2); u[it][1] = cadd(v1, v3); u[it][2] = csub(v0, v2); u[it][3] = csub(v1, v3);
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < MAX_IT; ++it) {
        const int b = tid + it * nthreads;
        if (b < total) {
          int f, i;
          line_of(b, f, i);
          const int k = mod_p(i);
          float2* y = buf + f * stride + ((i - k) << 2) + k;
          y[0] = u[it][0]; y[p] = u[it][1]; y[2 * p] = u[it][2]; y[3 * p] = u[it][3];
        }
      }
      __syncthreads();
    } else if (R == 2) {
      // radix-2 stage: 2*T == n, so up to 2*MAX_IT butterflies per thread
#pragma unroll
      for (int it = 0; it < 2 * MAX_IT; ++it) {
        const int b = tid + it * nthreads;
        if (b < total) {
          int f, i;
          line_of(b, f, i);
          const int k = mod_p(i);
          const float2* x = buf + f * stride + i;
          float2 u0 = x[0], u1 = x[T];
          float2 w = tw[k * twstep];
          if (INV) w.y = -w.y;
          u1 = cmul(u1, w);
          u[it >> 1][(it & 1) * 2] = cadd(u0, u1);
          u[it >> 1][(it & 1) * 2 + 1] = csub(u0, u1);
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 2 * MAX_IT; ++it) {
        const int b = tid + it * nthreads;
        if (b < total) {
          int f, i;
          line_of(b, f, i);
          const int k = mod_p(i);
          float2* y = buf + f * stride + ((i - k) << 1) + k;
          y[0] = u[it >> 1][(it & 1) * 2];
          y[p] = u[it >> 1][(it & 1) * 2 + 1];
        }
      }
      __syncthreads();
    } else {
      // radix-3 stage (2304 = 2^8 * 3^2 for the 4K geometry): w3 = -1/2 -+ i*sqrt(3)/2
      const float c3 = 0.8660254037844386f;
#pragma unroll
      for (int it = 0; it < MAX_IT; ++it) {
        const int b = tid + it * nthreads;
        if (b < total) {
          int f, i;
          line_of(b, f, i);
          const int k = mod_p(i);
          const float2* x = buf + f * stride + i;
          float2 u0 = x[0], u1 = x[T], u2 = x[2 * T];
          if (p > 1) {
            float2 w1 = tw[k * twstep], w2 = tw[2 * k * twstep];
            if (INV) { w1.y = -w1.y; w2.y = -w2.y; }
            u1 = cmul(u1, w1); u2 = cmul(u2, w2);
          }
          const float2 t1 = cadd(u1, u2);
          const float2 t2 = make_float2(u0.x - 0.5f * t1.x, u0.y - 0.5f * t1.y);
          const float2 d = csub(u1, u2);
          const float2 t3 = INV ? make_float2(-c3 * d.y, c3 * d.x) : make_float2(c3 * d.y, -c3 * d.x);
          u[it][0] = cadd(u0, t1); u[it][1] = cadd(t2, t3); u[it][2] = csub(t2, t3);
        }
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < MAX_IT; ++it) {
        const int b = tid + it * nthreads;
        if (b < total) {
          int f, i;
          line_of(b, f, i);
          const int k = mod_p(i);
          float2* y = buf + f * stride + (i - k) * 3 + k;
          y[0] = u[it][0]; y[p] = u[it][1]; y[2 * p] = u[it][2];
        }
      }
      __syncthreads();
    }
    p *= R;
  }
}

__device__ __forceinline__ float2 apply_filter(float2 z, float2 f, int op) {
  switch (op) {
    case F_MUL: return cmul(z, f);
    case F_MUL_CONJ: return cmul(z, make_float2(f.x, -f.y));
    case F_DIV: {
      const float inv = 1.f / (f.x * f.x + f.y * f.y);
      const float2 r = cmul(z, make_float2(f.x, -f.y));
      return make_float2(r.x * inv, r.y * inv);
    }
    case F_DIV_CONJ: {
      const float inv = 1.f / (f.x * f.x + f.y * f.y);
      const float2 r = cmul(z, f);
      return make_float2(r.x * inv, r.y * inv);
    }
    default: return z;
  }
}

// The same on U values at once with the operation chosen OUTSIDE the element loop: `apply_filter` inside an unrolled loop compiles to
// a run-time switch per element behind each element's own load and wait — one exposed memory latency per element (round 4: the
// column passes spent most of their time there).  Same arithmetic per element, so the same bits.
template <int U>
__device__ __forceinline__ void apply_filter_batch(float2 (&z)[U], const float2 (&f)[U], int op) {
  switch (op) {  // uniform
    case F_MUL:
#pragma unroll
      for (int u = 0; u < U; ++u) z[u] = apply_filter(z[u], f[u], F_MUL);
      break;
    case F_MUL_CONJ:
#pragma unroll
      for (int u = 0; u < U; ++u) z[u] = apply_filter(z[u], f[u], F_MUL_CONJ);
      break;
    case F_DIV:
#pragma unroll
      for (int u = 0; u < U; ++u) z[u] = apply_filter(z[u], f[u], F_DIV);
      break;
    case F_DIV_CONJ:
#pragma unroll
      for (int u = 0; u < U; ++u) z[u] = apply_filter(z[u], f[u], F_DIV_CONJ);
      break;
    default: break;
  }
}

// ---------------------------------------------------------------------------------------- Bluestein (chirp-z) lengths
// A length n outside 2^a 3^b (832 = 2^6 13: the 192^2 frame with the CLI's default pad 320) is transformed as a circular convolution of
// length m >= 2n - 1 (bluestein_len: a power of two, or 2^a 3^b above 4096) with the chirp c[j] = exp(i pi j^2 / n):
//     X[k] = conj(c[k]) * sum_j (x[j] conj(c[j])) c[k - j]            (n k = (j^2 + k^2 - (k - j)^2) / 2)
// = conj(c) . IFFT_m( FFT_m(x conj(c), zero-extended) . FFT_m(c wrapped) ): two Stockham transforms of length m on the same LDS line plus
// three pointwise passes.  The table lhg_fft_twiddles builds for such an n is [m twiddles of length m][n chirp values][m values of
// FFT_m(c wrapped) / m], all evaluated from exact integer j^2 mod 2n in double.  The inverse transform is conj(forward(conj(x))).
// tw: the m twiddles (LDS); tab: the table in global memory.  All threads of the workgroup call this.
template <bool INV>
__device__ __forceinline__ void bluestein_fft(float2* buf, int n, int m, int nf, int stride, const float2* tw, const float2* __restrict__ tab) {
  const float2* chirp = tab + m;
  const float2* bf = tab + m + n;
  const int tid = threadIdx.x, nth = blockDim.x;
  for (int i = tid; i < nf * m; i += nth) {
    const int f = i / m, j = i - f * m;
    float2 z = make_float2(0.f, 0.f);
    if (j < n) {
      z = buf[f * stride + j];
      if (INV) z.y = -z.y;
      const float2 c = chirp[j];
      z = cmul(z, make_float2(c.x, -c.y));
    }
    buf[f * stride + j] = z;
  }
  __syncthreads();
  lds_fft<false>(buf, m, nf, stride, tw);
  for (int i = tid; i < nf * m; i += nth) {
    const int f = i / m, j = i - f * m;
    buf[f * stride + j] = cmul(buf[f * stride + j], bf[j]);
  }
  __syncthreads();
  lds_fft<true>(buf, m, nf, stride, tw);
  for (int i = tid; i < nf * n; i += nth) {
    const int f = i / n, k = i - f * n;
    const float2 c = chirp[k];
    float2 z = cmul(buf[f * stride + k], make_float2(c.x, -c.y));
    if (INV) z.y = -z.y;
    buf[f * stride + k] = z;
  }
  __syncthreads();
}

// one 1-D transform per line: directly (m == 0) or through Bluestein's convolution of length m
template <bool INV, bool PRIMES>
__device__ __forceinline__ void fft_line(float2* buf, int n, int m, int nf, int stride, const float2* tw, const float2* __restrict__ tab) {
  if (m == 0) lds_fft<INV, PRIMES>(buf, n, nf, stride, tw);
  else bluestein_fft<INV>(buf, n, m, nf, stride, tw, tab);
}

extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

// ---------------------------------------------------------------------------------------- pass 1
// grid.x = ceil(planes*rows0 / nf); block 256 threads; LDS: tw[n] + nf*n complex
// MAXT: 256 for every line up to 4096 (the register budget the Stockham stages were tuned with), 1024 for the Bluestein lines above
template <int MAXT, bool PRIMES>
__global__ __launch_bounds__(MAXT) void rows_forward_kernel(const float* __restrict__ in_a, const float* __restrict__ in_b, int in_mode,
                                                            float phase_scale, int total_rows, int cols0, int pad_c, int n, int m, int nf,
                                                            int tw_in_lds, const float2* __restrict__ twg, float2* __restrict__ t1) {
  const int L = m ? m : n;  // LDS line length (Bluestein: the convolution length)
  float2* twl = reinterpret_cast<float2*>(lds_raw);
  float2* buf = tw_in_lds ? twl + L : twl;
  const float2* tw = tw_in_lds ? twl : twg;  // the longest lines leave no room for the twiddles: read them from global memory
  const int tid = threadIdx.x, nth = blockDim.x;
  if (tw_in_lds)
    for (int i = tid; i < L; i += nth) twl[i] = twg[i];
  for (int i = tid; i < nf * L; i += nth) buf[i] = make_float2(0.f, 0.f);
  __syncthreads();
  const int row0 = blockIdx.x * nf;
  // RU loads of every thread in flight before the first is used (a workgroup per CU with one load per thread leaves the memory latency
  // exposed; round 4): addresses are clamped to a valid element and the value is dropped when the row is past the end
  constexpr int RU = 4;
  auto polar = [&](float amp, float ph_raw) {
    float sn, cs;
    sincosf(ph_raw * phase_scale, &sn, &cs);
    return make_float2(amp * cs, amp * sn);
  };
  {
    const int total = nf * cols0;
    int i = tid;
    for (; i + (RU - 1) * nth < total; i += RU * nth) {
      float2 zc[RU];
      float va[RU], vb[RU];
      int dst[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int idx = i + u * nth, f = idx / cols0, x = idx - f * cols0;
        const bool ok = row0 + f < total_rows;
        const size_t o = ok ? (size_t)(row0 + f) * cols0 + x : 0;
        dst[u] = ok ? f * L + pad_c + x : -1;
        if (in_mode == IN_COMPLEX) zc[u] = reinterpret_cast<const float2*>(in_a)[o];
        else if (in_mode == IN_POLAR) { va[u] = in_a[o]; vb[u] = in_b[o]; }
        else { va[u] = 1.f; vb[u] = in_a[o]; }
      }
#pragma unroll
      for (int u = 0; u < RU; ++u)
        if (dst[u] >= 0) buf[dst[u]] = in_mode == IN_COMPLEX ? zc[u] : polar(va[u], vb[u]);
    }
    for (; i < total; i += nth) {
      const int f = i / cols0, x = i - f * cols0;
      const int row = row0 + f;
      if (row < total_rows) {
        const size_t o = (size_t)row * cols0 + x;
        buf[f * L + pad_c + x] = in_mode == IN_COMPLEX ? reinterpret_cast<const float2*>(in_a)[o]
                                                      : polar(in_mode == IN_POLAR ? in_a[o] : 1.f, in_mode == IN_POLAR ? in_b[o] : in_a[o]);
      }
    }
  }
  __syncthreads();
  fft_line<false, PRIMES>(buf, n, m, nf, L, tw, twg);
  for (int i = tid; i < nf * n; i += nth) {
    const int f = i / n;
    const int row = row0 + f;
    if (row < total_rows) t1[(size_t)row * n + (i - f * n)] = buf[f * L + (i - f * n)];
  }
}

// ---------------------------------------------------------------------------------------- pass 2
struct ColsParams {
  const float2* src; int src_rows, src_off;   // rows present in src (rows0 or R) and their offset inside R
  const int* src_index;                       // nullptr, or per output plane the plane of src it reads (several filters of one field: lhg_asm_propagate_shared)
  float2* dst; int dst_rows, dst_off;
  int planes, R, C, G;                        // G columns per workgroup
  int M;                                      // 0, or the Bluestein convolution length of R
  int tw_in_lds;                              // 0: the line leaves no room for the twiddles in LDS
  int do_fwd, do_inv;
  float scale;
  const float2* f1; const int* f1_index; int f1_op;
  const float2* f2; const int* f2_index; int f2_op;
  const float2* tw;                           // R entries
};

// (PRIMES: at most 512 threads, so that the 13-point butterflies keep their values in registers — with 1024 threads the 128-VGPR budget
// spills them to scratch)
// MAXT = 768 (round 5): six waves per SIMD for TWO 12-wave workgroups on a CU (<= 85 VGPRs, <= 80 KB of LDS each) — the 2304-point columns of the
// 4K frame: with one 1024-thread workgroup per CU (four columns + twiddles = 92 KB) the load, transform and store phases of a workgroup had
// nothing to overlap with.
template <bool PRIMES, int MAXT = (PRIMES ? 512 : 1024)>
__global__ __launch_bounds__(MAXT, MAXT == 768 ? 6 : 1) void cols_filter_kernel(const ColsParams p) {
  const int L = p.M ? p.M : p.R;
  float2* twl = reinterpret_cast<float2*>(lds_raw);
  float2* buf = p.tw_in_lds ? twl + L : twl;
  const float2* tw = p.tw_in_lds ? twl : p.tw;
  const int stride = L + 1;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int groups = p.C / p.G;
  // XCD-contiguous block order: a workgroup's G columns are G * 8 bytes of every row — half a 128-byte line at G = 8 — and the workgroup
  // that owns the other half is the next block: in the hardware's round-robin order it runs on ANOTHER XCD, whose L2 fetches the line again
  const int bx = (int)xcd_contiguous(blockIdx.x, gridDim.x);
  const int plane = bx / groups, c0 = (bx - plane * groups) * p.G;
  if (p.tw_in_lds)
    for (int i = tid; i < L; i += nth) twl[i] = p.tw[i];
  // load G columns (zero outside the stored rows)
  const int lg = __ffs(p.G) - 1, gm = p.G - 1;  // G is a power of two
  // (CU = 4 accesses of every thread in flight: one 1024-thread workgroup per CU with one load per thread is ~8 KB in flight, far
  //  below what the memory latency needs; addresses are clamped and the value selected, so no load sits behind a branch)
  constexpr int CU = 4;
  const int total = p.R * p.G;
  {
    const float2* srcp = p.src + (size_t)(p.src_index ? p.src_index[plane] : plane) * p.src_rows * p.C + c0;
    int i = tid;
    for (; i + (CU - 1) * nth < total; i += CU * nth) {
      float2 z[CU];
#pragma unroll
      for (int u = 0; u < CU; ++u) {
        const int idx = i + u * nth, g = idx & gm, sr = (idx >> lg) - p.src_off;
        const bool ok = sr >= 0 && sr < p.src_rows;
        const float2 l = srcp[(size_t)(ok ? sr : 0) * p.C + g];
        z[u] = ok ? l : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < CU; ++u) {
        const int idx = i + u * nth;
        buf[(idx & gm) * stride + (idx >> lg)] = z[u];
      }
    }
    for (; i < total; i += nth) {
      const int g = i & gm, r = i >> lg, sr = r - p.src_off;
      float2 z = make_float2(0.f, 0.f);
      if (sr >= 0 && sr < p.src_rows) z = srcp[(size_t)sr * p.C + g];
      buf[g * stride + r] = z;
    }
  }
  __syncthreads();
  if (p.do_fwd) fft_line<false, PRIMES>(buf, p.R, p.M, p.G, stride, tw, p.tw);
  const int s1 = p.f1_op ? (p.f1_index ? p.f1_index[plane] : 0) : 0;
  const int s2 = p.f2_op ? (p.f2_index ? p.f2_index[plane] : 0) : 0;
  if (p.f1_op || p.f2_op || p.scale != 1.f) {
    const float2* f1p = p.f1_op ? p.f1 + (size_t)s1 * p.R * p.C + c0 : nullptr;
    const float2* f2p = p.f2_op ? p.f2 + (size_t)s2 * p.R * p.C + c0 : nullptr;
    int i = tid;
    for (; i + (CU - 1) * nth < total; i += CU * nth) {
      float2 z[CU], fa[CU], fb[CU];
      if (p.f1_op) {
#pragma unroll
        for (int u = 0; u < CU; ++u) { const int idx = i + u * nth; fa[u] = f1p[(size_t)(idx >> lg) * p.C + (idx & gm)]; }
      }
      if (p.f2_op) {
#pragma unroll
        for (int u = 0; u < CU; ++u) { const int idx = i + u * nth; fb[u] = f2p[(size_t)(idx >> lg) * p.C + (idx & gm)]; }
      }
#pragma unroll
      for (int u = 0; u < CU; ++u) { const int idx = i + u * nth; z[u] = buf[(idx & gm) * stride + (idx >> lg)]; }
      if (p.f1_op) apply_filter_batch<CU>(z, fa, p.f1_op);
      if (p.f2_op) apply_filter_batch<CU>(z, fb, p.f2_op);
#pragma unroll
      for (int u = 0; u < CU; ++u) {
        const int idx = i + u * nth;
        buf[(idx & gm) * stride + (idx >> lg)] = make_float2(z[u].x * p.scale, z[u].y * p.scale);
      }
    }
    for (; i < total; i += nth) {
      const int g = i & gm, k = i >> lg;
      float2 z = buf[g * stride + k];
      if (p.f1_op) z = apply_filter(z, f1p[(size_t)k * p.C + g], p.f1_op);
      if (p.f2_op) z = apply_filter(z, f2p[(size_t)k * p.C + g], p.f2_op);
      z.x *= p.scale; z.y *= p.scale;
      buf[g * stride + k] = z;
    }
    __syncthreads();
  }
  if (p.do_inv) fft_line<true, PRIMES>(buf, p.R, p.M, p.G, stride, tw, p.tw);
  for (int i = tid; i < p.dst_rows * p.G; i += nth) {
    const int g = i & gm, r = i >> lg;
    p.dst[((size_t)plane * p.dst_rows + r) * p.C + c0 + g] = buf[g * stride + r + p.dst_off];
  }
}

#include "asm_cols_reg.inc"

// ---------------------------------------------------------------------------------------- pass 3
template <int MAXT, bool PRIMES>
__global__ __launch_bounds__(MAXT) void rows_inverse_kernel(const float2* __restrict__ t2, int total_rows, int cols0, int pad_c, int n, int m, int nf,
                                                            int tw_in_lds, const float2* __restrict__ twg, float* __restrict__ out_a,
                                                            float* __restrict__ out_b, float2* __restrict__ out_c, int out_mode) {
  const int L = m ? m : n;
  float2* twl = reinterpret_cast<float2*>(lds_raw);
  float2* buf = tw_in_lds ? twl + L : twl;
  const float2* tw = tw_in_lds ? twl : twg;
  const int tid = threadIdx.x, nth = blockDim.x;
  if (tw_in_lds)
    for (int i = tid; i < L; i += nth) twl[i] = twg[i];
  const int row0 = blockIdx.x * nf;
  {
    constexpr int RU = 4;  // loads of every thread in flight (see rows_forward_kernel)
    const int total = nf * n;
    int i = tid;
    for (; i + (RU - 1) * nth < total; i += RU * nth) {
      float2 z[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int idx = i + u * nth, f = idx / n;
        const bool ok = row0 + f < total_rows;
        const float2 l = t2[ok ? (size_t)(row0 + f) * n + (idx - f * n) : 0];
        z[u] = ok ? l : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < RU; ++u) {
        const int idx = i + u * nth, f = idx / n;
        buf[f * L + (idx - f * n)] = z[u];
      }
    }
    for (; i < total; i += nth) {
      const int f = i / n;
      const int row = row0 + f;
      buf[f * L + (i - f * n)] = row < total_rows ? t2[(size_t)row * n + (i - f * n)] : make_float2(0.f, 0.f);
    }
  }
  __syncthreads();
  fft_line<true, PRIMES>(buf, n, m, nf, L, tw, twg);
  for (int i = tid; i < nf * cols0; i += nth) {
    const int f = i / cols0, x = i - f * cols0;
    const int row = row0 + f;
    if (row >= total_rows) continue;
    const float2 z = buf[f * L + pad_c + x];
    const size_t o = (size_t)row * cols0 + x;
    if (out_mode == OUT_COMPLEX) {
      out_c[o] = z;
    } else {
      out_a[o] = hypotf(z.x, z.y);
      if (out_mode == OUT_ABS_ANGLE) out_b[o] = atan2f(z.y, z.x);
      if (out_c) out_c[o] = z;
    }
  }
}

// Bluestein table pieces (after the m twiddles): chirp c[j] = exp(i pi j^2 / n), j < n, with j^2 reduced mod 2n in integers
__global__ void chirp_kernel(float2* chirp, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const long long r = ((long long)j * j) % (2ll * n);
  double s, c;
  sincospi((double)r / (double)n, &s, &c);
  chirp[j] = make_float2((float)c, (float)s);
}

// bf = FFT_m(c wrapped: b[j] = c[j], b[m - j] = c[j], 0 elsewhere) / m.  One workgroup; LDS: m twiddles + m values.
__global__ __launch_bounds__(1024) void bluestein_filter_kernel(float2* tab, int n, int m) {
  float2* buf = reinterpret_cast<float2*>(lds_raw);
  const float2* tw = tab;  // global memory: one workgroup, once per length
  const float2* chirp = tab + m;
  for (int i = threadIdx.x; i < m; i += blockDim.x) {
    float2 b = make_float2(0.f, 0.f);
    if (i < n) b = chirp[i];
    else if (m - i < n) b = chirp[m - i];
    buf[i] = b;
  }
  __syncthreads();
  lds_fft<false>(buf, m, 1, m, tw);
  const float inv = 1.f / (float)m;
  for (int i = threadIdx.x; i < m; i += blockDim.x) tab[m + n + i] = make_float2(buf[i].x * inv, buf[i].y * inv);
}

__global__ void twiddle_kernel(float2* tw, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double s, c;
  sincospi(2.0 * (double)k / (double)n, &s, &c);
  tw[k] = make_float2((float)c, (float)(-s));
}

// ---------------------------------------------------------------------------------------- host
// butterfly inputs per thread and stage: 16 for radix 4 / 2 (MAX_IT butterflies), 12 for radix 3, R * (16 / R) for the odd primes
static int fft_budget(int n) {
  int b = 16;
  if (n % 3 == 0) b = std::min(b, 12);
  if (n % 5 == 0) b = std::min(b, 15);
  if (n % 7 == 0) b = std::min(b, 14);
  if (n % 11 == 0) b = std::min(b, 11);
  if (n % 13 == 0) b = std::min(b, 13);
  return b;
}
static bool has_prime_radix(int n) { return n % 5 == 0 || n % 7 == 0 || n % 11 == 0 || n % 13 == 0; }
// lengths transformed directly: products of 2, 3, 5, 7, 11, 13 in [16, 4096] that one 256-thread workgroup can hold as a row
static bool smooth_in_range(int n) {
  if (n < 16 || n > 4096 || n > fft_budget(n) * 256) return false;
  for (int q : {2, 3, 5, 7, 11, 13})
    while (n % q == 0) n /= q;
  return n == 1;
}
// threads of a row-pass workgroup: 256, more only for lines longer than 16 butterfly inputs per thread (Bluestein lengths above 4096)
static int rows_threads(int L) { return L <= 4096 ? 256 : (L <= 8192 ? 512 : 1024); }
// Bluestein convolution length m for an extent n the Stockham stages do not cover directly (0 = direct transform, or unsupported: check
// with length_supported): any m >= 2n - 1 the stages can transform.  Up to 4096 the power of two (specialised stages); above, the
// SHORTEST 2^a 3^b length one workgroup's threads can hold (12 butterfly inputs per thread) — 4976 columns of the 4K frame with the
// CLI's pad 320: 10368 = 2^7 3^4 instead of 16384 (83 KB of LDS instead of 128, 0.6 of the butterflies): 4.9 ms per A5 call against
// 6.1, and against 5.3 on the rocFFT route (tools/bench_bluestein.py).  LHG_BLUESTEIN_SMOOTH=0: powers of two only (measurements).
static int bluestein_len(int n) {
  if (smooth_in_range(n) || n < 16 || 2 * n - 1 > 16384) return 0;
  int m = 64;
  while (m < 2 * n - 1) m <<= 1;
  static const bool smooth = [] { const char* e = getenv("LHG_BLUESTEIN_SMOOTH"); return !e || atoi(e) != 0; }();
  if (smooth && m > 4096) {
    int best = m;
    for (int t = 3; t <= 243; t *= 3)
      for (int c = t; c < best; c <<= 1)
        if (c >= 2 * n - 1 && c % 64 == 0 && c <= 12 * rows_threads(c)) { best = c; break; }
    m = best;
  }
  return m;
}
static bool length_supported(int n) { return smooth_in_range(n) || bluestein_len(n) != 0; }

static int rows_nf(int n) {
  const int cap = fft_budget(n) * 256;
  return std::max(1, std::min(64, cap / n));
}
// twiddles go to LDS next to the lines when both fit
static bool tw_fits(size_t line_bytes, int L) { return line_bytes + (size_t)L * sizeof(float2) <= 150 * 1024; }
// Row passes: twiddles in LDS beside the lines only while that keeps >= `min_wg` workgroups on a CU (LHG_ASM_ROWS_MIN_WG, default 4).  A
// 4096-point row (the 4K frame) is one 32 KB line per 256-thread workgroup: with its 32 KB of twiddles in LDS as well a CU holds two
// workgroups — eight waves to hide the latency of 70 MB planes — without them four or five; the twiddles then come from L2.
static bool rows_tw_fits(size_t line_bytes, int L) {
  static const int min_wg = [] { const char* e = getenv("LHG_ASM_ROWS_MIN_WG"); return e ? atoi(e) : 4; }();
  const size_t with_tw = line_bytes + (size_t)L * sizeof(float2);
  if (!tw_fits(line_bytes, L)) return false;
  return min_wg <= 1 || with_tw * (size_t)min_wg <= 160 * 1024 || line_bytes * (size_t)min_wg > 160 * 1024;  // (no gain from dropping them if the lines alone do not allow min_wg)
}

static int set_dyn_lds(const void* fn, size_t bytes) {
  if (bytes > 160 * 1024) return fail(LHG_E_ARG, "asm: LDS request %zu exceeds 160 KiB", bytes);
  static_assert(MAX_IT == 4, "fft_budget assumes 4 butterflies per thread per stage");
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(LHG_E_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  return LHG_OK;
}

// LHG_ASM_LDS_PAD=<bytes> (diagnostics): every angular-spectrum launch requests at least this much dynamic LDS, so that a workgroup can be
// given a compute unit's LDS to itself (no co-resident workgroup with a large LDS footprint)
static size_t lds_request(size_t need) {
  static const size_t pad = [] { const char* e = getenv("LHG_ASM_LDS_PAD"); return e ? (size_t)atoll(e) : (size_t)0; }();
  return std::max(need, std::min<size_t>(pad, 160 * 1024));
}

static bool asm_reg_enabled() {
  static const bool reg = [] { const char* e = getenv("LHG_ASM_REG"); return e ? atoi(e) != 0 : true; }();
  return reg;
}

template <int N1, int NF>
static int run_rows_forward_reg(const float* in_a, const float* in_b, int in_mode, float phase_scale, int total_rows, int cols0, int pad_c,
                                const float* tw_cols, float2* t1, hipStream_t st) {
  const size_t lds = lds_request(RegFftLds<N1>::bytes(NF));
  int rc = set_dyn_lds(reinterpret_cast<const void*>(rows_fwd_reg_kernel<N1, NF>), lds);
  if (rc) return rc;
  hipLaunchKernelGGL((rows_fwd_reg_kernel<N1, NF>), dim3((total_rows + NF - 1) / NF), dim3(N1 * NF), lds, st, in_a, in_b, in_mode, phase_scale,
                     total_rows, cols0, pad_c, reinterpret_cast<const float2*>(tw_cols), t1);
  return check_launch("rows_forward");
}

template <int N1, int NF>
static int run_rows_inverse_reg(const float2* t2, int total_rows, int cols0, int pad_c, const float* tw_cols, float* out_a, float* out_b,
                                float* out_c, int out_mode, hipStream_t st) {
  const size_t lds = lds_request(RegFftLds<N1>::bytes(NF));
  int rc = set_dyn_lds(reinterpret_cast<const void*>(rows_inv_reg_kernel<N1, NF>), lds);
  if (rc) return rc;
  hipLaunchKernelGGL((rows_inv_reg_kernel<N1, NF>), dim3((total_rows + NF - 1) / NF), dim3(N1 * NF), lds, st, t2, total_rows, cols0, pad_c,
                     reinterpret_cast<const float2*>(tw_cols), out_a, out_b, reinterpret_cast<float2*>(out_c), out_mode);
  return check_launch("rows_inverse");
}

static int run_rows_forward(const float* in_a, const float* in_b, int in_mode, float phase_scale, int planes, int rows0, int cols0,
                            int pad_c, int cols, const float* tw_cols, float2* t1, hipStream_t st) {
  if (asm_reg_enabled() && cols == 1024) return run_rows_forward_reg<32, 8>(in_a, in_b, in_mode, phase_scale, planes * rows0, cols0, pad_c, tw_cols, t1, st);
  if (asm_reg_enabled() && cols == 256) return run_rows_forward_reg<16, 16>(in_a, in_b, in_mode, phase_scale, planes * rows0, cols0, pad_c, tw_cols, t1, st);
  const int m = bluestein_len(cols), L = m ? m : cols;
  const int nf = rows_nf(L);
  const int tw_lds = rows_tw_fits((size_t)nf * L * sizeof(float2), L);
  const size_t lds = lds_request((size_t)((tw_lds ? L : 0) + nf * L) * sizeof(float2));
  const int total_rows = planes * rows0;
  auto launch = [&](auto kernel) {
    int rc = set_dyn_lds(reinterpret_cast<const void*>(kernel), lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kernel, dim3((total_rows + nf - 1) / nf), dim3(rows_threads(L)), lds, st, in_a, in_b, in_mode, phase_scale, total_rows,
                       cols0, pad_c, cols, m, nf, tw_lds, reinterpret_cast<const float2*>(tw_cols), t1);
    return (int)LHG_OK;
  };
  int rc = L > 4096 ? launch(rows_forward_kernel<1024, false>) : (has_prime_radix(L) ? launch(rows_forward_kernel<256, true>) : launch(rows_forward_kernel<256, false>));
  if (rc) return rc;
  return check_launch("rows_forward");
}

static int run_rows_inverse(const float2* t2, int planes, int rows0, int cols0, int pad_c, int cols, const float* tw_cols, float* out_a,
                            float* out_b, float* out_c, int out_mode, hipStream_t st) {
  if (asm_reg_enabled() && cols == 1024) return run_rows_inverse_reg<32, 8>(t2, planes * rows0, cols0, pad_c, tw_cols, out_a, out_b, out_c, out_mode, st);
  if (asm_reg_enabled() && cols == 256) return run_rows_inverse_reg<16, 16>(t2, planes * rows0, cols0, pad_c, tw_cols, out_a, out_b, out_c, out_mode, st);
  const int m = bluestein_len(cols), L = m ? m : cols;
  const int nf = rows_nf(L);
  const int tw_lds = rows_tw_fits((size_t)nf * L * sizeof(float2), L);
  const size_t lds = lds_request((size_t)((tw_lds ? L : 0) + nf * L) * sizeof(float2));
  const int total_rows = planes * rows0;
  auto launch = [&](auto kernel) {
    int rc = set_dyn_lds(reinterpret_cast<const void*>(kernel), lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kernel, dim3((total_rows + nf - 1) / nf), dim3(rows_threads(L)), lds, st, t2, total_rows, cols0, pad_c, cols, m, nf, tw_lds,
                       reinterpret_cast<const float2*>(tw_cols), out_a, out_b, reinterpret_cast<float2*>(out_c), out_mode);
    return (int)LHG_OK;
  };
  int rc = L > 4096 ? launch(rows_inverse_kernel<1024, false>) : (has_prime_radix(L) ? launch(rows_inverse_kernel<256, true>) : launch(rows_inverse_kernel<256, false>));
  if (rc) return rc;
  return check_launch("rows_inverse");
}

template <int N1, int G>
static int run_cols_reg(ColsParams& p, hipStream_t st) {
  const size_t lds = lds_request(RegFftLds<N1>::bytes(G));
  int rc = set_dyn_lds(reinterpret_cast<const void*>(cols_reg_kernel<N1, G>), lds);
  if (rc) return rc;
  p.G = G;
  hipLaunchKernelGGL((cols_reg_kernel<N1, G>), dim3(p.planes * (p.C / G)), dim3(N1 * G), lds, st, p);
  return check_launch("cols_reg");
}

static int run_cols(ColsParams& p, hipStream_t st) {
  // column lengths 256 and 1024: the register-resident two-step transform (asm_cols_reg.inc); LHG_ASM_REG=0 keeps the Stockham kernel
  if (asm_reg_enabled() && p.C % 16 == 0) {
    if (p.R == 1024) return run_cols_reg<32, 8>(p, st);   // 76 KB of LDS: two workgroups per CU overlap each other's memory phases
    if (p.R == 256) return run_cols_reg<16, 16>(p, st);
  }
  // G columns per workgroup: 16 (128-byte segments) while G*R <= 16*threads keeps <= MAX_IT butterflies per thread
  p.M = bluestein_len(p.R);
  const int L = p.M ? p.M : p.R;  // LDS line length
  int G = 16;
  while (G > 1 && (size_t)G * (L + 1) * sizeof(float2) + (size_t)L * sizeof(float2) > 150 * 1024) G >>= 1;
  p.tw_in_lds = tw_fits((size_t)G * (L + 1) * sizeof(float2), L);
  int threads = has_prime_radix(L) ? 512 : 1024;
  const long long budget = fft_budget(L);
  while (G > 1 && (long long)G * L > budget * threads) G >>= 1;
  if ((long long)G * L > budget * threads) return fail(LHG_E_ARG, "asm: column length %d unsupported", p.R);
  while (G > 1 && p.C % G != 0) G >>= 1;
  while (threads > 64 && (long long)G * L <= 2ll * threads) threads >>= 1;  // small transforms: fewer idle waves
  p.G = G;
  p.tw_in_lds = tw_fits((size_t)G * (L + 1) * sizeof(float2), L);
  // two workgroups per CU where the default leaves one (direct 2^a 3^b lengths whose line set exceeds half the LDS): the largest G whose
  // lines fit 80 KB, twiddles from L2, <= 768 threads (LHG_ASM_COLS_2WG=0: the one-workgroup form, for A/B measurements)
  static const bool two_wg = [] { const char* e = getenv("LHG_ASM_COLS_2WG"); return !e || atoi(e) != 0; }();
  if (two_wg && p.M == 0 && !has_prime_radix(L) && ((size_t)G * (L + 1) + (p.tw_in_lds ? L : 0)) * sizeof(float2) > 80 * 1024) {
    int G2 = 16;
    while (G2 > 1 && ((size_t)G2 * (L + 1) * sizeof(float2) > 80 * 1024 || p.C % G2 != 0)) G2 >>= 1;
    const long long t2 = (((long long)G2 * L + budget - 1) / budget + 63) / 64 * 64;
    if (G2 >= 4 && t2 <= 768 && (size_t)G2 * (L + 1) * sizeof(float2) <= 80 * 1024) {
      p.G = G2;
      p.tw_in_lds = 0;
      const size_t lds2 = lds_request((size_t)G2 * (L + 1) * sizeof(float2));
      int rc = set_dyn_lds(reinterpret_cast<const void*>(cols_filter_kernel<false, 768>), lds2);
      if (rc) return rc;
      hipLaunchKernelGGL((cols_filter_kernel<false, 768>), dim3(p.planes * (p.C / G2)), dim3((unsigned)t2), lds2, st, p);
      return check_launch("cols_filter");
    }
  }
  const size_t lds = lds_request(((size_t)G * (L + 1) + (p.tw_in_lds ? L : 0)) * sizeof(float2));
  auto launch = [&](auto kernel) {
    int rc = set_dyn_lds(reinterpret_cast<const void*>(kernel), lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kernel, dim3(p.planes * (p.C / G)), dim3(threads), lds, st, p);
    return (int)LHG_OK;
  };
  int rc = has_prime_radix(L) ? launch(cols_filter_kernel<true, 512>) : launch(cols_filter_kernel<false, 1024>);
  if (rc) return rc;
  return check_launch("cols_filter");
}

static int check_geometry(int planes, int rows0, int cols0, int pad_r, int pad_c, const char* what) {
  const int R = rows0 + 2 * pad_r, C = cols0 + 2 * pad_c;
  LHG_REQUIRE(planes > 0 && rows0 > 0 && cols0 > 0 && pad_r >= 0 && pad_c >= 0, "%s: bad extents", what);
  LHG_REQUIRE(length_supported(R) && length_supported(C),
              "%s: padded extents %dx%d: each must be in [16,8192] (2^a 3^b 5^c 7^d 11^e 13^f up to 4096 directly, anything else as a Bluestein convolution)", what, R, C);
  return LHG_OK;
}

static void fill_filter(ColsParams& p, const lhg_asm_filter* f) {
  if (!f) return;
  p.f1 = reinterpret_cast<const float2*>(f->f1); p.f1_index = f->f1_index; p.f1_op = f->f1 ? f->f1_op : 0;
  p.f2 = reinterpret_cast<const float2*>(f->f2); p.f2_index = f->f2_index; p.f2_op = f->f2 ? f->f2_op : 0;
}

}  // namespace lhg

using namespace lhg;

extern "C" {

long long lhg_fft_table_floats(int n) {
  const int m = n > 0 ? bluestein_len(n) : 0;
  return m ? 2ll * (2 * m + n) : 2ll * std::max(n, 0);
}

int lhg_fft_twiddles(float* twiddle, int n, lhg_stream_t s) {
  LHG_REQUIRE(n > 0, "fft_twiddles: n must be positive");
  float2* tab = reinterpret_cast<float2*>(twiddle);
  const int m = bluestein_len(n);
  if (!m) {
    hipLaunchKernelGGL(twiddle_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(s), tab, n);
    return check_launch("twiddles");
  }
  // Bluestein table: [m twiddles of length m][n chirp values][FFT_m(wrapped chirp) / m]
  hipLaunchKernelGGL(twiddle_kernel, dim3((m + 255) / 256), dim3(256), 0, as_stream(s), tab, m);
  hipLaunchKernelGGL(chirp_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(s), tab + m, n);
  const size_t lds = (size_t)m * sizeof(float2);
  int rc = set_dyn_lds(reinterpret_cast<const void*>(bluestein_filter_kernel), lds);
  if (rc) return rc;
  hipLaunchKernelGGL(bluestein_filter_kernel, dim3(1), dim3(rows_threads(m)), lds, as_stream(s), tab, n, m);
  return check_launch("bluestein table");
}

int lhg_asm_propagate(const float* in_a, const float* in_b, int in_mode, float phase_scale, int planes, int rows0, int cols0, int pad_r,
                      int pad_c, const lhg_asm_filter* filt, float* out_a, float* out_b, float* out_complex, int out_mode, float* ws,
                      size_t ws_bytes, const float* twiddle_rows, const float* twiddle_cols, lhg_stream_t s) {
  return lhg_asm_propagate_shared(in_a, in_b, in_mode, phase_scale, planes, nullptr, planes, rows0, cols0, pad_r, pad_c, filt, out_a, out_b, out_complex,
                                  out_mode, ws, ws_bytes, twiddle_rows, twiddle_cols, s);
}

// `in_planes` input fields, `planes` outputs: output plane q = crop(ifft2(filter_q . fft2(pad(input plane plane_src[q])))).  The first pass
// (polar -> complex, row transforms) runs ONCE per input field, however many filters are applied to it: the reference's multi-distance
// __call__ (angular_spectrum_method.py:503-522) propagates every field to D planes — 8 at the 4K frame: 24 of its 27 row-transform planes
// were the same three transformed eight times (2.8 ms of a 55 ms frame).  plane_src == nullptr: in_planes == planes, one to one.
int lhg_asm_propagate_shared(const float* in_a, const float* in_b, int in_mode, float phase_scale, int in_planes, const int* plane_src, int planes,
                             int rows0, int cols0, int pad_r, int pad_c, const lhg_asm_filter* filt, float* out_a, float* out_b, float* out_complex,
                             int out_mode, float* ws, size_t ws_bytes, const float* twiddle_rows, const float* twiddle_cols, lhg_stream_t s) {
  int rc = check_geometry(planes, rows0, cols0, pad_r, pad_c, "asm_propagate");
  if (rc) return rc;
  LHG_REQUIRE(in_planes > 0 && (plane_src != nullptr || in_planes == planes), "asm_propagate: %d input planes for %d outputs need a plane_src index", in_planes, planes);
  const int R = rows0 + 2 * pad_r, C = cols0 + 2 * pad_c;
  const size_t plane_bytes = (size_t)rows0 * C * sizeof(float2);
  if (ws_bytes < (size_t)(in_planes + planes) * plane_bytes)
    return fail(LHG_E_WORKSPACE, "asm_propagate: workspace %zu < %zu", ws_bytes, (size_t)(in_planes + planes) * plane_bytes);
  float2* t1 = reinterpret_cast<float2*>(ws);
  float2* t2 = t1 + (size_t)in_planes * rows0 * C;
  hipStream_t st = as_stream(s);
  rc = run_rows_forward(in_a, in_b, in_mode, phase_scale, in_planes, rows0, cols0, pad_c, C, twiddle_cols, t1, st);
  if (rc) return rc;
  ColsParams p{};
  p.src_index = plane_src;
  p.src = t1; p.src_rows = rows0; p.src_off = pad_r; p.dst = t2; p.dst_rows = rows0; p.dst_off = pad_r;
  p.planes = planes; p.R = R; p.C = C; p.do_fwd = 1; p.do_inv = 1; p.scale = 1.f / ((float)R * (float)C);
  p.tw = reinterpret_cast<const float2*>(twiddle_rows);
  fill_filter(p, filt);
  rc = run_cols(p, st);
  if (rc) return rc;
  return run_rows_inverse(t2, planes, rows0, cols0, pad_c, C, twiddle_cols, out_a, out_b, out_complex, out_mode, st);
}

int lhg_asm_to_spectrum(const float* in_a, const float* in_b, int in_mode, float phase_scale, int planes, int rows0, int cols0, int pad_r,
                        int pad_c, const lhg_asm_filter* filt, float* spectrum, float* ws, size_t ws_bytes, const float* twiddle_rows,
                        const float* twiddle_cols, lhg_stream_t s) {
  int rc = check_geometry(planes, rows0, cols0, pad_r, pad_c, "asm_to_spectrum");
  if (rc) return rc;
  const int R = rows0 + 2 * pad_r, C = cols0 + 2 * pad_c;
  const size_t tbytes = (size_t)planes * rows0 * C * sizeof(float2);
  if (ws_bytes < tbytes) return fail(LHG_E_WORKSPACE, "asm_to_spectrum: workspace %zu < %zu", ws_bytes, tbytes);
  float2* t1 = reinterpret_cast<float2*>(ws);
  hipStream_t st = as_stream(s);
  rc = run_rows_forward(in_a, in_b, in_mode, phase_scale, planes, rows0, cols0, pad_c, C, twiddle_cols, t1, st);
  if (rc) return rc;
  ColsParams p{};
  p.src = t1; p.src_rows = rows0; p.src_off = pad_r; p.dst = reinterpret_cast<float2*>(spectrum); p.dst_rows = R; p.dst_off = 0;
  p.planes = planes; p.R = R; p.C = C; p.do_fwd = 1; p.do_inv = 0; p.scale = 1.f;
  p.tw = reinterpret_cast<const float2*>(twiddle_rows);
  fill_filter(p, filt);
  return run_cols(p, st);
}

int lhg_asm_from_spectrum(const float* spectrum, int planes, int rows0, int cols0, int pad_r, int pad_c, const lhg_asm_filter* filt,
                          float* out_a, float* out_b, float* out_complex, int out_mode, float* ws, size_t ws_bytes,
                          const float* twiddle_rows, const float* twiddle_cols, lhg_stream_t s) {
  return lhg_asm_from_spectrum_shared(spectrum, nullptr, planes, rows0, cols0, pad_r, pad_c, filt, out_a, out_b, out_complex, out_mode, ws, ws_bytes,
                                      twiddle_rows, twiddle_cols, s);
}

// output plane q = crop(ifft2(filter_q . spectrum[plane_src[q]])): several filters of one spectrum (plane_src == nullptr: one to one)
int lhg_asm_from_spectrum_shared(const float* spectrum, const int* plane_src, int planes, int rows0, int cols0, int pad_r, int pad_c,
                                 const lhg_asm_filter* filt, float* out_a, float* out_b, float* out_complex, int out_mode, float* ws, size_t ws_bytes,
                                 const float* twiddle_rows, const float* twiddle_cols, lhg_stream_t s) {
  int rc = check_geometry(planes, rows0, cols0, pad_r, pad_c, "asm_from_spectrum");
  if (rc) return rc;
  const int R = rows0 + 2 * pad_r, C = cols0 + 2 * pad_c;
  const size_t tbytes = (size_t)planes * rows0 * C * sizeof(float2);
  if (ws_bytes < tbytes) return fail(LHG_E_WORKSPACE, "asm_from_spectrum: workspace %zu < %zu", ws_bytes, tbytes);
  float2* t2 = reinterpret_cast<float2*>(ws);
  hipStream_t st = as_stream(s);
  ColsParams p{};
  p.src_index = plane_src;
  p.src = reinterpret_cast<const float2*>(spectrum); p.src_rows = R; p.src_off = 0; p.dst = t2; p.dst_rows = rows0; p.dst_off = pad_r;
  p.planes = planes; p.R = R; p.C = C; p.do_fwd = 0; p.do_inv = 1; p.scale = 1.f / ((float)R * (float)C);
  p.tw = reinterpret_cast<const float2*>(twiddle_rows);
  fill_filter(p, filt);
  rc = run_cols(p, st);
  if (rc) return rc;
  return run_rows_inverse(t2, planes, rows0, cols0, pad_c, C, twiddle_cols, out_a, out_b, out_complex, out_mode, st);
}

}  // extern "C"
