// Minimal stand-alone form of the non-repeat of DESIGN.md §5: ONE process, two HIP streams.
//   stream A ("victim")   : every lane evaluates the same fused multiply-add chain twice — once with the PACKED fp32 instruction
//                           (v_pk_fma_f32, one VGPR pair per lane) and once with two scalar v_fma_f32 — and counts bit differences.
//   stream B ("co-tenant"): workgroups that do nothing but dependent v_mfma_f32_32x32x16_f16 on one accumulator tile (the order of the
//                           BM=64 gather-GEMM consumers), resident on the same CUs.
// IEEE fp32 fma is exact-rounded, so the two evaluations are bit-identical on a correct machine whatever else runs.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/pk_mfma.hip -o tools/micro/pk_mfma
//   tools/micro/pk_mfma [launches] [co-tenant: 0 none, 1 mfma, 2 plain-VALU spin] [victim rounds] [victim: 0 packed, 1 scalar-vs-scalar control]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

extern __shared__ unsigned lds[];

struct Report { unsigned bad, first_lane, first_round, first_k, got_lo, got_hi, want_lo, want_hi; };

template <int CONTROL>
__global__ __launch_bounds__(512) void victim_kernel(int rounds, Report* rep, unsigned salt) {
  const unsigned tid = threadIdx.x;
  lds[tid] = tid;  // hold the dynamic LDS like an FFT pass does
  f32x2 x[8], y[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    x[k] = f32x2{0.25f + 1e-3f * (float)((tid * 8u + k + salt) & 1023u), -0.5f + 7e-4f * (float)((tid * 5u + 3u * k + salt) & 1023u)};
    y[k] = f32x2{0.125f + 3e-4f * (float)((tid + 11u * k) & 511u), 0.375f - 2e-4f * (float)((tid * 3u + k) & 511u)};
  }
  const f32x2 c = f32x2{0.99609375f + 1e-6f * (float)(tid & 63u), -0.998046875f};
  unsigned bad = 0, f_round = 0, f_k = 0, g0 = 0, g1 = 0, w0 = 0, w1 = 0;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f32x2 p;
      float s0, s1;
      if (CONTROL) {
        float p0, p1;
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(p0) : "v"(x[k].x), "v"(c.x), "v"(y[k].x));
        asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(p1) : "v"(x[k].y), "v"(c.y), "v"(y[k].y));
        p = f32x2{p0, p1};
      } else {
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(x[k]), "v"(c), "v"(y[k]));
      }
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(x[k].x), "v"(c.x), "v"(y[k].x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(x[k].y), "v"(c.y), "v"(y[k].y));
      const unsigned a0 = __float_as_uint(p.x), a1 = __float_as_uint(p.y), b0 = __float_as_uint(s0), b1 = __float_as_uint(s1);
      if ((a0 != b0 || a1 != b1) && !bad++) { f_round = r; f_k = k; g0 = a0; g1 = a1; w0 = b0; w1 = b1; }
      x[k] = f32x2{s0, s1};  // continue from the scalar result: one damaged packed result is one event
    }
    if ((r & 15) == 15) __syncthreads();
  }
  if (bad) {
    Report* o = rep + blockIdx.x;
    if (atomicAdd(&o->bad, bad) == 0) { o->first_lane = tid; o->first_round = f_round; o->first_k = f_k; o->got_lo = g0; o->got_hi = g1; o->want_lo = w0; o->want_hi = w1; }
  }
}

__global__ __launch_bounds__(512) void mfma_cotenant(int iters, float* out) {
  const unsigned tid = threadIdx.x;
  lds[tid] = tid;
  f16x8 a0, a1, b0, b1;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a0[i] = (_Float16)(0.01f * (float)((tid + i) & 31u));
    a1[i] = (_Float16)(1e-4f * (float)((tid * 3u + i) & 31u));
    b0[i] = (_Float16)(0.02f * (float)((tid * 7u + i) & 15u));
    b1[i] = (_Float16)(2e-4f * (float)((tid * 5u + i) & 15u));
  }
  f32x16 acc = {0};
  for (int i = 0; i < iters; ++i) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] *= 0.5f;
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += acc[q];
  out[blockIdx.x * 512 + tid] = s;
}

__global__ __launch_bounds__(512) void valu_cotenant(int iters, float* out) {
  const unsigned tid = threadIdx.x;
  lds[tid] = tid;
  float acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 1e-3f * (float)(tid + q);
  for (int i = 0; i < iters * 6; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = __builtin_fmaf(acc[q], 0.999f, 1e-4f);
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += acc[q];
  out[blockIdx.x * 512 + tid] = s;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 2000, cot = argc > 2 ? atoi(argv[2]) : 1, rounds = argc > 3 ? atoi(argv[3]) : 2000;
  const int control = argc > 4 ? atoi(argv[4]) : 0;
  const int blocks = 256, lds_victim = 33792, lds_cot = 40960;
  Report* rep;
  float* out;
  hipMalloc(&rep, blocks * sizeof(Report));
  hipMemset(rep, 0, blocks * sizeof(Report));
  hipMalloc(&out, (size_t)blocks * 512 * sizeof(float));
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  for (int l = 0; l < launches; ++l) {
    if (cot == 1) hipLaunchKernelGGL(mfma_cotenant, dim3(blocks), dim3(512), lds_cot, sb, 600, out);
    if (cot == 2) hipLaunchKernelGGL(valu_cotenant, dim3(blocks), dim3(512), lds_cot, sb, 600, out);
    if (control) hipLaunchKernelGGL(victim_kernel<1>, dim3(blocks), dim3(512), lds_victim, sa, rounds, rep, (unsigned)l * 97u);
    else hipLaunchKernelGGL(victim_kernel<0>, dim3(blocks), dim3(512), lds_victim, sa, rounds, rep, (unsigned)l * 97u);
  }
  hipDeviceSynchronize();
  Report h[256];
  hipMemcpy(h, rep, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long bad = 0;
  for (auto& q : h) bad += q.bad;
  const double evals = (double)launches * blocks * 512 * rounds * 8;
  printf("{\"victim\": \"%s\", \"cotenant\": \"%s\", \"launches\": %d, \"packed_fma_evaluations\": %.3g, \"mismatches\": %llu", control ? "scalar (control)" : "v_pk_fma_f32",
         cot == 1 ? "mfma" : cot == 2 ? "valu" : "none", launches, evals, bad);
  for (int b = 0; b < blocks; ++b)
    if (h[b].bad) {
      printf(", \"first\": {\"block\": %d, \"lane\": %u, \"round\": %u, \"k\": %u, \"got\": [%u, %u], \"want\": [%u, %u]}", b, h[b].first_lane, h[b].first_round, h[b].first_k,
             h[b].got_lo, h[b].got_hi, h[b].want_lo, h[b].want_hi);
      break;
    }
  printf("}\n");
  return 0;
}
