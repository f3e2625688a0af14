// Micro-benchmark: bf16 MFMA issue rate on random operands, 32x32x16 vs 16x16x32, one or two waves per SIMD, same output tile per
// wave (64x64) and the same FLOPs.  Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shape.cpp -o /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const bf16x8* __restrict__ in, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 65535]; b[i] = in[(tid * 8 + 4 + i) & 65535]; }
  float s = 0;
  if (SHAPE == 32) {
    f32x16 acc[2][2] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)      // 4 x (2x2 tiles) = 16 MFMAs of 32x32x16 = 64x64x64
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + kk) & 3], b[(j + kk) & 3], acc[i][j], 0, 0, 0);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  } else {
    f32x4 acc[4][4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)      // 2 x (4x4 tiles) = 32 MFMAs of 16x16x32 = 64x64x64
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + kk) & 3], b[(j + kk) & 3], acc[i][j], 0, 0, 0);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  }
  out[tid] = s;
}

int main() {
  std::vector<unsigned short> h(65536 * 8);
  srand(1);
  for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  bf16x8* din; float* dout;
  hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 2048 * 256 * 4);
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 40000;
  for (int waves = 1; waves <= 2; ++waves)
    for (int shape : {32, 16}) {
      const int blocks = 256 * waves;   // 256-thread blocks: 4 waves each -> `waves` waves per SIMD
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
        else hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, din, dout, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 2.0 * 64 * 64 * 64;
        if (rep == 2) printf("waves/SIMD %d shape %2d: %.3f ms  %.0f TFLOP/s\n", waves, shape, ms, flop / ms / 1e9);
      }
    }
  return 0;
}
