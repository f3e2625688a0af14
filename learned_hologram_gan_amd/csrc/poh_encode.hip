// AP2POH tail: per-colour radially symmetric 3x3 stencil on Re and Im of the
// back-propagated field, amplitude normalisation by 1.01 * per-plane max, double-phase encoding
// on a unit checkerboard.  ref: AP2POH.py:105-116, utilities.py:53-66,
// neural_network_components.py:68-75.  Streaming kernels, 8-byte lanes on interleaved complex64.
#include "common.h"

namespace lhg {

__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
  // non-negative floats order like their bit patterns
  atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void symconv_field_kernel(const float2* __restrict__ field, int planes, int rows, int cols,
                                                            const float* __restrict__ taps, const float* __restrict__ bias,
                                                            float2* __restrict__ mod, float* __restrict__ plane_max) {
  const int plane = blockIdx.y;
  const int colour = plane % 3;
  const float w0 = taps[colour * 3 + 0], w1 = taps[colour * 3 + 1], w2 = taps[colour * 3 + 2], b = bias[colour];
  const float2* f = field + (size_t)plane * rows * cols;
  float local_max = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += gridDim.x * blockDim.x) {
    const int y = i / cols, x = i - y * cols;
    float2 acc = make_float2(b, b);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = y + dy;
      if ((unsigned)yy >= (unsigned)rows) continue;
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = x + dx;
        if ((unsigned)xx >= (unsigned)cols) continue;
        const float w = (dy == 0 && dx == 0) ? w0 : ((dy == 0 || dx == 0) ? w1 : w2);
        const float2 v = f[(size_t)yy * cols + xx];
        acc.x += w * v.x;
        acc.y += w * v.y;
      }
    }
    mod[(size_t)plane * rows * cols + i] = acc;
    local_max = fmaxf(local_max, hypotf(acc.x, acc.y));
  }
  // wave reduction then one atomic per wave
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local_max = fmaxf(local_max, __shfl_down(local_max, off, 64));
  if ((threadIdx.x & 63) == 0) atomic_max_nonneg(plane_max + plane, local_max);
}

__global__ __launch_bounds__(256) void double_phase_kernel(const float2* __restrict__ mod, const float* __restrict__ plane_max, int planes,
                                                           int rows, int cols, float* __restrict__ poh) {
  const int plane = blockIdx.y;
  const float denom = plane_max[plane] * 1.01f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows * cols; i += gridDim.x * blockDim.x) {
    const int y = i / cols, x = i - y * cols;
    const float2 z = mod[(size_t)plane * rows * cols + i];
    const float a = hypotf(z.x, z.y) / denom;
    const float phi = atan2f(z.y, z.x);
    const float ac = acosf(a);
    poh[(size_t)plane * rows * cols + i] = ((x + y) & 1) ? phi - ac : phi + ac;
  }
}

}  // namespace lhg

using namespace lhg;

extern "C" {

int lhg_symconv_field(const float* field, int planes, int rows, int cols, const float* taps, const float* bias, float* mod,
                      float* plane_max, lhg_stream_t s) {
  LHG_REQUIRE(planes > 0 && rows > 0 && cols > 0, "symconv_field: bad extents");
  const int bx = std::min(64, (rows * cols + 255) / 256);
  hipLaunchKernelGGL(symconv_field_kernel, dim3(bx, planes), dim3(256), 0, as_stream(s), reinterpret_cast<const float2*>(field), planes,
                     rows, cols, taps, bias, reinterpret_cast<float2*>(mod), plane_max);
  return check_launch("symconv_field");
}

int lhg_double_phase_encode(const float* mod, const float* plane_max, int planes, int rows, int cols, float* poh, lhg_stream_t s) {
  LHG_REQUIRE(planes > 0 && rows > 0 && cols > 0, "double_phase_encode: bad extents");
  const int bx = std::min(64, (rows * cols + 255) / 256);
  hipLaunchKernelGGL(double_phase_kernel, dim3(bx, planes), dim3(256), 0, as_stream(s), reinterpret_cast<const float2*>(mod), plane_max,
                     planes, rows, cols, poh);
  return check_launch("double_phase_encode");
}

}  // extern "C"
