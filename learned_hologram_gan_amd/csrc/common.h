// Shared helpers for the gfx950 kernels of liblhg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

#include "../../include/lhg_hip.h"

namespace lhg {

inline char* err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(LHG_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return LHG_OK;
}

inline hipStream_t as_stream(lhg_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// element type of every NHWC activation tensor the kernels see (lhg_set_activation_dtype): one instance per shared object
inline int& act_dtype() {
  static int d = LHG_DTYPE_F32;
  return d;
}
inline bool act_is_bf16() { return act_dtype() == LHG_DTYPE_BF16; }

constexpr int kWave = 64;  // CDNA wavefront

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Workgroups are dealt round-robin over the 8 XCDs (linear id % 8 labels the XCD group), each XCD with a private 4 MiB L2.
// Renumber them so that every XCD walks ONE contiguous range of the tile order: tiles that share input rows (the three rows of
// a 3x3 gather), an A panel (all n-tiles of one m-tile) or a K slab (all tiles of one weight-gradient split) then meet in the
// same L2 instead of being fetched once per XCD.  Bijective for any grid size; placement is a speed matter only.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned lin, unsigned total) {
  constexpr unsigned XCDS = 8;
  const unsigned x = lin % XCDS, j = lin / XCDS, q = total / XCDS, r = total % XCDS;
  return x * q + (x < r ? x : r) + j;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case LHG_ACT_RELU: return v > 0.f ? v : 0.f;
    case LHG_ACT_LEAKY: return v > 0.f ? v : v * slope;
    case LHG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    default: return v;
  }
}

// derivative of the activation expressed through the forward OUTPUT y
__device__ __forceinline__ float act_grad_from_output(float y, int act, float slope) {
  switch (act) {
    case LHG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case LHG_ACT_LEAKY: return y > 0.f ? 1.f : slope;
    case LHG_ACT_SIGMOID: return y * (1.f - y);
    default: return 1.f;
  }
}

}  // namespace lhg

#define LHG_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return lhg::fail(LHG_E_ARG, __VA_ARGS__); \
  } while (0)
