// Device helpers of wgrad6.hip (its own translation unit): the few definitions of the gather-GEMM include files that wg6_kernel.inc
// uses — fp16 vector types, the fp16 MFMA, buffer loads, the LDS barrier, the per-channel power-of-two scales, the exact two-term
// split, the transposed fragment read and the XCD-contiguous block order.  Same code as in gg2_kernel.inc / gg3s_kernel.inc /
// wg4s_kernel.inc / conv_engine.hip (those live inside conv_engine.hip's translation unit and are not visible here).
#pragma once
#include "common.h"

namespace lhg {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma_k16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ f32x4 buffer_load_f32x4(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voffset, (int)soffset, 0));
}

__device__ __forceinline__ void lds_barrier() {
  // all of this wave's LDS traffic retired, then the workgroup barrier; vmcnt is deliberately not waited for
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// power-of-two scale that brings a channel's largest element into [2^14, 2^15) (gg3s_kernel.inc): exact to apply and to undo
__device__ __forceinline__ int split_scale_exp(float amax) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  if (e == 0 || e == 255) return 127;
  return min(max(268 - e, 1), 253);
}
__device__ __forceinline__ float split_scale(float amax) { return __uint_as_float((unsigned)split_scale_exp(amax) << 23); }
__device__ __forceinline__ float split_unscale(float amax) { return __uint_as_float((unsigned)(254 - split_scale_exp(amax)) << 23); }

struct SplitF16 {
  f16x4 p[2];
};
// x = h0 + h1 + e, h0 = fp16(x), h1 = fp16(x - h0), |e| <= 2^-23 |x| while h1 is a normal fp16 (gg3s_kernel.inc)
__device__ __forceinline__ SplitF16 split2_f16(f32x4 v) {
  SplitF16 out;
  out.p[0] = __builtin_convertvector(v, f16x4);
  out.p[1] = __builtin_convertvector(v - __builtin_convertvector(out.p[0], f32x4), f16x4);
  return out;
}

// two transposed 4 x 16 blocks, 4 rows apart: element j of the result = row 8 h + j of this lane's column (wg4s_kernel.inc)
__device__ __forceinline__ f16x8 tr_read_frag_f16(const _Float16* p0, int row_stride_elems) {
  typedef s16x4 __attribute__((address_space(3))) * lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0 + 4 * row_stride_elems));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(f16x8, v);
}

// (xcd_contiguous: common.h)
struct Block3 { int x, y, z; };
__device__ __forceinline__ Block3 xcd_block3(bool enabled) {
  if (!enabled) return {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  const unsigned gx = gridDim.x, gy = gridDim.y;
  const unsigned v = xcd_contiguous(blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z), gx * gy * gridDim.z);
  const unsigned z = v / (gx * gy), rem = v - z * gx * gy;
  return {(int)(rem % gx), (int)(rem / gx), (int)z};
}

}  // namespace lhg
