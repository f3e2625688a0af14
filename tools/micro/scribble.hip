// Debug aid (not part of liblhg_hip.so): fill the LDS of every CU and a wide range of VGPRs with a pattern, so that a kernel launched
// next that reads LDS or registers it never wrote sees that pattern instead of what its own previous launch left there.
// tools/scribble_probe.py runs the train step with different patterns between all ABI calls and compares the results bit for bit.
#include <hip/hip_runtime.h>

#define MOV8(b) asm volatile("v_mov_b32 v" #b "0, %0\n v_mov_b32 v" #b "1, %0\n v_mov_b32 v" #b "2, %0\n v_mov_b32 v" #b "3, %0\n v_mov_b32 v" #b "4, %0\n v_mov_b32 v" #b "5, %0\n v_mov_b32 v" #b "6, %0\n v_mov_b32 v" #b "7, %0\n v_mov_b32 v" #b "8, %0\n v_mov_b32 v" #b "9, %0" ::"s"(pat) : "v" #b "0", "v" #b "1", "v" #b "2", "v" #b "3", "v" #b "4", "v" #b "5", "v" #b "6", "v" #b "7", "v" #b "8", "v" #b "9")

__global__ __launch_bounds__(256) void scribble_kernel(unsigned pat, int lds_words, unsigned* sink) {
  extern __shared__ unsigned lds[];
  for (int i = threadIdx.x; i < lds_words; i += blockDim.x) lds[i] = pat;
  MOV8(3); MOV8(4); MOV8(5); MOV8(6); MOV8(7); MOV8(8); MOV8(9); MOV8(10); MOV8(11); MOV8(12); MOV8(13); MOV8(14); MOV8(15); MOV8(16);
  MOV8(17); MOV8(18); MOV8(19); MOV8(20); MOV8(21); MOV8(22); MOV8(23); MOV8(24);
  __syncthreads();
  if (sink && lds[(threadIdx.x * 97) % lds_words] != pat) *sink = 1;  // keeps the stores alive
}

extern "C" int scribble(unsigned pat, void* stream) {
  static int lds_bytes = [] {
    int want = 160 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(scribble_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, want) != hipSuccess) {
      (void)hipGetLastError();
      want = 64 * 1024;
    }
    return want;
  }();
  static unsigned* sink = [] { unsigned* p = nullptr; (void)hipMalloc(&p, 4); (void)hipMemset(p, 0, 4); return p; }();
  // 4 rounds of one workgroup per CU (256 CUs): every CU's LDS and register file is visited
  hipLaunchKernelGGL(scribble_kernel, dim3(1024), dim3(256), lds_bytes, reinterpret_cast<hipStream_t>(stream), pat, lds_bytes / 4, sink);
  return (int)hipGetLastError() * 1000 + lds_bytes / 1024;
}
