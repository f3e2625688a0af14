// Canary for the two-process non-repeat: does anything a CO-RESIDENT workgroup of another process does reach this kernel's
// LDS or registers?  Every workgroup fills its dynamic LDS with an address pattern and 48 VGPRs per lane with a lane pattern, then keeps
// re-checking both for ~`spin` rounds (with barriers in between, like the FFT passes) and reports the first mismatches.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/canary.hip -o tools/micro/canary && tools/micro/canary [launches] [lds_bytes] [spin]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Report { unsigned lds_bad, reg_bad, first_lds_off, first_lds_val, first_reg_idx, first_reg_val, block, round; };

extern __shared__ unsigned lds[];

__global__ __launch_bounds__(256) void canary_kernel(int lds_words, int spin, Report* rep, unsigned salt) {
  const unsigned tid = threadIdx.x;
  for (int i = tid; i < lds_words; i += 256) lds[i] = (unsigned)i * 2654435761u ^ salt ^ blockIdx.x;
  unsigned r[48];
#pragma unroll
  for (int k = 0; k < 48; ++k) r[k] = (tid * 48u + k) * 40503u ^ salt;
  __syncthreads();
  unsigned lds_bad = 0, reg_bad = 0, f_off = 0, f_val = 0, f_idx = 0, f_rv = 0, f_round = 0;
  for (int s = 0; s < spin; ++s) {
    for (int i = tid; i < lds_words; i += 256) {
      const unsigned v = lds[i], want = (unsigned)i * 2654435761u ^ salt ^ blockIdx.x;
      if (v != want && !lds_bad++) { f_off = i; f_val = v; f_round = s; }
    }
#pragma unroll
    for (int k = 0; k < 48; ++k) {
      asm volatile("" : "+v"(r[k]));  // keep the value in a VGPR across the rounds
      const unsigned want = (tid * 48u + k) * 40503u ^ salt;
      if (r[k] != want && !reg_bad++) { f_idx = k; f_rv = r[k]; f_round = s; }
    }
    __syncthreads();
  }
  if (lds_bad || reg_bad) {
    Report* o = rep + blockIdx.x;
    if (atomicAdd(&o->lds_bad, lds_bad) == 0 && atomicAdd(&o->reg_bad, reg_bad) == 0) {
      o->first_lds_off = f_off; o->first_lds_val = f_val; o->first_reg_idx = f_idx; o->first_reg_val = f_rv; o->block = blockIdx.x; o->round = f_round;
    }
  }
}

// Cross-wave exchange, the pattern of an FFT stage: every round each thread writes its own LDS slots, barrier, reads slots written by the
// OTHER waves (rotated by round), barrier.  A barrier that releases early, or an LDS write that is not yet visible behind it, shows as a
// stale value (the previous round's salt).
__global__ __launch_bounds__(256) void exchange_kernel(int lds_words, int rounds, Report* rep, unsigned salt) {
  const unsigned tid = threadIdx.x;
  unsigned bad = 0, f_off = 0, f_val = 0, f_round = 0;
  const int per = lds_words / 256;
  for (int s = 0; s < rounds; ++s) {
    const unsigned rs = salt + 7919u * s;
    for (int k = 0; k < per; ++k) lds[k * 256 + tid] = (unsigned)(k * 256 + tid) * 2654435761u ^ rs;
    __syncthreads();
    const unsigned src = (tid + 64u * (1 + (s % 3)) + (s >> 2)) & 255u;  // a lane of another wave
    for (int k = 0; k < per; ++k) {
      const unsigned v = lds[k * 256 + src], want = (unsigned)(k * 256 + src) * 2654435761u ^ rs;
      if (v != want && !bad++) { f_off = k * 256 + src; f_val = v ^ ((unsigned)(k * 256 + src) * 2654435761u); f_round = s; }
    }
    __syncthreads();
  }
  if (bad) {
    Report* o = rep + blockIdx.x;
    if (atomicAdd(&o->lds_bad, bad) == 0) { o->first_lds_off = f_off; o->first_lds_val = f_val; o->block = blockIdx.x; o->round = f_round; o->first_reg_val = salt; }
  }
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 2000, lds_bytes = argc > 2 ? atoi(argv[2]) : 33792, spin = argc > 3 ? atoi(argv[3]) : 40;
  const int blocks = 48;
  Report* rep;
  hipMalloc(&rep, blocks * sizeof(Report));
  hipMemset(rep, 0, blocks * sizeof(Report));
  if (lds_bytes > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(canary_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const int mode = argc > 4 ? atoi(argv[4]) : 0;  // 0: hold-and-check canary, 1: cross-wave exchange
  if (mode == 1 && lds_bytes > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(exchange_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  for (int l = 0; l < launches; ++l) {
    if (mode == 1) hipLaunchKernelGGL(exchange_kernel, dim3(blocks), dim3(256), lds_bytes, 0, lds_bytes / 4, spin, rep, (unsigned)l * 97u + 1u);
    else hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), lds_bytes, 0, lds_bytes / 4, spin, rep, (unsigned)l * 97u);
  }
  hipDeviceSynchronize();
  std::vector<Report> h(blocks);
  hipMemcpy(h.data(), rep, blocks * sizeof(Report), hipMemcpyDeviceToHost);
  unsigned long long lb = 0, rb = 0;
  for (auto& q : h) { lb += q.lds_bad; rb += q.reg_bad; }
  printf("{\"mode\": %d, \"canary_launches\": %d, \"lds_bytes\": %d, \"lds_mismatches\": %llu, \"reg_mismatches\": %llu", mode, launches, lds_bytes, lb, rb);
  for (auto& q : h)
    if (q.lds_bad || q.reg_bad) { printf(", \"first\": {\"block\": %u, \"round\": %u, \"lds_off\": %u, \"lds_val\": %u, \"reg_idx\": %u, \"reg_val\": %u}", q.block, q.round, q.first_lds_off, q.first_lds_val, q.first_reg_idx, q.first_reg_val); break; }
  printf("}\n");
  return 0;
}
