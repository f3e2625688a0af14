// Stand-alone use of the C ABI (include/lhg_hip.h) without Python or PyTorch: the caller owns every buffer and the stream, exactly as a
// C / C++ host of the reference's hot path would.  Build: see __graft_entry__.build();  run on an MI355X:  examples/abi_demo
//   1. y = relu(conv3x3(x, W) + b) through lhg_pack_weight + lhg_conv2d_forward, checked against a scalar CPU loop;
//   2. the fused angular-spectrum operator with no filter: crop(ifft2(fft2(pad(z)))) must return z.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/lhg_hip.h"

#define HIP_OK(e)                                                                  \
  do {                                                                             \
    hipError_t err_ = (e);                                                         \
    if (err_ != hipSuccess) {                                                      \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)
#define LHG_OK_OR_DIE(call)                                                        \
  do {                                                                             \
    if ((call) != 0) {                                                             \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, lhg_last_error());    \
      return 3;                                                                    \
    }                                                                              \
  } while (0)

static float frand(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xFFFF) / 65535.f - 0.5f;
}

int main() {
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  unsigned seed = 7;

  // ---------------------------------------------------------------- 1. convolution
  const int N = 2, H = 12, W = 20, Ci = 64, Co = 64, K = 3, rows_pad = 64;
  std::vector<float> x((size_t)N * H * W * Ci), w((size_t)Co * Ci * K * K), b(Co), y((size_t)N * H * W * Co), ref(y.size());
  for (auto& v : x) v = frand(seed);
  for (auto& v : w) v = frand(seed) * 0.1f;
  for (auto& v : b) v = frand(seed);
  for (int n = 0; n < N; ++n)
    for (int i = 0; i < H; ++i)
      for (int j = 0; j < W; ++j)
        for (int co = 0; co < Co; ++co) {
          double acc = b[co];
          for (int kh = 0; kh < K; ++kh)
            for (int kw = 0; kw < K; ++kw) {
              const int ii = i + kh - 1, jj = j + kw - 1;
              if (ii < 0 || ii >= H || jj < 0 || jj >= W) continue;
              for (int ci = 0; ci < Ci; ++ci)
                acc += (double)x[(((size_t)n * H + ii) * W + jj) * Ci + ci] * w[(((size_t)co * Ci + ci) * K + kh) * K + kw];
            }
          ref[(((size_t)n * H + i) * W + j) * Co + co] = acc > 0 ? (float)acc : 0.f;
        }
  float *dx, *dw, *db, *dy, *dwp, *dxmax;
  HIP_OK(hipMalloc(&dx, x.size() * 4));
  HIP_OK(hipMalloc(&dw, w.size() * 4));
  HIP_OK(hipMalloc(&db, b.size() * 4));
  HIP_OK(hipMalloc(&dy, y.size() * 4));
  HIP_OK(hipMalloc(&dwp, (size_t)lhg_packed_weight_floats(K * K, rows_pad, Ci) * 4));  // panel size depends on the precision mode
  HIP_OK(hipMalloc(&dxmax, LHG_ABSMAX_WORDS * 4));
  HIP_OK(hipMemcpyAsync(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(db, b.data(), b.size() * 4, hipMemcpyHostToDevice, stream));
  LHG_OK_OR_DIE(lhg_pack_weight(dw, Co, Ci, K, K, /*rows_from_d0=*/1, dwp, rows_pad, Ci, stream));
  HIP_OK(hipMemsetAsync(dxmax, 0, LHG_ABSMAX_WORDS * 4, stream));  // lhg_absmax max-accumulates into its slot
  LHG_OK_OR_DIE(lhg_absmax(dx, (long long)N * H * W, Ci, Ci, dxmax, stream));  // the tensor scale of the fp16-split GEMM mode (ignored by the others)
  LHG_OK_OR_DIE(lhg_conv2d_forward(dx, N, H, W, Ci, Ci, dwp, rows_pad, K, K, /*stride=*/1, dy, Co, Co, db, nullptr, nullptr, nullptr, 0,
                                   LHG_ACT_RELU, 0.f, /*planar_out=*/0, dxmax, /*y_absmax=*/nullptr, stream));
  HIP_OK(hipMemcpyAsync(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  double err = 0, mag = 0;
  for (size_t i = 0; i < y.size(); ++i) {
    err = std::fmax(err, std::fabs((double)y[i] - ref[i]));
    mag = std::fmax(mag, std::fabs((double)ref[i]));
  }
  std::printf("conv3x3 64->64 + bias + relu : max rel err %.2e\n", err / mag);
  if (!(err / mag < 2e-5)) return 1;

  // ---------------------------------------------------------------- 2. angular-spectrum operator, identity filter
  const int planes = 3, r0 = 48, c0 = 48, pad = 8, R = r0 + 2 * pad, C = c0 + 2 * pad;
  std::vector<float> z((size_t)planes * r0 * c0 * 2), out(z.size());
  for (auto& v : z) v = frand(seed);
  float *dz, *dout, *dws, *twr, *twc;
  const size_t ws_bytes = 2 * (size_t)planes * r0 * C * 8;
  HIP_OK(hipMalloc(&dz, z.size() * 4));
  HIP_OK(hipMalloc(&dout, z.size() * 4));
  HIP_OK(hipMalloc(&dws, ws_bytes));
  HIP_OK(hipMalloc(&twr, (size_t)R * 8));
  HIP_OK(hipMalloc(&twc, (size_t)C * 8));
  HIP_OK(hipMemcpyAsync(dz, z.data(), z.size() * 4, hipMemcpyHostToDevice, stream));
  LHG_OK_OR_DIE(lhg_fft_twiddles(twr, R, stream));
  LHG_OK_OR_DIE(lhg_fft_twiddles(twc, C, stream));
  LHG_OK_OR_DIE(lhg_asm_propagate(dz, nullptr, /*in_mode: complex*/ 2, 1.f, planes, r0, c0, pad, pad, nullptr, nullptr, nullptr, dout,
                                  /*out_mode: complex*/ 0, dws, ws_bytes, twr, twc, stream));
  HIP_OK(hipMemcpyAsync(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  err = 0;
  for (size_t i = 0; i < z.size(); ++i) err = std::fmax(err, std::fabs((double)out[i] - z[i]));
  std::printf("crop(ifft2(fft2(pad(z)))) == z   : max abs err %.2e\n", err);
  if (!(err < 1e-5)) return 1;
  std::printf("abi_demo ok (ABI version %d)\n", lhg_abi_version());
  return 0;
}
