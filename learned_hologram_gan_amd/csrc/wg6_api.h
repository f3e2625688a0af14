// Host interface of the tap-fused weight-gradient GEMM (wgrad6.hip) for the rest of the library (conv_engine.hip).
#pragma once
#include <cstddef>

#include "common.h"

namespace lhg {

// dW[tap][m][n] = sum over positions of strip[src(position, tap)][m] * point[position][n]   (wg6_kernel.inc)
struct Wg6Problem {
  const float* strip;  // [N][Hs][Ws][lds]  channels Cm
  const float* point;  // [N][gh][gw][ldp]  channels Cn
  const float* strip_cmax;  // per-channel max|.| (device): Cm / Cn floats
  const float* point_cmax;
  int N, Hs, Ws, Cm, lds;
  int gh, gw, Cn, ldp;
  int krows, nt, stride;  // taps = krows x nt: (3, 3, 1 | 2) conv 3x3, (1, 1, 1) conv 1x1, (2, 2, 2) transposed conv 2x2
  int dy0, dx0;           // strip pixel of (point pixel (gi, gj), kernel row r, tap t) = (gi * stride + dy0 + r, gj * stride + dx0 + t)
  int m_pad, n_pad;       // multiples of 64 covering Cm / Cn
};

struct Wg6Plan {
  int variant;  // index into the instantiation table, -1: not supported (the caller keeps the per-tap kernels)
  int S;        // K splits
  int fused;    // 1: the launch reduces its slabs itself (last arriver per tile, needs `tickets`), 0: lhg_wgrad_reduce follows
};

// variant / split count for a geometry (deterministic: the same on every rank and in every process; LHG_WG6_VARIANT / LHG_WG6_SPLITS override)
Wg6Plan wg6_plan(const Wg6Problem& q);
size_t wg6_slab_floats(const Wg6Problem& q, const Wg6Plan& plan);  // S * taps * m_pad * n_pad
int wg6_tickets(const Wg6Problem& q, const Wg6Plan& plan);        // 32-bit words the fused form needs zeroed in front of the launch
// launch the GEMM into `slabs`; with plan.fused also the reduction into grad[n][m][tap] (D1 = Cm) — `tickets` zero-filled
int wg6_launch(const Wg6Problem& q, const Wg6Plan& plan, float* slabs, unsigned* tickets, float* grad, int accumulate, hipStream_t st);
// sum of the slabs in split order into grad[n][m][tap] (the separate launch when plan.fused == 0)
int wg6_reduce(const float* slabs, int S, int T, int m_pad, int n_pad, float* grad, int Cn, int Cm, int accumulate, hipStream_t st);
int wg6_variant_count();
const char* wg6_variant_name(int v);

}  // namespace lhg
