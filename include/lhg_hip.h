/*
 * lhg_hip.h — C ABI of liblhg_hip.so, the MI355X (gfx950) implementation of the
 * RGBD -> phase-only-hologram hot path of WeijieXie/learned_hologram_gan.
 *
 * The reference has no FFI layer: its boundary is "Python module -> ATen op"
 * (SURVEY.md §8b).  Each entry point below therefore replaces one ATen call site of
 * the reference (cited as ref: <file>:<line> into the reference tree) and is what a
 * ctypes / cffi binding on the reference side would bind.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, ints, floats and a hipStream_t (passed as void*);
 *     no torch types.  The library never allocates or frees: every workspace is passed
 *     in, every launch goes to `stream`.  It synchronises in ONE place: the first launch
 *     of a GEMM geometry it has not seen drains the device (hipDeviceSynchronize) and times
 *     its tiling variants, five launches each, with HIP events on `stream`
 *     (lhg_autotune(0) / LHG_AUTOTUNE=0 turns that off; it is skipped while the stream is
 *     being captured, so warm the geometries up before capturing a graph).  With
 *     LHG_TUNE_CACHE=<file> the choices are appended to that file and reused by later processes.
 *   - threading: the mode switches, the autotune cache, the profiling timers and the
 *     last-error text are process-global and not locked.  One host thread per process
 *     calls the library, as in the reference (single-threaded, DataLoader(num_workers=0)).
 *   - activations are NHWC fp32 (bf16 after lhg_set_activation_dtype(LHG_DTYPE_BF16)).
 *     A tensor is (ptr, N, H, W, C, ld) where ld >= C is
 *     the distance in elements between consecutive pixels, so a channel slice of a
 *     wider buffer (concat-free UNet skips) is addressable.  C must be a multiple of
 *     32 for GEMM inputs (producers zero-pad: lhg_nchw_to_nhwc).
 *   - optical fields are planar (B,3,rows,cols) fp32 / interleaved complex64, exactly
 *     the reference's NCHW tensors.
 *   - return value: 0 = ok, otherwise an LHG_E_* code; lhg_last_error() gives the text.
 */
#ifndef LHG_HIP_H
#define LHG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LHG_ABI_VERSION 10

enum {
  LHG_OK = 0,
  LHG_E_ARG = 1,      /* bad shape / alignment / unsupported size */
  LHG_E_LAUNCH = 2,   /* hipGetLastError() != hipSuccess after a launch */
  LHG_E_WORKSPACE = 3 /* workspace too small */
};

enum { LHG_ACT_NONE = 0, LHG_ACT_RELU = 1, LHG_ACT_LEAKY = 2, LHG_ACT_SIGMOID = 3 };
enum { LHG_PRECISION_F32 = 0, LHG_PRECISION_BF16 = 1, LHG_PRECISION_F32_SPLIT = 2, LHG_PRECISION_F32_SPLIT2 = 3, LHG_PRECISION_F32_SPLIT_F16 = 4 };

enum { LHG_DTYPE_F32 = 0, LHG_DTYPE_BF16 = 1 };

typedef void* lhg_stream_t; /* hipStream_t */

int lhg_abi_version(void);
const char* lhg_last_error(void);

/* Per-launch timing of the two GEMM kernels (0: gather-GEMM gg_kernel, 1: wgrad-GEMM wg_kernel) with HIP
 * events recorded on the launch stream; used by bench.py for the roofline figures.  lhg_profile_read
 * synchronises the recorded events and returns the summed duration, the launch count and the executed
 * (padded-tile) flops since the last enable. */
/* The GEMM launcher times its tiling variants once per new geometry (idle device, five launches per variant between
 * HIP events on the caller's stream, skipped while the stream is being captured) and caches the fastest — in the
 * process, and in the file LHG_TUNE_CACHE names if set; every variant gives identical values.
 * With LHG_PROFILE_LOG=<file>, lhg_profile_read also appends one CSV line per timed launch (geometry, variant, flops, ms).
 * lhg_autotune(0) turns this off (a fixed heuristic is used instead); env LHG_AUTOTUNE=0 does the same. */
int lhg_autotune(int on);
int lhg_profile_enable(int kernel, int on);
int lhg_profile_read(int kernel, double* total_ms, long long* launches, double* executed_flops);

/* Element type of the NHWC ACTIVATION tensors (process-wide, default LHG_DTYPE_F32).  With LHG_DTYPE_BF16 (BASELINE configs[2], [4]:
 * "bf16", "bf16 + fp32 FFT") every `float*` argument that names an NHWC activation or activation-gradient tensor addresses bf16
 * elements instead (`ld` / `C` stay element counts); parameters, statistics, workspaces, weight gradients, NCHW tensors, planar
 * outputs and everything of the angular-spectrum / loss / optimiser kernels stay fp32, and all arithmetic is fp32.  Requires
 * LHG_PRECISION_BF16 for the conv GEMMs (their operands are read from HBM as bf16 without conversion). */
int lhg_set_activation_dtype(int dtype);
int lhg_get_activation_dtype(void);

/* ------------------------------------------------------------------ layout */
/* NCHW (planar) -> NHWC with `ld` floats per pixel; channels C..ld-1 are zero-filled.
 * Entry of the UNet / critic (ref: neural_network_components.py:303, discriminator.py:44). */
int lhg_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int ld, lhg_stream_t s);
/* NHWC (first C of ld channels) -> NCHW.  Adjoint of the above. */
int lhg_nhwc_to_nchw(const float* src, int ld, float* dst, int N, int C, int H, int W, lhg_stream_t s);

/* Operand precision of the gather-GEMM behind conv / conv-transpose forward and input-gradient (process-wide; see
 * lhg_default_conv_precision).  LHG_PRECISION_F32 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32).  LHG_PRECISION_BF16 (BASELINE configs[2], [4]): tensors stay fp32 in memory, the GEMM
 * rounds its operands to bf16 (nearest even) and accumulates in fp32 on v_mfma_f32_32x32x16_bf16; lhg_pack_weight then writes
 * bf16 panels into `dst`, so panels must be re-packed after a mode change.  The weight-gradient GEMMs round x and gy the same
 * way (lhg_conv2d_wgrad_splits depends on the mode).  Thin convolutions and every non-GEMM kernel are unaffected.
 *
 * LHG_PRECISION_F32_SPLIT: fp32 tensors, fp32-faithful results on the bf16 matrix pipe.  Every operand is split exactly into three
 * bf16 terms (x = x0 + x1 + x2) and a product is the sum of the six bf16 products a_i*b_j with i + j < 3 (each exact in fp32; the
 * dropped ones are below 2^-24 |a b|), accumulated in fp32: 6/16 of the exact kernel's matrix time, results within a few fp32 ulps
 * of the accumulated sum of the exact kernel.  Applies to the gather-GEMM and to the weight-gradient GEMMs; lhg_pack_weight writes
 * split panels of lhg_packed_weight_floats() floats.  LHG_PRECISION_F32_SPLIT2 keeps two terms (three products, ~2^-16 relative):
 * for measurements only.
 *
 * LHG_PRECISION_F32_SPLIT_F16: fp32 tensors, fp32-faithful results from TWO fp16 terms per operand and THREE products.  fp16 holds
 * 11 significand bits: x*s = h0 + h1 + e with |e| <= 2^-23 |x*s|, and a0*b0 + a0*b1 + a1*b0 misses a*b by ~2^-23 |a b| — inside
 * the rounding of the fp32 accumulation it is summed into (measured against float64: the same error as the exact fp32 kernels) —
 * at half the matrix work of LHG_PRECISION_F32_SPLIT.  fp16's 5-bit exponent needs a power-of-two scale s per tensor,
 * s = 2^(14 - floor(log2 max|x|)): the caller measures max|x| of each activation operand with lhg_absmax and passes the device
 * pointer (`*_absmax` arguments below, ignored in every other mode); lhg_pack_weight measures max|w| itself and keeps it behind
 * the panels.  Dynamic range inside one tensor (ABI 4): the residual plane of a GATHERED activation operand (forward / input-gradient
 * GEMMs) is stored 2^11 times larger and multiplied with b0 2^-11, so activations keep their full relative accuracy down to 2^-29 max|x|
 * (an absolute error below 2^-50 max|x| under that); weights keep it down to 2^-18 max|w| (2^-39 max|w| absolute under that).  The
 * weight-gradient GEMMs contract over pixels, where a power-of-two scale PER CHANNEL of either operand factors out of the sum exactly:
 * their `x_absmax` / `gy_absmax` arguments are per-channel vectors (lhg_channel_absmax), so a channel that sits far below the rest of
 * its tensor loses nothing.  A non-finite maximum leaves the tensor (channel) unscaled (non-finite values then propagate as in fp32). */
int lhg_set_conv_precision(int precision);
int lhg_get_conv_precision(void);
/* the mode the library starts in: LHG_PRECISION_F32_SPLIT_F16 unless the environment variable LHG_CONV_PRECISION
 * (fp32 | fp32_split | fp32_split2 | fp32_split_f16 | bf16) names another one */
int lhg_default_conv_precision(void);
/* floats the caller must allocate for lhg_pack_weight's `dst` in the current precision mode */
long long lhg_packed_weight_floats(int taps, int rows_pad, int k_pad);
/* An operand's `*_absmax` argument points to LHG_ABSMAX_WORDS floats whose MAXIMUM is (an upper bound of) max|x|.  One word today:
 * spreading the producers' atomics over 16 words was measured and is slower (each word ramps up from zero on its own). */
#define LHG_ABSMAX_WORDS 1
/* out[0] = max(out[0], max |x|) over the first C of ld channels of `pixels` pixels (fp32 tensors; stream-ordered, no host
 * synchronisation).  The caller zero-fills the LHG_ABSMAX_WORDS floats of `out` first (one fill can prepare many slots); several calls on one slot give the
 * maximum over several tensors.  The tensor-scale input of LHG_PRECISION_F32_SPLIT_F16: measure each GEMM operand once and pass
 * `out` as its `*_absmax`; any upper bound of max|x| is valid, a loose one only narrows the window of full relative accuracy. */
int lhg_absmax(const float* x, long long pixels, int C, int ld, float* out, lhg_stream_t s);
/* out[c] = max over pixels |x[p][c]|, c < C (C and ld multiples of 4, x 16-byte aligned, fp32 tensors; `out` is overwritten;
 * ws: >= 2048*C floats; two launches, no atomics).  The per-channel scales of the weight-gradient GEMMs in the LHG_PRECISION_F32_SPLIT_F16 mode:
 * pass `out` as the `x_absmax` (Ci floats) / `gy_absmax` (Co floats) of lhg_conv2d_backward_weight / lhg_conv_transpose2x2_backward_weight.
 * Any per-channel upper bound is valid. */
int lhg_channel_absmax(const float* x, long long pixels, int C, int ld, float* out, float* ws, lhg_stream_t s);
/* The same maxima WITHOUT a pass of their own (ABI 5): the kernels that write most weight-gradient operands — lhg_bn_apply (a conv's
 * input) and lhg_bn_backward (a conv's output gradient) — leave per-workgroup partial maxima of what they write in
 * `*_chanmax_partial` (lhg_chanmax_partial_rows(pixels, C) x C floats, overwritten) through lhg_bn_apply_chanmax /
 * lhg_bn_backward_chanmax below; lhg_channel_absmax_finish(partial, pixels, C, out) reduces them to `out` (C floats) on any stream
 * ordered behind the producer — one small launch where lhg_channel_absmax reads the whole tensor again (a train step at 384 x 384,
 * batch 4: 77 of the 98 operands' maxima come this way, 1.5 GB of re-reads less). */
long long lhg_chanmax_partial_rows(long long pixels, int C);
int lhg_channel_absmax_finish(const float* partial, long long pixels, int C, float* out, lhg_stream_t s);

/* Pack a PyTorch 4-D weight w[D0][D1][KH][KW] into GEMM panels dst[KH*KW][rows_pad][k_pad],
 * K contiguous, zero padded.  rows_from_d0 = 1: rows = D0, K = D1 (Conv2d forward,
 * ConvTranspose2d dgrad);  0: rows = D1, K = D0 (Conv2d dgrad, ConvTranspose2d forward).
 * ref: F.conv2d / nn.LazyConv2d weights (OIHW) neural_network_components.py:9-19,
 *      nn.LazyConvTranspose2d weights (IOHW) :270-286. */
int lhg_pack_weight(const float* w, int D0, int D1, int KH, int KW, int rows_from_d0,
                    float* dst, int rows_pad, int k_pad, lhg_stream_t s);
/* The same for `n` weights at once (a HOST array of descriptors, read before the call returns): what a model needs after every
 * optimiser step (ref: the Adam steps of watermelon.py:137-138 touch every conv weight of the stepped network).  In
 * LHG_PRECISION_F32_SPLIT_F16 the fill of the max|w| words, the max|w| passes and the packs of up to 64 weights share three launches
 * instead of three per weight; the other modes pack one weight per launch.  Each `dst` holds lhg_packed_weight_floats() floats;
 * results are identical to n lhg_pack_weight calls. */
typedef struct lhg_pack_item {
  const float* w;
  float* dst;
  int D0, D1, KH, KW, rows_from_d0, rows_pad, k_pad;
} lhg_pack_item;
int lhg_pack_weights(const lhg_pack_item* items, int n, lhg_stream_t s);

/* ------------------------------------------------------------------ convolution family
 * One implicit-GEMM engine (MFMA v_mfma_f32_32x32x2_f32, LDS-tiled, no im2col buffer).
 * `wp` is a packed panel set from lhg_pack_weight.  Epilogue, in this order:
 *   v = acc + bias[c];  v = v*scale[c] + shift[c];  v += res[pixel][c];  v = act(v)
 * (each pointer may be NULL).  planar_out != 0 stores NCHW instead of NHWC.
 * `x_absmax` / `gy_absmax`: device pointer to max|.| of that operand (lhg_absmax), required in the
 * LHG_PRECISION_F32_SPLIT_F16 mode, ignored (may be NULL) in every other.  `y_absmax` (forward calls, may be NULL):
 * the fp16-split kernels max-accumulate max|y| of an NHWC output into it — lhg_absmax for the next GEMM's operand
 * without a pass of its own (several launches may share one slot: the halves of a concatenation buffer).          */

/* y = conv2d(x, W, stride, padding=KH/2).  ref: neural_network_components.py:27-30
 * (3x3 s1, 1x1), discriminator.py:34-38 (3x3 s1/s2), :25 (1024->1 head). */
int lhg_conv2d_forward(const float* x, int N, int H, int W, int Ci, int ldx,
                       const float* wp, int rows_pad, int KH, int KW, int stride,
                       float* y, int Co, int ldy,
                       const float* bias, const float* scale, const float* shift,
                       const float* res, int ldres, int act, float slope, int planar_out,
                       const float* x_absmax, float* y_absmax, lhg_stream_t s);

/* ABI 10: lhg_conv2d_forward (bias only: no affine, residual, activation; NHWC) for a convolution whose output goes straight into a
 * train-mode BatchNorm (ref: neural_network_components.py:27-30 conv -> bn, discriminator.py:34-39): the GEMM's epilogue leaves the
 * FIRST STAGE of the batch statistics behind — `stat_partial[row][0][c]` = sum of (y - bias[c]), `[row][1][c]` = sum of (y - bias[c])^2
 * over the output pixels of one consumer-wave row of the chosen tiling (fixed order: bit-repeatable for a given tiling choice; which
 * tiling the autotuner picks decides how the pixels are grouped) — and lhg_bn_stats_finish folds the rows in double: no statistics
 * pass over y.  `stat_partial`: >= lhg_conv2d_stats_rows_bound(N, Ho, Wo) * 2 * Co floats; `*stat_rows` (host) receives the rows
 * written, 0 when the launch ran a kernel without this epilogue (then call lhg_bn_stats).  Every arithmetic mode. */
long long lhg_conv2d_stats_rows_bound(int N, int Ho, int Wo);
int lhg_conv2d_forward_stats(const float* x, int N, int H, int W, int Ci, int ldx,
                             const float* wp, int rows_pad, int KH, int KW, int stride,
                             float* y, int Co, int ldy, const float* bias,
                             const float* x_absmax, float* y_absmax,
                             float* stat_partial, int* stat_rows, lhg_stream_t s);

/* ABI 10: lhg_conv2d_forward (3x3, stride 1, at most 64 output channels, NHWC fp32, LHG_PRECISION_F32_SPLIT_F16) whose residual is not a
 * tensor but a 1x1 convolution of a thin NCHW tensor, evaluated by the epilogue:
 *   v = act((acc + bias)*scale + shift + res_b[c] + sum_{k < res_c} res_x_nchw[n][k][h][w] * res_w[c*res_c + k])
 * — the shortcut `convolution_layer_3(X)` of the generator's first ResidualBlock (neural_network_components.py:22-31: X is the 4-channel
 * RGBD frame) without ever writing it: 2.1 GB written and read back per 4K frame otherwise.  The multiply-adds run in the order of the
 * thin-input kernel (bias first, channels ascending), so the stored bits are those of lhg_conv2d_thin_forward + lhg_conv2d_forward(res).
 * res_c in 1..4; images of at least 512 pixels; act NONE / RELU / LEAKY. */
int lhg_conv2d_forward_thin_res(const float* x, int N, int H, int W, int Ci, int ldx,
                                const float* wp, int rows_pad, int KH, int KW, int stride,
                                float* y, int Co, int ldy, const float* bias, const float* scale, const float* shift,
                                const float* res_x_nchw, int res_c, const float* res_w, const float* res_b, int act, float slope,
                                const float* x_absmax, float* y_absmax, lhg_stream_t s);

/* ABI 10, split K.  A gather-GEMM launch (lhg_conv2d_forward[_stats], lhg_conv2d_backward_input*) that cannot fill the chip — at most 160
 * output tiles of 128 x 128 and at least 64 K steps: the UNet's 24^2 x 1024-channel bottleneck (neural_network_components.py:246-250) —
 * cuts its K axis into 2 - 4 ranges of whole 32-channel chunks, a function of the geometry alone (LHG_PRECISION_F32_SPLIT_F16 mode).
 * lhg_gather_gemm_splitk_floats: the workspace such a launch needs, 0 when it does not split — M output pixels of the launch
 * (N*Ho*Wo; input-gradient: N*H*W), M_padded = N*H*(W+2) for a 3x3 stride-1 launch (else 0), rows_pad of its packed weight, K = the
 * gathered tensor's (padded) channels, taps = KH*KW.  lhg_gather_gemm_workspace hands a buffer to the NEXT gather-GEMM launch
 * (consumed by it, used or not; process-wide like the library's other switches: one host thread drives the library).  Without one the library keeps a grow-only buffer per stream, which cannot grow inside a graph
 * capture: callers that capture pass their own. */
long long lhg_gather_gemm_splitk_floats(long long M, long long M_padded, int rows_pad, int K, int taps);
int lhg_gather_gemm_workspace(float* ws, long long floats);

/* gx = conv2d_backward_input(gy, W).  (H, W) are the INPUT extents of the forward conv.
 * `wp` packed with rows_from_d0 = 0.  Replaces the autograd node of the call sites above;
 * also the "double backward w.r.t. gy" of lhg_conv2d_forward.  ref: watermelon.py:466-473. */
int lhg_conv2d_backward_input(const float* gy, int N, int H, int W, int Co, int ldgy,
                              const float* wp, int rows_pad, int KH, int KW, int stride,
                              float* gx, int Ci, int ldgx, const float* gy_absmax, lhg_stream_t s);
/* gx = conv2d_backward_input(gy, W) + res: the sum autograd forms when the conv's input has a second consumer (the skip path of
 * ResidualBlock, ref: neural_network_components.py:22-31: `X` feeds convolution_layer_1 and convolution_layer_3 / the identity),
 * taken in the GEMM epilogue instead of by a separate pass over both gradients.  `res` (gx's pixels, ldres floats apart, >= Ci
 * channels) may be NULL. */
int lhg_conv2d_backward_input_add(const float* gy, int N, int H, int W, int Co, int ldgy,
                                  const float* wp, int rows_pad, int KH, int KW, int stride,
                                  float* gx, int Ci, int ldgx, const float* res, int ldres,
                                  const float* gy_absmax, lhg_stream_t s);

/* Partial weight gradients: slabs[S][KH*KW][ci_pad][co_pad] (S = split count chosen by
 * lhg_conv2d_wgrad_splits), to be summed by lhg_wgrad_reduce.  LHG_PRECISION_F32_SPLIT_F16: `x_absmax` points to Ci floats and
 * `gy_absmax` to Co floats — per-CHANNEL max|.| of the operands as passed here (lhg_channel_absmax); ignored in every other mode. */
int lhg_conv2d_wgrad_splits(int N, int H, int W, int Ci, int Co, int KH, int KW, int stride);
int lhg_conv2d_backward_weight(const float* x, int N, int H, int W, int Ci, int ldx,
                               const float* gy, int Co, int ldgy, int KH, int KW, int stride,
                               float* slabs, int S, int ci_pad, int co_pad,
                               const float* x_absmax, const float* gy_absmax, lhg_stream_t s);

/* ---- thin convolutions: 3x3 (pad 1) or 1x1, stride 1, where one side has very few channels — the RGBD / RGB inputs
 * (4 -> 64 neural_network_components.py:244-249, 3 -> 32 discriminator.py:16-19), the 64 -> 6 head (:288-291) and the
 * 1024 -> 1 critic head (discriminator.py:41).  On an MFMA tile these are > 90 % padding; they run as direct fp32 FMA
 * kernels bound by HBM instead.  `w` is the checkpoint OIHW tensor (Co, Ci, k, k) itself, no packed panel.
 * lhg_conv2d_thin_supported: 0 = not a thin convolution (use the engine above), 1 = thin input (Ci <= 4, Co = 4*2^n <= 256),
 * 2 = thin output (Co <= 8, Ci = 4*2^n <= 256 or 1024 with Co = 1; 5..8 output channels only for 1x1).
 * Forward epilogue: v = act((acc + bias[c]) * scale[c] + shift[c]) (scale / shift only with a thin input); planar_out stores a
 * thin output as (N, Co, H, W).  The three calls are closed under differentiation exactly like the engine above. */
int lhg_conv2d_thin_supported(int Ci, int Co, int k, int stride);
int lhg_conv2d_thin_forward(const float* x, int N, int H, int W, int Ci, int ldx, const float* w, int Co, int k,
                            float* y, int ldy, const float* bias, const float* scale, const float* shift,
                            int act, float slope, int planar_out, lhg_stream_t s);
/* ABI 9: the same with `y_absmax` (may be NULL; thin INPUT only): max|y| max-accumulated on the way out, as lhg_conv2d_forward's — the first conv of
 * an eval-mode generator (4 -> 64, folded BatchNorm + ReLU) feeds a GEMM directly, and lhg_absmax over its output was 0.44 ms of a 4K frame. */
int lhg_conv2d_thin_forward_amax(const float* x, int N, int H, int W, int Ci, int ldx, const float* w, int Co, int k,
                                 float* y, int ldy, const float* bias, const float* scale, const float* shift,
                                 int act, float slope, int planar_out, float* y_absmax, lhg_stream_t s);
/* ABI 10: the thin-INPUT form reading the reference's NCHW input tensor as it is (x_nchw: (N, Ci, H, W) fp32) — the RGBD frame of
 * generatePOH.py:41-70 goes into the first two convs of the generator (3x3 and the 1x1 shortcut, neural_network_components.py:22-31)
 * without the NCHW -> NHWC(32) conversion pass: 1 GB written and read twice per 4K frame. */
int lhg_conv2d_thin_forward_nchw(const float* x_nchw, int N, int H, int W, int Ci, const float* w, int Co, int k,
                                 float* y, int ldy, const float* bias, const float* scale, const float* shift,
                                 int act, float slope, float* y_absmax, lhg_stream_t s);
int lhg_conv2d_thin_backward_input(const float* gy, int N, int H, int W, int Co, int ldgy, const float* w, int Ci, int k,
                                   float* gx, int ldgx, lhg_stream_t s);
/* gw (Co, Ci, k, k) written in place (deterministic two-stage sum); ws >= lhg_conv2d_thin_wgrad_workspace bytes. */
size_t lhg_conv2d_thin_wgrad_workspace(int N, int H, int W, int Ci, int Co, int k);
int lhg_conv2d_thin_backward_weight(const float* x, int N, int H, int W, int Ci, int ldx, const float* gy, int Co, int ldgy,
                                    int k, float* gw, float* ws, size_t ws_bytes, lhg_stream_t s);

/* y = conv_transpose2d(x, W, kernel 2, stride 2).  ref: neural_network_components.py:270-286. */
int lhg_conv_transpose2x2_forward(const float* x, int N, int H, int W, int Ci, int ldx,
                                  const float* wp, int rows_pad, float* y, int Co, int ldy,
                                  const float* bias, const float* x_absmax, float* y_absmax, lhg_stream_t s);
int lhg_conv_transpose2x2_backward_input(const float* gy, int N, int H, int W, int Co, int ldgy,
                                         const float* wp, int rows_pad, float* gx, int Ci, int ldgx,
                                         const float* gy_absmax, lhg_stream_t s);
/* ABI 9: the two input-gradient calls with `gx_absmax` (may be NULL; LHG_ABSMAX_WORDS zero-filled floats): max|gx| — of what is stored, the
 * added gradient included — is max-accumulated by the GEMM epilogue, as `y_absmax` of the forward calls: gx is often the operand of the
 * next backward GEMM (the gradient of a skip-concatenation buffer feeds the transposed conv's backward), whose tensor scale then costs
 * no pass of its own. */
int lhg_conv2d_backward_input_add_amax(const float* gy, int N, int H, int W, int Co, int ldgy,
                                       const float* wp, int rows_pad, int KH, int KW, int stride,
                                       float* gx, int Ci, int ldgx, const float* res, int ldres,
                                       const float* gy_absmax, float* gx_absmax, lhg_stream_t s);
int lhg_conv_transpose2x2_backward_input_amax(const float* gy, int N, int H, int W, int Co, int ldgy,
                                              const float* wp, int rows_pad, float* gx, int Ci, int ldgx,
                                              const float* gy_absmax, float* gx_absmax, lhg_stream_t s);
int lhg_conv_transpose2x2_wgrad_splits(int N, int H, int W, int Ci, int Co);
int lhg_conv_transpose2x2_backward_weight(const float* x, int N, int H, int W, int Ci, int ldx,
                                          const float* gy, int Co, int ldgy,
                                          float* slabs, int S, int ci_pad, int co_pad,
                               const float* x_absmax, const float* gy_absmax, lhg_stream_t s);

/* grad[D0][D1][KH][KW] (+)= sum_s slabs[s][t][m][n]  (m = conv-input channel, n = conv-output
 * channel).  m_is_d1 = 1 for Conv2d (OIHW: D0 = n, D1 = m), 0 for ConvTranspose2d (IOHW).
 * accumulate != 0 adds to `grad` (a parameter's slot of the flat gradient buffer: what autograd's AccumulateGrad would do
 * with one more launch; ref: loss.backward() call sites watermelon.py:256, 275). */
int lhg_wgrad_reduce(const float* slabs, int S, int T, int m_pad, int n_pad,
                     float* grad, int D0, int D1, int m_is_d1, int accumulate, lhg_stream_t s);

/* ABI 6 — the weight gradient straight into the gradient tensor: grad (OIHW for Conv2d, IOHW for ConvTranspose2d) (+)= dW, GEMM and
 * reduction behind ONE call.  ref: the weight gradients of F.conv2d / ConvTranspose2d (neural_network_components.py:9-30, 270-286,
 * discriminator.py:16-41) as loss.backward() produces them (watermelon.py:256, 275).  In LHG_PRECISION_F32_SPLIT_F16 with fp32 storage
 * the GEMM is the tap-fused kernel (csrc/wg6_kernel.inc): the taps of a kernel row (or the whole 3x3 kernel) share one staged tile of
 * each operand, K is split just far enough to fill the chip, and the LAST workgroup of an output tile to finish sums the partial slabs
 * in split order inside the same launch (deterministic; for many splits a reduce launch follows instead); every other mode runs the
 * per-tap GEMM + lhg_wgrad_reduce.  `ws`: 256-byte aligned scratch of >= ..._workspace(...) bytes (partial slabs and the tickets of
 * the in-launch reduction, zeroed by the call); `x_absmax` / `gy_absmax`: per-channel maxima as for lhg_conv2d_backward_weight. */
size_t lhg_conv2d_backward_weight_workspace(int N, int H, int W, int Ci, int Co, int KH, int KW, int stride);
int lhg_conv2d_backward_weight_into(const float* x, int N, int H, int W, int Ci, int ldx,
                                    const float* gy, int Co, int ldgy, int KH, int KW, int stride,
                                    float* grad, int accumulate, void* ws, size_t ws_bytes,
                                    const float* x_absmax, const float* gy_absmax, lhg_stream_t s);
size_t lhg_conv_transpose2x2_backward_weight_workspace(int N, int H, int W, int Ci, int Co);
int lhg_conv_transpose2x2_backward_weight_into(const float* x, int N, int H, int W, int Ci, int ldx,
                                               const float* gy, int Co, int ldgy,
                                               float* grad, int accumulate, void* ws, size_t ws_bytes,
                                               const float* x_absmax, const float* gy_absmax, lhg_stream_t s);
/* Debugging aid of the tap-fused kernel (tools/wg6_sweep.py, tests): force its tile variant (0 .. lhg_wg6_variants() - 1), split count
 * and reduction form (1 in-launch, 0 separate launch) for the following calls; -1 = the library's deterministic plan. */
/* lhg_channel_absmax_finish over an explicit number of partial rows: several producers (the two halves of a batch normalised
 * separately: hip_ops.BatchNormPairTrainFn) have left their rows one behind the other in one buffer. */
int lhg_channel_absmax_finish_rows(const float* partial, int rows, int C, float* out, lhg_stream_t s);
int lhg_wg6_force(int variant, int splits, int fused);
int lhg_wg6_last_plan(int* variant, int* splits, int* fused); /* what the last tap-fused launch of this process ran with */
int lhg_wg6_variants(void);
const char* lhg_wg6_variant_name(int variant);

/* out[c] (+)= sum over pixels of x[pixel][c]  (bias gradients).  ws: >= 2048*C floats. */
int lhg_channel_sum(const float* x, long long pixels, int C, int ld, float* out, int accumulate, float* ws, lhg_stream_t s);

/* ------------------------------------------------------------------ batch norm (train / eval)
 * ref: nn.LazyBatchNorm2d neural_network_components.py:23-24, nn.BatchNorm2d discriminator.py:39.
 * eps 1e-5.  stats[0..C) = batch mean, stats[C..2C) = 1/sqrt(biased var + eps).
 * running_* (may be NULL) are updated with `momentum` using the unbiased variance.
 * ws: >= 4104*C floats (2048 partial blocks x 2 sums + the reduced sums; reduced in double). */
int lhg_bn_stats(const float* x, long long pixels, int C, int ld, float* stats,
                 float* running_mean, float* running_var, float momentum, float eps,
                 float* ws, lhg_stream_t s);
/* ABI 10: lhg_bn_stats from the partial rows of lhg_conv2d_forward_stats (`rows` x 2 x C floats, sums shifted by `shift[c]` — the
 * conv's bias, NULL: zero) instead of a pass over the tensor: stats / running_* exactly as lhg_bn_stats defines them. */
int lhg_bn_stats_finish(const float* partial, int rows, const float* shift, long long pixels, int C, float* stats,
                        float* running_mean, float* running_var, float momentum, float eps, lhg_stream_t s);
/* y = act((x - mean)*invstd*gamma + beta + res).  y_absmax (may be NULL, else LHG_ABSMAX_WORDS zero-filled floats): max-accumulates max|y| on the way out — the
 * lhg_absmax of the output for free, for a following GEMM in the LHG_PRECISION_F32_SPLIT_F16 mode (zero-filled slot, fp32 storage). */
int lhg_bn_apply(const float* x, int ldx, long long pixels, int C, const float* stats,
                 const float* gamma, const float* beta, const float* res, int ldres,
                 int act, float slope, float* y, int ldy, float* y_absmax, lhg_stream_t s);
/* lhg_bn_apply (fp32 storage) that also writes the per-channel partial maxima of y (see lhg_channel_absmax_finish). */
int lhg_bn_apply_chanmax(const float* x, int ldx, long long pixels, int C, const float* stats,
                         const float* gamma, const float* beta, const float* res, int ldres,
                         int act, float slope, float* y, int ldy, float* y_absmax, float* y_chanmax_partial, lhg_stream_t s);
/* Backward of y = act(bn(x) + res) given gy:  g = gy * act'(y);  gres = g (if non-NULL);
 * gx = gamma*invstd*(g - mean(g) - xhat*mean(g*xhat)); ggamma (+)= sum g*xhat; gbeta (+)= sum g  (accumulate != 0 adds).
 * `y` is the forward OUTPUT (sign gives the activation mask).  ws: >= 8192*C floats.
 * gx_absmax (may be NULL): max-accumulates max|gx|, as y_absmax above (gx is the gy operand of the preceding conv's backward GEMMs).
 * y may be NULL when the forward had no residual and a ReLU / LeakyReLU: the kernels then recompute the pre-activation from x, gamma
 * and `beta` exactly as lhg_bn_apply rounded it (one fma per element) and take the mask from its sign — one tensor read less in each
 * of the two passes; `beta` is only read in that case. */
int lhg_bn_backward(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy,
                    long long pixels, int C, const float* stats, const float* gamma,
                    int act, float slope, float* gx, int ldgx, float* gres, int ldgres,
                    float* ggamma, float* gbeta, int accumulate, float* ws, float* gx_absmax,
                    const float* beta, lhg_stream_t s);
/* lhg_bn_backward (fp32 storage) that also writes the per-channel partial maxima of gx and, when both `gres` and
 * `gres_chanmax_partial` are non-NULL, of gres (the gy operand of the block's shortcut conv) — see lhg_channel_absmax_finish.
 * `gres_absmax` (ABI 7; NULL: not measured): max|gres| is max-accumulated into it like max|gx| into `gx_absmax` — the tensor scale
 * the shortcut conv's input-gradient GEMM needs, without a pass of its own over gres. */
int lhg_bn_backward_chanmax(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy,
                            long long pixels, int C, const float* stats, const float* gamma,
                            int act, float slope, float* gx, int ldgx, float* gres, int ldgres,
                            float* ggamma, float* gbeta, int accumulate, float* ws, float* gx_absmax,
                            const float* beta, float* gx_chanmax_partial, float* gres_chanmax_partial, float* gres_absmax,
                            lhg_stream_t s);
/* Double backward of the gx output above (WGAN-GP, ref: watermelon.py:466-473):
 * given ggx (cotangent of gx) returns ggy (cotangent of gy), gx2 (cotangent of x) and
 * ggamma2 (cotangent of gamma).  ws: >= 5*4096*C floats. */
int lhg_bn_backward_backward(const float* ggx, const float* gy, const float* x, const float* y,
                             long long pixels, int C, const float* stats, const float* gamma,
                             int act, float slope, float* ggy, float* gx2, float* ggamma2,
                             float* ws, lhg_stream_t s);

/* Synchronised batch statistics for data-parallel replicas (the reference is single-device: a global batch of 2B on two replicas of B has
 * to normalise over all 2B samples to reproduce it; SURVEY §5 "Distributed comm backend").  The two calls above split in halves with the
 * per-channel sums in the caller's hands: `..._sums` writes the LOCAL sums (2*C floats: sum g, sum g*xhat; double backward 5*C floats:
 * S_g, S_gx, S_q, S_qx, S_gq with xc = x - mean), the caller all-reduces them (and takes ggamma / gbeta from the LOCAL ones), `..._apply`
 * finishes with inv_count = 1 / (samples behind the sums).  `stats` are the GLOBAL mean / invstd.  ws as for the unsplit calls. */
int lhg_bn_backward_sums(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy, long long pixels, int C,
                         const float* stats, const float* gamma, int act, float slope, float* sums, float* ws, const float* beta,
                         lhg_stream_t s);
int lhg_bn_backward_apply(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy, long long pixels, int C,
                          const float* stats, const float* gamma, const float* sums, float inv_count, int act, float slope,
                          float* gx, int ldgx, float* gres, int ldgres, float* gx_absmax, const float* beta, lhg_stream_t s);
int lhg_bn_backward_backward_sums(const float* ggx, const float* gy, const float* x, const float* y, long long pixels, int C,
                                  const float* stats, const float* gamma, int act, float slope, float* sums, float* ws, lhg_stream_t s);
int lhg_bn_backward_backward_apply(const float* ggx, const float* gy, const float* x, const float* y, long long pixels, int C,
                                   const float* stats, const float* gamma, const float* sums, float inv_count, int act, float slope,
                                   float* ggy, float* gx2, float* ggamma2, lhg_stream_t s);

/* ------------------------------------------------------------------ fused calls (ABI 9)
 * Every reduction above is two launches: per-workgroup partial rows, then a small kernel that folds the rows.  The calls below finish
 * the rows INSIDE the launch that writes them (the last workgroup of a group of 32 rows folds its group, the last group folds the
 * groups; fixed summation order, double accumulation: bit-repeatable whatever the arrival order) and put a whole BatchNorm forward /
 * backward behind one entry point: 2 launches where lhg_bn_stats + lhg_bn_apply_chanmax + lhg_channel_absmax_finish were 4, 2 where
 * lhg_bn_backward_chanmax + two finishes were 5.  A train step at 384 x 384, batch 4 makes ~200 launches fewer.
 * Scratch: `ws` = LHG_FUSED_WS_FLOATS floats (16-byte aligned, contents irrelevant) and `tickets` = LHG_FUSED_TICKETS unsigned that are
 * ZERO before the first call and that every call leaves at zero again.  The caller keeps ONE pair per stream: two calls may share a pair
 * only if their launches cannot overlap (same stream).  C <= 4096.
 * ref: F.batch_norm(training=True) via nn.LazyBatchNorm2d neural_network_components.py:23-32, nn.BatchNorm2d discriminator.py:34-40,
 * the double backward watermelon.py:466-473. */
#define LHG_FUSED_WS_FLOATS 1114112
#define LHG_FUSED_TICKETS 4160
long long lhg_fused_workspace_floats(void);
int lhg_fused_ticket_count(void);
/* lhg_bn_stats + lhg_bn_apply(_chanmax) in one call: batch statistics of x -> `stats` (2*C floats: mean, invstd) and the running
 * statistics (may be NULL), then y = act((x - mean)*invstd*gamma + beta + res).  y_absmax as for lhg_bn_apply.  y_chanmax (may be
 * NULL; fp32 storage) with chanmax_finish != 0: C floats that receive the FINISHED per-channel max|y| (the apply launch folds its
 * own partial rows: measured ~11 us longer per launch, on the stream the step waits for); with chanmax_finish == 0 it is the
 * partial-row buffer of lhg_bn_apply_chanmax (lhg_chanmax_partial_rows x C floats), finished by lhg_channel_absmax_finish on
 * whatever stream asks — the op layer's choice: that launch runs on the weight-gradient stream, beside the main chain. */
int lhg_bn_forward_train(const float* x, int ldx, long long pixels, int C, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float momentum, float eps,
                         const float* res, int ldres, int act, float slope, float* y, int ldy,
                         float* stats, float* y_absmax, float* y_chanmax, int chanmax_finish,
                         float* ws, unsigned* tickets, lhg_stream_t s);
/* lhg_bn_backward(_chanmax) in one call and two launches; gx_chanmax / gres_chanmax (may be NULL; fp32 storage): finished maxima
 * (C floats each) or partial-row buffers, by chanmax_finish as above.  Everything else as lhg_bn_backward_chanmax. */
int lhg_bn_backward_fused(const float* gy, int ldgy, const float* x, int ldx, const float* y, int ldy, long long pixels, int C,
                          const float* stats, const float* gamma, const float* beta, int act, float slope,
                          float* gx, int ldgx, float* gres, int ldgres, float* ggamma, float* gbeta, int accumulate,
                          float* gx_absmax, float* gres_absmax, float* gx_chanmax, float* gres_chanmax, int chanmax_finish,
                          float* ws, unsigned* tickets, lhg_stream_t s);
/* lhg_bn_backward_backward in two launches. */
int lhg_bn_backward_backward_fused(const float* ggx, const float* gy, const float* x, const float* y, long long pixels, int C,
                                   const float* stats, const float* gamma, int act, float slope,
                                   float* ggy, float* gx2, float* ggamma2, float* ws, unsigned* tickets, lhg_stream_t s);
/* lhg_channel_sum / lhg_channel_absmax in one launch each. */
int lhg_channel_sum_fused(const float* x, long long pixels, int C, int ld, float* out, int accumulate, float* ws, unsigned* tickets, lhg_stream_t s);
int lhg_channel_absmax_fused(const float* x, long long pixels, int C, int ld, float* out, float* ws, unsigned* tickets, lhg_stream_t s);

/* ------------------------------------------------------------------ pointwise / pooling */
/* 2x2 stride-2 max pool, NHWC.  ref: neural_network_components.py:252-268. */
int lhg_maxpool2x2_forward(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, lhg_stream_t s);
/* gx = gy routed to the first maximal element of each window (PyTorch tie rule). */
int lhg_maxpool2x2_backward(const float* x, int ldx, const float* gy, int ldgy, int N, int H, int W, int C,
                            float* gx, int ldgx, lhg_stream_t s);
/* ABI 8: the same with `gadd` (N, H, W, C; row pitch ldgadd) added to gx on the way out — the gradient that reached x through its other
 * consumer (the skip connection of the UNet, ref: neural_network_components.py:310-313): autograd would add the two full-size gradients
 * in a pass of its own over a strided slice. */
int lhg_maxpool2x2_backward_add(const float* x, int ldx, const float* gy, int ldgy, int N, int H, int W, int C,
                                const float* gadd, int ldgadd, float* gx, int ldgx, lhg_stream_t s);
/* g_out = g * act'(y) with the mask taken from the forward output y. */
int lhg_act_backward(const float* g, int ldg, const float* y, int ldy, long long pixels, int C,
                     int act, float slope, float* out, int ldo, lhg_stream_t s);

/* ------------------------------------------------------------------ angular spectrum (A5 A8 A9)
 * One fused operator  out = crop( IFFT2( F1 (.) F2 (.) FFT2( pad( in ) ) ) )  on `planes`
 * independent (rows0 x cols0) fields, done as three LDS-resident batched 1-D FFT passes
 * that never materialise the zero padding.  ref: angular_spectrum_method.py:374-392,
 * 503-552 (torch.fft.fft2 / ifft2 / F.pad / slicing call sites).
 *   in_mode : 0 polar (a, phi*phase_scale)   1 phase only (e^{i phi})   2 complex
 *   out_mode: 0 complex   1 |z| and angle(z) (+ optional complex copy)   2 |z| only
 *   filter op per factor: 0 none, 1 multiply, 2 multiply by conj, 3 divide, 4 divide by conj
 *   f_index[p] selects the (rows x cols) slab of the factor used by plane p.
 * rows, cols (padded extents) must be of the form 2^a * 3^b in [16, 4096], at most 3072 when divisible by 3 (Stockham radix 4 / 2 / 3 in LDS).
 * ws: 2 * planes * rows0 * cols complex64.                                             */
typedef struct lhg_asm_filter {
  const float* f1; const int32_t* f1_index; int f1_op;  /* device pointers */
  const float* f2; const int32_t* f2_index; int f2_op;
} lhg_asm_filter;

int lhg_asm_propagate(const float* in_a, const float* in_b, int in_mode, float phase_scale,
                      int planes, int rows0, int cols0, int pad_r, int pad_c,
                      const lhg_asm_filter* filt,
                      float* out_a, float* out_b, float* out_complex, int out_mode,
                      float* ws, size_t ws_bytes, const float* twiddle_rows, const float* twiddle_cols,
                      lhg_stream_t s);
/* ABI 9: the same with `in_planes` input fields and `planes` outputs — output q = crop(ifft2(filter_q . fft2(pad(input plane_src[q])))), plane_src
 * a device int32[planes] (NULL: one to one).  The first pass (polar -> complex, row transforms) runs once per INPUT field: the reference's
 * multi-distance __call__ (angular_spectrum_method.py:503-522) applies D transfer functions to every field.  ws >= (in_planes + planes) * rows0 * C * 8 bytes. */
int lhg_asm_propagate_shared(const float* in_a, const float* in_b, int in_mode, float phase_scale,
                             int in_planes, const int* plane_src, int planes, int rows0, int cols0, int pad_r, int pad_c,
                             const lhg_asm_filter* filt, float* out_a, float* out_b, float* out_complex, int out_mode,
                             float* ws, size_t ws_bytes, const float* twiddle_rows, const float* twiddle_cols, lhg_stream_t s);
/* spectrum = F1 (.) F2 (.) FFT2(pad(in)) written in full (planes, rows, cols) complex64
 * (API parity with propagate_POH2Freq_forward / filter_AP2filteredFreq). */
int lhg_asm_to_spectrum(const float* in_a, const float* in_b, int in_mode, float phase_scale,
                        int planes, int rows0, int cols0, int pad_r, int pad_c,
                        const lhg_asm_filter* filt, float* spectrum,
                        float* ws, size_t ws_bytes, const float* twiddle_rows, const float* twiddle_cols,
                        lhg_stream_t s);
/* out = crop(IFFT2(F1 (.) F2 (.) spectrum)) from a full spectrum. */
int lhg_asm_from_spectrum(const float* spectrum, int planes, int rows0, int cols0, int pad_r, int pad_c,
                          const lhg_asm_filter* filt,
                          float* out_a, float* out_b, float* out_complex, int out_mode,
                          float* ws, size_t ws_bytes, const float* twiddle_rows, const float* twiddle_cols,
                          lhg_stream_t s);
/* ABI 9: several filters of one spectrum — output q = crop(ifft2(filter_q . spectrum[plane_src[q]])), plane_src a device int32[planes] (NULL: one to one). */
int lhg_asm_from_spectrum_shared(const float* spectrum, const int* plane_src, int planes, int rows0, int cols0, int pad_r, int pad_c,
                                 const lhg_asm_filter* filt, float* out_a, float* out_b, float* out_complex, int out_mode,
                                 float* ws, size_t ws_bytes, const float* twiddle_rows, const float* twiddle_cols, lhg_stream_t s);
/* The `twiddle_rows` / `twiddle_cols` table of a transform length n: lhg_fft_table_floats(n) floats, filled by lhg_fft_twiddles.
 * n a product of 2, 3, 5, 7, 11, 13 in [16, 4096] (and <= 256 x the per-thread budget of its largest radix: 3072 with a factor 3,
 * 3328 with 13, ...): twiddle[k] = exp(-2 pi i k / n), k < n (2n floats), computed in double on the device; the transform runs on
 * radix-4 / 2 / 3 / 5 / 7 / 11 / 13 Stockham stages.  Any other n in [16, 8192] runs as a Bluestein (chirp-z) convolution of length m >= 2n - 1 (a power of two up to 4096, the shortest 2^a 3^b length above) inside the same
 * kernels; its table is [m twiddles of length m][n chirp values exp(i pi j^2 / n)][FFT_m(wrapped chirp) / m].
 * ref: torch.fft.fft2 / ifft2 accept any extent, angular_spectrum_method.py:382-383 (e.g. 192 + 2*320 = 832 = 2^6 * 13). */
long long lhg_fft_table_floats(int n);
int lhg_fft_twiddles(float* twiddle, int n, lhg_stream_t s);

/* ------------------------------------------------------------------ AP2POH tail (A6 A7), forward and backward
 * field (planes,rows,cols) complex64 -> POH.  ref: AP2POH.py:105-116, utilities.py:53-66,
 * neural_network_components.py:68-75.  taps[plane%3][3] = (centre, edge, corner), bias[plane%3].
 *   mod = stencil(field) + bias;  a = |mod| / (1.01 max_plane |mod|);  POH = angle(mod) +/- acos(a) on a checkerboard.
 * plane_peak[planes] (uint64, ZERO it first) receives (float bits of the peak) << 32 | (0xFFFFFFFF - argmax index). */
int lhg_symconv_field(const float* field, int planes, int rows, int cols, const float* taps, const float* bias,
                      float* mod, unsigned long long* plane_peak, lhg_stream_t s);
int lhg_double_phase_encode(const float* mod, const unsigned long long* plane_peak, int planes, int rows, int cols,
                            float* poh, lhg_stream_t s);
/* g_mod (complex) from g_poh, including the gradient through the per-plane max (routed to the arg-max pixel).
 * ws: planes * lhg_poh_partial_blocks(rows, cols) floats. */
int lhg_poh_partial_blocks(int rows, int cols);
int lhg_double_phase_encode_backward(const float* g_poh, const float* mod, const unsigned long long* plane_peak,
                                     int planes, int rows, int cols, float* g_mod, float* ws, lhg_stream_t s);
/* g_field = stencil(g_mod); partial[planes][blocks][4] = per-block sums for d(centre, edge, corner, bias) of the
 * plane's colour (sum them over planes of one colour and blocks). */
int lhg_symconv_field_backward(const float* g_mod, const float* field, int planes, int rows, int cols, const float* taps,
                               float* g_field, float* partial, lhg_stream_t s);

/* ABI 9: the pointwise Jacobians at the two ends of the (linear) angular-spectrum operator, one launch each (they were ~25 torch
 * elementwise launches per operator call).  ref: the autograd of abs() / angle() / amp * exp(1j * phase) at angular_spectrum_method.py:374-392, 533-552.
 *   output side: gz (n complex64) = scale * ( g_abs * z / |z| + g_angle * (-Im z, Re z) / |z|^2 ), zero where z == 0; either cotangent may be NULL;
 *   input side:  u = a * exp(i s phi) with cotangent g = pre_scale * g_in: g_a = Re(conj(e) g), g_phi = Im(conj(e) g) * a * s, e = exp(i s phi);
 *                phi == NULL (phase-only input u = exp(i s a)): g_a = Im(conj(e) g) * s. */
int lhg_polar_output_cotangent(const float* z, const float* g_abs, const float* g_angle, float scale, float* gz, long long n, lhg_stream_t s);
int lhg_polar_input_cotangent(const float* g_in, const float* a, const float* phi, float phase_scale, float pre_scale,
                              float* g_a, float* g_phi, long long n, lhg_stream_t s);

/* ------------------------------------------------------------------ reconstruction losses (A13), fused
 * (focal sin/cos phase-gradient, pixel MSE, |TV(hat) - TV(target)|) of (planes,H,W) fp32 amplitudes / phases.
 * ref: loss_func.py:66-98, 135-163; watermelon.py:418-445.  sums9: the nine global reductions (kept for backward),
 * losses3 = (focal, pixel, tv).  ws: lhg_recon_loss_blocks(...) * 9 floats.  Backward takes d(loss)/d(losses3). */
int lhg_recon_loss_blocks(int planes, int H, int W);
int lhg_recon_loss_forward(const float* hat_amp, const float* tgt_amp, const float* hat_phs, const float* tgt_phs,
                           int planes, int H, int W, float* sums9, float* losses3, float* ws, lhg_stream_t s);
int lhg_recon_loss_backward(const float* hat_amp, const float* tgt_amp, const float* hat_phs, const float* tgt_phs,
                            int planes, int H, int W, const float* sums9, const float* upstream3,
                            float* g_hat_amp, float* g_hat_phs, lhg_stream_t s);

/* PSNR and SSIM of `hat` against `tgt` (planar (planes, H, W), e.g. planes = B*3) exactly as the reference records them every batch
 * with torchmetrics' defaults (watermelon.py:134-135, 447-456): out2[0] = 10 log10((max tgt - min tgt)^2 / mse), out2[1] = mean SSIM
 * (11x11 Gaussian window, sigma 1.5, data range max(range hat, range tgt), k1 0.01, k2 0.03, border of 5 pixels cropped).
 * Device-side reductions only (no host synchronisation); ws >= lhg_psnr_ssim_workspace bytes. */
size_t lhg_psnr_ssim_workspace(int planes, int H, int W);
int lhg_psnr_ssim(const float* hat, const float* tgt, int planes, int H, int W, float* out2, void* ws, size_t ws_bytes, lhg_stream_t s);

/* ------------------------------------------------------------------ optimiser
 * torch.optim.Adam (no weight decay, no amsgrad) on one flat tensor.  ref: watermelon.py:137-138. */
int lhg_adam_step(float* p, const float* g, float* m, float* v, long long n,
                  float lr, float beta1, float beta2, float eps, int step, lhg_stream_t s);
/* ABI 6: the same step on g * grad_scale (data-parallel runs hand over the all-reduced SUM and 1 / world: the division costs no pass of
 * its own; 1.0 is exact) and, with consts4 != NULL, with {1 - beta1^t, sqrt(1 - beta2^t), grad_scale, lr} read from DEVICE memory at
 * execution time instead of the by-value arguments (`step`, `grad_scale`, `lr` are then ignored): a launch captured into a hipGraph
 * follows the step count and a learning-rate schedule through that buffer. */
int lhg_adam_step_scaled(float* p, const float* g, float* m, float* v, long long n,
                         float lr, float beta1, float beta2, float eps, int step,
                         float grad_scale, const float* consts4, lhg_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* LHG_HIP_H */
