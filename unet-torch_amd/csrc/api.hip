// C-ABI dispatch layer of libunetmi: validates arguments and picks the MFMA fast path or the
// generic kernel.  See include/unetmi.h for the contract of every entry point.
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

// generic_kernels.hip
int umi_conv_fwd_generic(const void* x, int ldx, const void* tx, const void* wp, const float* bias, void* y, int ldy,
                         float* stat_part, int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad,
                         int Ho, int Wo, int off_h, int off_w, int out_H, int out_W, int in_dtype, int out_dtype,
                         int flags, hipStream_t s);
size_t umi_conv_wgrad_generic_ws_bytes(int N, int Ho, int Wo, int Ci, int Co, int R, int S);
int umi_conv_wgrad_generic(const void* x, int ldx, const void* txa, const void* dy, int lddy, const void* txb, float* dW,
                           long s_co, long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci, int Co, int R,
                           int S, int stride, int pad, int Ho, int Wo, int dtype, void* ws, size_t ws_bytes,
                           hipStream_t st);
// conv_mfma.hip
bool umi_conv3x3_mfma_ok(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo,
                         int ldx, int ldy, int in_dtype, int out_dtype, int flags, const float* bias);
int umi_conv3x3_mfma_stat_rows(int N, int Ho, int Wo, int Ci, int Co, int ldx);
int umi_conv3x3_mfma(const void* x, int ldx, const void* tx, const void* wp8, void* y, int ldy, float* stat_part,
                     int N, int H, int W, int Ci, int Co, hipStream_t s);
// conv1x1_mfma.hip
bool umi_conv1x1_mfma_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldx, int ldy, int in_dtype,
                         int out_dtype, int flags);
struct UmiLinearEpi { int mode; float p; unsigned seed; const unsigned* seed_dev; void* mask; const void* aux; int ldaux; void* y2; int ldy2;
                      const void* bn_tx; const float* bn_rstd; float* bn_part; };
int umi_conv1x1_bnred_rows(long M, int Ntot);
int umi_conv1x1_mfma(const void* x, int ldx, const void* tx, const void* wp8, const float* bias, void* y, int ldy,
                     int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo, int off_h,
                     int off_w, int out_H, int out_W, int flags, hipStream_t s, const UmiLinearEpi* epi = nullptr);
// stem_head.hip
bool umi_stem_fwd_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldy, int in_dtype, int out_dtype, int flags,
                     const float* bias);
int umi_stem_stat_rows(int N, int H, int W);
int umi_stem_fwd(const void* x, int ldx, const void* tx, const void* wp, void* y, int ldy, float* part, int N, int H, int W,
                 int Ci, int Co, hipStream_t s);
bool umi_stem_wgrad_ok(int Ci, int Co, int R, int S, int stride, int pad, int lddy, int dtype, int flags, const void* txb);
size_t umi_stem_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co);
int umi_stem_wgrad_bnapply(const void* x, int ldx, const void* txa, const void* da, int ldda, const void* y, int ldy,
                           const void* tx_bn, const float* rstd, const float* sum_dz, const float* sum_dzx, float* dW, long s_co,
                           long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci, int Co, void* ws, size_t ws_bytes,
                           hipStream_t s);
int umi_stem_wgrad(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co, long s_ci,
                   long s_t, float out_scale, int N, int H, int W, int Ci, int Co, void* ws, size_t ws_bytes, hipStream_t s);
bool umi_head_fwd_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldx, int in_dtype, int out_dtype, int flags);
int umi_head_fwd(const void* x, int ldx, const void* tx, const void* wp, const float* bias, void* y, int ldy, float* part,
                 long P, int Ci, int Co, int out_dtype, hipStream_t s);
int umi_head_stat_rows(long P, int Ci);
bool umi_smallk_fwd_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldy, int in_dtype, int out_dtype, int flags,
                       const void* tx, const float* bias);
int umi_smallk_fwd(const void* x, int ldx, const void* wp, void* y, int ldy, long P, int Ci, int Co, hipStream_t s);
bool umi_head_wgrad_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldx, int dtype, int flags, const void* txb);
size_t umi_head_wgrad_ws_bytes(long P, int Ci, int Co);
int umi_head_wgrad(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co, long s_ci,
                   long s_t, float out_scale, long P, int Ci, int Co, void* ws, size_t ws_bytes, hipStream_t s);
bool umi_wgrad1x1_mfma_ok(long M, int Ci, int Co, int R, int S, int stride, int pad, int ldx, int lddy, int dtype, int flags,
                          const void* txb);
size_t umi_wgrad1x1_mfma_ws_bytes(long M, int Ci, int Co);
int umi_wgrad1x1_mfma(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co, long s_ci,
                      long s_t, float out_scale, long M, int Ci, int Co, void* ws, size_t ws_bytes, hipStream_t s);
bool umi_wgradT_mfma_ok(int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo, int ldx,
                        int lddy, int dtype, int flags, const void* txa);
size_t umi_wgradT_mfma_ws_bytes(int N, int Ho, int Wo, int Ci, int Co);
int umi_wgradT_mfma(const void* x, int ldx, const void* dy, int lddy, const void* txb, float* dW, long s_co, long s_ci,
                    long s_t, float out_scale, int N, int Ho, int Wo, int Ci, int Co, void* ws, size_t ws_bytes,
                    hipStream_t s);
// narrow_convs.hip
bool umi_root_fwd_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldy, int in_dtype, int out_dtype, int flags,
                     const void* tx, const float* bias);
int umi_root_fwd(const void* x, int ldx, const void* wp, void* y, int ldy, int N, int H, int W, int Ho, int Wo, int Co,
                 hipStream_t s);
bool umi_root_wgrad_ok(int Ci, int Co, int R, int S, int stride, int pad, int lddy, int dtype, int flags, const void* txa,
                       const void* txb);
size_t umi_root_wgrad_ws_bytes(int N, int Ho, int Wo, int Co);
int umi_root_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dW, long s_co, long s_ci, long s_t, float out_scale,
                   int N, int H, int W, int Ho, int Wo, int Co, void* ws, size_t ws_bytes, hipStream_t s);
bool umi_head3_fwd_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldx, int in_dtype, int out_dtype, int flags);
int umi_head3_fwd(const void* x, int ldx, const void* tx, const void* wp, const float* bias, void* y, int ldy, int N, int H,
                  int W, int Ci, int Co, hipStream_t s);
bool umi_head3_wgrad_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldx, int dtype, int flags, const void* txb);
size_t umi_head3_wgrad_ws_bytes(long P, int Ci, int Co);
int umi_head3_wgrad(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co, long s_ci, long s_t,
                    float out_scale, int N, int H, int W, int Ci, int Co, void* ws, size_t ws_bytes, hipStream_t s);
bool umi_wgrad_gather_mfma_ok(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo, int ldx,
                              int lddy, int dtype, int flags, const void* txb);
size_t umi_wgrad_gather_mfma_ws_bytes(int N, int Ho, int Wo, int Ci, int Co, int R, int S);
int umi_wgrad_gather_mfma(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co, long s_ci,
                          long s_t, float out_scale, int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad,
                          int Ho, int Wo, void* ws, size_t ws_bytes, hipStream_t s);
bool umi_wgrad3x3_mfma_ok(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo,
                          int ldx, int lddy, int dtype, int flags, const void* txb);
size_t umi_wgrad3x3_mfma_ws_bytes(int N, int H, int W, int Ci, int Co);
int umi_wgrad1x1_mfma_group(int n, const void* const* x, int ldx, const void* const* dy, int lddy, float* const* dW, long s_co,
                            long s_ci, float out_scale, long M, int Ci, int Co, hipStream_t s);
int umi_wgrad3x3_mfma_bnapply(const void* x, int ldx, const void* txa, const void* da, int ldda, const void* ybn, int ldybn,
                              const void* txbn, const float* rstd, const float* sum_dz, const float* sum_dzx, void* dz,
                              int lddz, float* dW, long s_co, long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci,
                              int Co, void* ws, size_t ws_bytes, hipStream_t s);
int umi_wgrad3x3_mfma(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co,
                      long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci, int Co, void* ws,
                      size_t ws_bytes, hipStream_t s);

extern "C" int umi_version(void) { return 1; }
extern "C" const char* umi_arch(void) { return "gfx950"; }

// A ViT-block linear with its elementwise tail in the GEMM epilogue (reference vit_seg_modeling.py:113-119 Mlp, :177-187 Block):
//   epi 1: y = x W + b,  y2 = dropout(GELU(y)),  mask          (fc1; y stays for the GELU backward)
//   epi 2: y = dropout(x W + b) + aux,           mask          (fc2 / attention output projection + residual)
// x [M, Ci], y / y2 / aux [M, Co] fp16 rows (ld in elements), wp8 = umi_pack_kn8 of the [Co, Ci] weight, mask = M * Co bytes.
// Same values as umi_conv_fwd followed by umi_dropout_fused (same random stream and roundings).  UMI_ERR_UNSUPPORTED where
// the pointwise matrix-core kernel does not apply: run the two calls instead.
extern "C" int umi_linear_fused(const void* x, int ldx, const void* wp8, const float* bias, void* y, int ldy, long M, int Ci,
                                int Co, int epi, float p, unsigned seed, const unsigned* seed_dev, void* mask, const void* aux,
                                int ldaux, void* y2, int ldy2, int dtype, umi_stream_t stream) {
    if (!x || !wp8 || !y || !mask || M <= 0 || M >= (1L << 31) || Ci <= 0 || Co <= 0 || p < 0.f || p >= 1.f) return UMI_ERR_BADARG;
    if (epi != 1 && epi != 2) return UMI_ERR_BADARG;
    if ((epi == 1 && (!y2 || ldy2 % 8)) || (epi == 2 && (!aux || ldaux % 8))) return UMI_ERR_BADARG;
    if (dtype != UMI_F16 || !umi_conv1x1_mfma_ok(Ci, Co, 1, 1, 1, 0, ldx, ldy, UMI_F16, UMI_F16, 0)) return UMI_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)wp8 | (uintptr_t)y2 | (uintptr_t)aux) & 15 || ((uintptr_t)mask & 7)) return UMI_ERR_UNSUPPORTED;
    const UmiLinearEpi e{epi, p, seed, seed_dev, mask, aux, ldaux, y2, ldy2, nullptr, nullptr, nullptr};
    return umi_conv1x1_mfma(x, ldx, nullptr, wp8, bias, y, ldy, 1, 1, (int)M, Ci, Co, 1, 1, 1, 0, 1, (int)M, 0, 0, 1, (int)M, 0,
                            (hipStream_t)stream, &e);
}

// A data gradient on the pointwise / tap-gather matrix-core kernel (ConvTranspose2d(2,2)'s = a stride-2 2x2 conv over d(up),
// reference Model.py:56-57 under autograd; plain 1x1 convs; UMI_CONV_DGRAD_STRIDED) that also emits stage 1 of the BatchNorm(+ReLU)
// backward of the layer whose activated output the gradient belongs to: part[rows][2][Co], rows = umi_conv_gather_bnred_rows(...)
// (0 = not on that kernel: run umi_conv_fwd and umi_bn_bwd_reduce).  No transform, no bias, no accumulation.
extern "C" int umi_conv_gather_bnred_rows(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo,
                                          int ldx, int ldy, int dtype, int flags) {
    if (flags & (UMI_CONV_UPSAMPLE2 | UMI_CONV_ACCUMULATE | UMI_CONV_FORCE_GENERIC)) return 0;
    if (umi_conv3x3_mfma_ok(N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, ldx, ldy, dtype, dtype, flags, nullptr)) return 0;
    if (!umi_conv1x1_mfma_ok(Ci, Co, R, S, stride, pad, ldx, ldy, dtype, dtype, flags)) return 0;
    return umi_conv1x1_bnred_rows((long)N * Ho * Wo, Co);
}
extern "C" int umi_conv_gather_bnred(const void* x, int ldx, const void* wp8, void* y, int ldy, const void* ybn, int ldybn,
                                     const void* txbn, const float* rstd, float* part, int N, int H, int W, int Ci, int Co, int R,
                                     int S, int stride, int pad, int Ho, int Wo, int dtype, int flags, umi_stream_t stream) {
    if (!x || !wp8 || !y || !ybn || !txbn || !rstd || !part || N <= 0 || H <= 0 || W <= 0) return UMI_ERR_BADARG;
    if (umi_conv_gather_bnred_rows(N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, ldx, ldy, dtype, flags) <= 0 || ldybn % 8 || ldybn < Co)
        return UMI_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)wp8 | (uintptr_t)ybn) & 15) return UMI_ERR_BADARG;
    const UmiLinearEpi e{3, 0.f, 0u, nullptr, nullptr, ybn, ldybn, nullptr, 0, txbn, rstd, part};
    return umi_conv1x1_mfma(x, ldx, nullptr, wp8, nullptr, y, ldy, N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, 0, 0, Ho, Wo, flags,
                            (hipStream_t)stream, &e);
}

extern "C" int umi_conv_fwd_plan(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int ldx,
                                 int ldy, int in_dtype, int out_dtype, int flags, int has_bias, int* layout,
                                 int* stat_rows) {
    if (N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || R <= 0 || S <= 0 || stride <= 0) return UMI_ERR_BADARG;
    const bool ups = flags & UMI_CONV_UPSAMPLE2;
    const int Ho = ups ? H : (H + 2 * pad - R) / stride + 1, Wo = ups ? W : (W + 2 * pad - S) / stride + 1;
    static const float one = 1.f;
    const bool dgs = flags & UMI_CONV_DGRAD_STRIDED;      // only the tap-gather MFMA kernel or the generic one take these
    const bool mfma = !dgs && umi_conv3x3_mfma_ok(N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, ldx, ldy, in_dtype, out_dtype,
                                                  flags, has_bias ? &one : nullptr);
    const bool mfma1 = !mfma && umi_conv1x1_mfma_ok(Ci, Co, R, S, stride, pad, ldx, ldy, in_dtype, out_dtype, flags);
    if ((flags & UMI_CONV_ACCUMULATE) && !mfma1) return UMI_ERR_UNSUPPORTED;
    if (layout) *layout = (mfma || mfma1) ? 1 : 0;
    const bool stem = !mfma && !mfma1 && !dgs &&
                      umi_stem_fwd_ok(Ci, Co, R, S, stride, pad, ldy, in_dtype, out_dtype, flags, has_bias ? &one : nullptr);
    const bool head = !mfma && !mfma1 && !dgs && !stem && out_dtype == UMI_F16 &&
                      umi_head_fwd_ok(Ci, Co, R, S, stride, pad, ldx, in_dtype, out_dtype, flags);
    if (stat_rows)
        *stat_rows = mfma ? umi_conv3x3_mfma_stat_rows(N, H, W, Ci, Co, ldx)
                          : (stem ? umi_stem_stat_rows(N, H, W)
                                  : (head ? umi_head_stat_rows((long)N * H * W, Ci) : umi_cdiv((long)N * Ho * Wo, 64)));
    return UMI_OK;
}

int umi_conv3x3_mfma_bnred(const void* dy, int lddy, const void* wp8, void* da, int ldda, const void* ybn, int ldybn,
                           const void* txbn, const float* rstd, float* part, int N, int H, int W, int Ci, int Co,
                           hipStream_t s);
void umi_launch_reduce_rows2(const float* ws, int rows, int C, float* out0, float* out1, float scale, hipStream_t s);

// 3x3 / stride 1 / pad 1 data gradient (x = dy, Ci = the forward conv's Co, weights rotated + transposed as for umi_conv_fwd)
// fused with stage 1 of the BatchNorm+ReLU backward of the layer whose activated output the gradient belongs to:
// part[rows][2][Co] <- per-tile sums of dz and dz*xhat (rows = umi_conv_fwd_plan's stat_rows for this problem).
// UMI_ERR_UNSUPPORTED when the shape is not on the MFMA path: the caller then runs the separate kernels.
extern "C" int umi_conv_dgrad_bnred(const void* dy, int lddy, const void* wp8, void* da, int ldda, const void* ybn, int ldybn,
                                    const void* txbn, const float* rstd, float* part, int N, int H, int W, int Ci, int Co,
                                    int dtype, umi_stream_t stream) {
    if (!dy || !wp8 || !da || !ybn || !txbn || !rstd || !part || N <= 0 || H <= 0 || W <= 0) return UMI_ERR_BADARG;
    if (!umi_conv3x3_mfma_ok(N, H, W, Ci, Co, 3, 3, 1, 1, H, W, lddy, ldda, dtype, dtype, 0, nullptr) || ldybn % 8 || ldybn < Co)
        return UMI_ERR_UNSUPPORTED;
    if (((uintptr_t)dy | (uintptr_t)da | (uintptr_t)wp8 | (uintptr_t)ybn) & 15) return UMI_ERR_BADARG;
    return umi_conv3x3_mfma_bnred(dy, lddy, wp8, da, ldda, ybn, ldybn, txbn, rstd, part, N, H, W, Ci, Co, (hipStream_t)stream);
}

int umi_smallk_bnred_rows(long P, int Co);
int umi_smallk_fwd_bnred(const void* x, int ldx, const void* wp, void* y, int ldy, const void* ybn, int ldybn, const void* txbn,
                         const float* rstd, float* part, long P, int Ci, int Co, hipStream_t s, float* dW = nullptr, long s_co = 0,
                         long s_ci = 0, float out_scale = 1.f, void* ws = nullptr, size_t ws_bytes = 0);

// The same fusion for the data gradient of a narrow pointwise conv (the segmentation head `OutConv`, reference Model.py:89-93:
// Ci <= 8 logit channels -> Co feature channels): da = dl * W^T plus stage 1 of the BatchNorm+ReLU backward of the layer whose
// activated output feeds the head.  rows = umi_head_dgrad_bnred_rows(P, Co).  wp = the generic [1][Ci][Co] fp16 packing.
extern "C" int umi_head_dgrad_bnred_rows(long P, int Ci, int Co, int ldda, int dtype) {
    if (!umi_smallk_fwd_ok(Ci, Co, 1, 1, 1, 0, ldda, dtype, dtype, 0, nullptr, nullptr)) return 0;
    return umi_smallk_bnred_rows(P, Co);
}
extern "C" int umi_head_dgrad_bnred(const void* dl, int lddl, const void* wp, void* da, int ldda, const void* ybn, int ldybn,
                                    const void* txbn, const float* rstd, float* part, long P, int Ci, int Co, int dtype,
                                    umi_stream_t stream) {
    if (!dl || !wp || !da || !ybn || !txbn || !rstd || !part || P <= 0) return UMI_ERR_BADARG;
    if (!umi_smallk_fwd_ok(Ci, Co, 1, 1, 1, 0, ldda, dtype, dtype, 0, nullptr, nullptr) || ldybn % 8 || ldybn < Co || lddl < Ci)
        return UMI_ERR_UNSUPPORTED;
    return umi_smallk_fwd_bnred(dl, lddl, wp, da, ldda, ybn, ldybn, txbn, rstd, part, P, Ci, Co, (hipStream_t)stream);
}
// ... and the head's WEIGHT gradient as well (reference Model.py:89-93 under autograd: dW[k][c] = sum_p act(ybn)[p][c] * dl[p][k]; the
// head's input is the activated ybn): dW[k * s_co + c * s_ci] <- out_scale * that.  ws: umi_head_bwd_fused_ws_bytes(P, Ci, Co).
extern "C" size_t umi_head_bwd_fused_ws_bytes(long P, int Ci, int Co) {
    return (size_t)umi_smallk_bnred_rows(P, Co) * Co * Ci * sizeof(float);
}
extern "C" int umi_head_bwd_fused(const void* dl, int lddl, const void* wp, void* da, int ldda, const void* ybn, int ldybn,
                                  const void* txbn, const float* rstd, float* part, float* dW, long s_co, long s_ci, float out_scale,
                                  void* ws, size_t ws_bytes, long P, int Ci, int Co, int dtype, umi_stream_t stream) {
    if (!dl || !wp || !da || !ybn || !txbn || !rstd || !part || !dW || !ws || P <= 0) return UMI_ERR_BADARG;
    if (!umi_smallk_fwd_ok(Ci, Co, 1, 1, 1, 0, ldda, dtype, dtype, 0, nullptr, nullptr) || ldybn % 8 || ldybn < Co || lddl < Ci)
        return UMI_ERR_UNSUPPORTED;
    return umi_smallk_fwd_bnred(dl, lddl, wp, da, ldda, ybn, ldybn, txbn, rstd, part, P, Ci, Co, (hipStream_t)stream, dW, s_co, s_ci,
                                out_scale, ws, ws_bytes);
}

int umi_conv3x3_mfma_act(const void* x, int ldx, const void* tx, const void* wp8, const void* out_tx, void* y, int ldy, int N,
                         int H, int W, int Ci, int Co, hipStream_t s);

// Inference form of conv3x3 + BatchNorm + ReLU (reference Model.py:15-22 under model.eval(), test_mc3serousv5.py:877-887):
// y = max(out_tx.scale * conv(tx(x), w) + out_tx.shift, out_tx.lo) stored activated, no statistics.  UMI_ERR_UNSUPPORTED
// when the shape is not on the matrix-core path (the caller then uses umi_conv_fwd and the consumer-side transform).
extern "C" int umi_conv3x3_fwd_act(const void* x, int ldx, const void* tx, const void* wp8, const void* out_tx, void* y, int ldy,
                                   int N, int H, int W, int Ci, int Co, int dtype, umi_stream_t stream) {
    if (!x || !wp8 || !out_tx || !y || N <= 0 || H <= 0 || W <= 0 || ldx < Ci || ldy < Co) return UMI_ERR_BADARG;
    if (!umi_conv3x3_mfma_ok(N, H, W, Ci, Co, 3, 3, 1, 1, H, W, ldx, ldy, dtype, dtype, 0, nullptr)) return UMI_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)wp8 | (uintptr_t)out_tx) & 15) return UMI_ERR_BADARG;
    return umi_conv3x3_mfma_act(x, ldx, tx, wp8, out_tx, y, ldy, N, H, W, Ci, Co, (hipStream_t)stream);
}

// stage 2 of the BatchNorm backward reduction on partial rows produced by umi_conv_dgrad_bnred
int umi_colsum_rows_f16v(long M, int C);
bool umi_bn_stats_f16v(const void* x, int ldx, float* part, long M, int C, hipStream_t s);

// BatchNorm batch statistics of a stored fp16 tensor as partial rows part[rows][2][C] (sum, sum of squares) for umi_bn_finalize:
// for producers without a statistics epilogue (the pointwise MFMA convolution).  rows = umi_bn_stats_rows(M, C) (0: unsupported).
extern "C" int umi_bn_stats_rows(long M, int C) { return umi_colsum_rows_f16v(M, C); }
extern "C" int umi_bn_stats(const void* x, int ldx, float* part, long M, int C, int dtype, umi_stream_t stream) {
    if (!x || !part || M <= 0 || C <= 0 || ldx < C) return UMI_ERR_BADARG;
    if (dtype != UMI_F16 || !umi_bn_stats_f16v(x, ldx, part, M, C, (hipStream_t)stream)) return UMI_ERR_UNSUPPORTED;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

int umi_pool2_bwd_bnred_rows(int N, int H, int W, int C);
bool umi_pool2_bwd_bnred_f16v(const void* dp, int lddp, const void* x, int ldx, const void* tx, const float* rstd, void* da,
                              int ldda, int accumulate, float* part, int N, int H, int W, int C, hipStream_t s);

// MaxPool2d(2) backward into `da` + stage 1 of the BatchNorm backward of the pooled layer (x = its raw output): only when
// this is the last contribution to `da`.  *rows receives the partial rows written (part[rows][2][C]).
extern "C" int umi_pool2_bwd_bnred(const void* dpool, int lddp, const void* x, int ldx, const void* tx, const float* rstd,
                                   void* da, int ldda, int accumulate, float* part, int N, int H, int W, int C, int dtype,
                                   umi_stream_t stream) {
    if (!dpool || !x || !tx || !rstd || !da || !part || N <= 0 || H <= 0 || W <= 0 || C <= 0) return UMI_ERR_BADARG;
    if (dtype != UMI_F16) return UMI_ERR_UNSUPPORTED;
    if (!umi_pool2_bwd_bnred_f16v(dpool, lddp, x, ldx, tx, rstd, da, ldda, accumulate, part, N, H, W, C, (hipStream_t)stream))
        return UMI_ERR_UNSUPPORTED;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
extern "C" int umi_pool2_bwd_bnred_stat_rows(int N, int H, int W, int C) { return umi_pool2_bwd_bnred_rows(N, H, W, C); }

extern "C" int umi_bn_bwd_from_partials(const float* part, int rows, int C, float* sum_dz, float* sum_dzx, umi_stream_t stream) {
    if (!part || !sum_dz || !sum_dzx || rows <= 0 || C <= 0) return UMI_ERR_BADARG;
    umi_launch_reduce_rows2(part, rows, C, sum_dz, sum_dzx, 1.f, (hipStream_t)stream);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// UMI_TRACE_GENERIC=1: report every conv that lands on the generic (non-MFMA) kernels -- a tuning aid, off by default
static bool trace_generic() {
    static const bool on = [] { const char* e = getenv("UMI_TRACE_GENERIC"); return e && e[0] == '1'; }();
    return on;
}
#define UMI_TRACE(kind)                                                                                                  \
    if (trace_generic())                                                                                                 \
        fprintf(stderr, "[umi generic %s] N=%d H=%d W=%d Ci=%d Co=%d R=%d stride=%d pad=%d flags=%d\n", kind, N, H, W, Ci, Co, R, \
                stride, pad, flags)

extern "C" int umi_conv_fwd(const void* x, int ldx, const void* tx, const void* wp, const float* bias, void* y, int ldy,
                            float* stat_part, int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad,
                            int Ho, int Wo, int off_h, int off_w, int out_H, int out_W, int in_dtype, int out_dtype,
                            int flags, umi_stream_t stream) {
    if (!x || !wp || !y || N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || R <= 0 || S <= 0 || stride <= 0 ||
        Ho <= 0 || Wo <= 0 || ldx < Ci || ldy < Co || out_H <= 0 || out_W <= 0)
        return UMI_ERR_BADARG;
    if (flags & UMI_CONV_ACCUMULATE) {
        // only the pointwise / tap-gather MFMA kernel adds into y, and where it is eligible it is the path taken below (the 3x3
        // kernel's stride-1 pad-1 problems are never eligible for it)
        if (stat_part || !umi_conv1x1_mfma_ok(Ci, Co, R, S, stride, pad, ldx, ldy, in_dtype, out_dtype, flags))
            return UMI_ERR_UNSUPPORTED;
    }
    if (flags & UMI_CONV_DGRAD_STRIDED) {
        // x = dy of a strided conv (H x W), output grid = the conv's input image (Ho x Wo)
        if (flags & UMI_CONV_UPSAMPLE2) return UMI_ERR_BADARG;
        if (H != (Ho + 2 * pad - R) / stride + 1 || W != (Wo + 2 * pad - S) / stride + 1) return UMI_ERR_BADARG;
        if (out_H != Ho || out_W != Wo || off_h || off_w || stat_part) return UMI_ERR_BADARG;
        if (umi_conv1x1_mfma_ok(Ci, Co, R, S, stride, pad, ldx, ldy, in_dtype, out_dtype, flags)) {
            if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)wp) & 15) return UMI_ERR_BADARG;
            return umi_conv1x1_mfma(x, ldx, tx, wp, bias, y, ldy, N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, 0, 0, Ho, Wo,
                                    flags, (hipStream_t)stream);
        }
        UMI_TRACE("dgrad_strided");
        return umi_conv_fwd_generic(x, ldx, tx, wp, bias, y, ldy, stat_part, N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo,
                                    off_h, off_w, out_H, out_W, in_dtype, out_dtype, flags, (hipStream_t)stream);
    }
    if (!(flags & UMI_CONV_UPSAMPLE2)) {
        if (Ho != (H + 2 * pad - R) / stride + 1 || Wo != (W + 2 * pad - S) / stride + 1) return UMI_ERR_BADARG;
        if (out_H != Ho || out_W != Wo || off_h || off_w) return UMI_ERR_BADARG;
    } else {
        if (Ho != H || Wo != W) return UMI_ERR_BADARG;
    }
    if (umi_conv3x3_mfma_ok(N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, ldx, ldy, in_dtype, out_dtype, flags, bias)) {
        // the caller packed the weights for this path (umi_conv_fwd_plan said layout 1): misalignment is an error,
        // not a reason to silently reinterpret them
        if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)wp) & 15) return UMI_ERR_BADARG;
        return umi_conv3x3_mfma(x, ldx, tx, wp, y, ldy, stat_part, N, H, W, Ci, Co, (hipStream_t)stream);
    }
    if (umi_conv1x1_mfma_ok(Ci, Co, R, S, stride, pad, ldx, ldy, in_dtype, out_dtype, flags)) {
        if (stat_part) return UMI_ERR_UNSUPPORTED;      // no BatchNorm follows a pointwise conv on this path
        if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)wp) & 15) return UMI_ERR_BADARG;
        return umi_conv1x1_mfma(x, ldx, tx, wp, bias, y, ldy, N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, off_h, off_w,
                                out_H, out_W, flags, (hipStream_t)stream);
    }
    if (umi_stem_fwd_ok(Ci, Co, R, S, stride, pad, ldy, in_dtype, out_dtype, flags, bias))
        return umi_stem_fwd(x, ldx, tx, wp, y, ldy, stat_part, N, H, W, Ci, Co, (hipStream_t)stream);
    if ((!stat_part || out_dtype == UMI_F16) && umi_head_fwd_ok(Ci, Co, R, S, stride, pad, ldx, in_dtype, out_dtype, flags))
        return umi_head_fwd(x, ldx, tx, wp, bias, y, ldy, stat_part, (long)N * H * W, Ci, Co, out_dtype, (hipStream_t)stream);
    if (!stat_part && umi_smallk_fwd_ok(Ci, Co, R, S, stride, pad, ldy, in_dtype, out_dtype, flags, tx, bias))
        return umi_smallk_fwd(x, ldx, wp, y, ldy, (long)N * H * W, Ci, Co, (hipStream_t)stream);
    if (!stat_part && umi_root_fwd_ok(Ci, Co, R, S, stride, pad, ldy, in_dtype, out_dtype, flags, tx, bias)) {
        const int st = umi_root_fwd(x, ldx, wp, y, ldy, N, H, W, Ho, Wo, Co, (hipStream_t)stream);
        if (st != UMI_ERR_UNSUPPORTED) return st;          // (rows too wide for its LDS staging: the generic kernel below)
    }
    if (!stat_part && umi_head3_fwd_ok(Ci, Co, R, S, stride, pad, ldx, in_dtype, out_dtype, flags))
        return umi_head3_fwd(x, ldx, tx, wp, bias, y, ldy, N, H, W, Ci, Co, (hipStream_t)stream);
    UMI_TRACE("fwd");
    return umi_conv_fwd_generic(x, ldx, tx, wp, bias, y, ldy, stat_part, N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo,
                                off_h, off_w, out_H, out_W, in_dtype, out_dtype, flags, (hipStream_t)stream);
}

extern "C" size_t umi_conv_wgrad_ws_bytes(int N, int Ho, int Wo, int Ci, int Co, int R, int S, int dtype, int flags) {
    // the call picks its path from more arguments than this query has; size for whichever needs more
    size_t g = umi_conv_wgrad_generic_ws_bytes(N, Ho, Wo, Ci, Co, R, S);
    if (umi_wgrad3x3_mfma_ok(N, Ho, Wo, Ci, Co, R, S, 1, 1, Ho, Wo, 8, 8, dtype, flags, nullptr)) {
        size_t m = umi_wgrad3x3_mfma_ws_bytes(N, Ho, Wo, Ci, Co);
        if (m > g) g = m;
    }
    if (umi_wgrad1x1_mfma_ok((long)N * Ho * Wo, Ci, Co, R, S, 1, 0, 8, 8, dtype, flags, nullptr)) {
        size_t m = umi_wgrad1x1_mfma_ws_bytes((long)N * Ho * Wo, Ci, Co);
        if (m > g) g = m;
    }
    if (umi_wgradT_mfma_ok(2 * Ho, 2 * Wo, Ci, Co, R, S, 2, 0, Ho, Wo, 8, 8, dtype, flags, nullptr)) {
        size_t m = umi_wgradT_mfma_ws_bytes(N, Ho, Wo, Ci, Co);
        if (m > g) g = m;
    }
    if (dtype == UMI_F16 && Ci % 64 == 0 && Co % 64 == 0 && R * S <= 49 && !(flags & UMI_CONV_FORCE_GENERIC)) {
        size_t m = umi_wgrad_gather_mfma_ws_bytes(N, Ho, Wo, Ci, Co, R, S);
        if (m > g) g = m;
    }
    if (umi_stem_wgrad_ok(Ci, Co, R, S, 1, 1, 8, dtype, flags, nullptr)) {
        size_t m = umi_stem_wgrad_ws_bytes(N, Ho, Wo, Ci, Co);
        if (m > g) g = m;
    }
    if (umi_head_wgrad_ok(Ci, Co, R, S, 1, 0, 8, dtype, flags, nullptr)) {
        size_t m = umi_head_wgrad_ws_bytes((long)N * Ho * Wo, Ci, Co);
        if (m > g) g = m;
    }
    if (umi_root_wgrad_ok(Ci, Co, R, S, 2, 3, 8, dtype, flags, nullptr, nullptr)) {
        size_t m = umi_root_wgrad_ws_bytes(N, Ho, Wo, Co);
        if (m > g) g = m;
    }
    if (umi_head3_wgrad_ok(Ci, Co, R, S, 1, 1, 8, dtype, flags, nullptr)) {
        size_t m = umi_head3_wgrad_ws_bytes((long)N * Ho * Wo, Ci, Co);
        if (m > g) g = m;
    }
    return g;
}

// umi_conv_wgrad whose final split-K reduction is RECORDED in *out instead of launched (umi_wgrad_reduce_group runs many of
// them at once).  `ws` must then stay untouched until that launch; out->part == NULL when the path taken had no separate
// reduction (the gradient is already in dW).
void umi_wgrad_defer_set(void* slot);
extern "C" int umi_conv_wgrad_deferred(const void* x, int ldx, const void* txa, const void* dy, int lddy, const void* txb,
                                       float* dW, long s_co, long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci,
                                       int Co, int R, int S, int stride, int pad, int Ho, int Wo, int dtype, int flags, void* ws,
                                       size_t ws_bytes, umi_wgrad_pending* out, umi_stream_t stream) {
    if (!out) return UMI_ERR_BADARG;
    out->part = nullptr;
    umi_wgrad_defer_set(out);
    const int st = umi_conv_wgrad(x, ldx, txa, dy, lddy, txb, dW, s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, R, S, stride, pad, Ho,
                                  Wo, dtype, flags, ws, ws_bytes, stream);
    umi_wgrad_defer_set(nullptr);
    return st;
}

// ConvTranspose2d(2,2) weight gradient (called like umi_conv_wgrad_deferred for it: x = d(up), dy = the ConvT's input) that also
// produces the BIAS gradient d bias[c] = out_scale * sum over pixels of x[.][c] from the operand tiles it stages (reference
// Model.py:56-57 under autograd; replaces umi_colsum's pass over x).  UMI_ERR_UNSUPPORTED (nothing launched) where the 2x2 / stride-2
// matrix-core kernel does not take the problem: run umi_colsum + umi_conv_wgrad instead.  `out` may be NULL (reduce at once).
void umi_wgradT_bias_set(float* bias_out);
extern "C" int umi_conv_wgrad_bias(const void* x, int ldx, const void* dy, int lddy, const void* txb, float* dW, long s_co,
                                   long s_ci, long s_t, float* dbias, float out_scale, int N, int H, int W, int Ci, int Co, int Ho,
                                   int Wo, int dtype, int flags, void* ws, size_t ws_bytes, umi_wgrad_pending* out,
                                   umi_stream_t stream) {
    if (!dbias) return UMI_ERR_BADARG;
    if (!umi_wgradT_mfma_ok(H, W, Ci, Co, 2, 2, 2, 0, Ho, Wo, ldx, lddy, dtype, flags, nullptr)) return UMI_ERR_UNSUPPORTED;
    if (out) { out->part = nullptr; umi_wgrad_defer_set(out); }
    umi_wgradT_bias_set(dbias);
    const int st = umi_conv_wgrad(x, ldx, nullptr, dy, lddy, txb, dW, s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, 2, 2, 2, 0, Ho, Wo,
                                  dtype, flags, ws, ws_bytes, stream);
    umi_wgradT_bias_set(nullptr);
    umi_wgrad_defer_set(nullptr);
    return st;
}

// umi_conv_wgrad for `n` pointwise convs / nn.Linear layers of ONE shape (M rows, Ci -> Co, same row strides) in one launch:
// dW[i][co*s_co + ci*s_ci] = out_scale * sum_p x[i][p][ci] * dy[i][p][co].  No workspace: each output tile is owned by one
// workgroup (fixed summation order).  UMI_ERR_UNSUPPORTED where the pointwise matrix-core kernel does not apply.
extern "C" int umi_conv_wgrad_group(int n, const void* const* x, int ldx, const void* const* dy, int lddy, float* const* dW,
                                    long s_co, long s_ci, float out_scale, long M, int Ci, int Co, int dtype,
                                    umi_stream_t stream) {
    if (n <= 0 || !x || !dy || !dW || M <= 0 || Ci <= 0 || Co <= 0 || ldx < Ci || lddy < Co) return UMI_ERR_BADARG;
    for (int i = 0; i < n; ++i)
        if (!x[i] || !dy[i] || !dW[i]) return UMI_ERR_BADARG;
    if (!umi_wgrad1x1_mfma_ok(M, Ci, Co, 1, 1, 1, 0, ldx, lddy, dtype, 0, nullptr)) return UMI_ERR_UNSUPPORTED;
    return umi_wgrad1x1_mfma_group(n, x, ldx, dy, lddy, dW, s_co, s_ci, out_scale, M, Ci, Co, (hipStream_t)stream);
}

// Weight gradient of a 3x3 conv whose output feeds BatchNorm(+ReLU), fused with stage 3 of that BatchNorm's backward: the
// gradient of the raw conv output, dz = gamma*rstd*(relu'(z)*dA - mean(dz) - xhat*mean(dz*xhat)) (umi_bn_bwd_apply's
// expression, bit for bit), is formed while dA is staged for the matrix cores and written to `dz` once for the
// data-gradient kernel.  `da` is left untouched.  UMI_ERR_UNSUPPORTED where the warp-specialised 3x3 kernel does not apply:
// the caller then runs umi_bn_bwd_apply + umi_conv_wgrad.
extern "C" int umi_conv_wgrad_bnapply(const void* x, int ldx, const void* txa, const void* da, int ldda, const void* y, int ldy,
                                      const void* tx_bn, const float* rstd, const float* sum_dz, const float* sum_dzx,
                                      void* dz, int lddz, float* dW, long s_co, long s_ci, long s_t, float out_scale, int N,
                                      int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int dtype, int flags,
                                      void* ws, size_t ws_bytes, umi_stream_t stream) {
    if (!dz) {
        // dz == NULL: nothing else needs dz (the layer's input takes no gradient: the network's first conv) -- the narrow-input
        // weight-gradient kernel forms it on the fly and never stores it
        if (!x || !da || !y || !tx_bn || !rstd || !sum_dz || !sum_dzx || !dW || !ws || N <= 0 || H <= 0 || W <= 0 || Ci <= 0 ||
            Co <= 0 || ldx < Ci || ldda < Co || ldy < Co)
            return UMI_ERR_BADARG;
        if (!umi_stem_wgrad_ok(Ci, Co, R, S, stride, pad, ldda, dtype, flags, nullptr) || ldy % 8) return UMI_ERR_UNSUPPORTED;
        return umi_stem_wgrad_bnapply(x, ldx, txa, da, ldda, y, ldy, tx_bn, rstd, sum_dz, sum_dzx, dW, s_co, s_ci, s_t, out_scale,
                                      N, H, W, Ci, Co, ws, ws_bytes, (hipStream_t)stream);
    }
    if (!x || !da || !y || !tx_bn || !rstd || !sum_dz || !sum_dzx || !dz || !dW || !ws || N <= 0 || H <= 0 || W <= 0 ||
        Ci <= 0 || Co <= 0 || ldx < Ci || ldda < Co || ldy < Co || lddz < Co || dz == da || dz == y)
        return UMI_ERR_BADARG;
    if (!umi_wgrad3x3_mfma_ok(N, H, W, Ci, Co, R, S, stride, pad, H, W, ldx, ldda, dtype, flags, nullptr) || ldy % 8 || lddz % 8 ||
        (long)H * W * (ldy > lddz ? ldy : lddz) * 2 >= 0x7FFFFFF0L)
        return UMI_ERR_UNSUPPORTED;
    return umi_wgrad3x3_mfma_bnapply(x, ldx, txa, da, ldda, y, ldy, tx_bn, rstd, sum_dz, sum_dzx, dz, lddz, dW, s_co, s_ci, s_t,
                                     out_scale, N, H, W, Ci, Co, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int umi_conv_wgrad(const void* x, int ldx, const void* txa, const void* dy, int lddy, const void* txb,
                              float* dW, long s_co, long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci,
                              int Co, int R, int S, int stride, int pad, int Ho, int Wo, int dtype, int flags, void* ws,
                              size_t ws_bytes, umi_stream_t stream) {
    if (!x || !dy || !dW || !ws || N <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0 || ldx < Ci || lddy < Co)
        return UMI_ERR_BADARG;
    if (umi_wgrad3x3_mfma_ok(N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, ldx, lddy, dtype, flags, txb))
        return umi_wgrad3x3_mfma(x, ldx, txa, dy, lddy, dW, s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, ws, ws_bytes,
                                 (hipStream_t)stream);
    if (umi_wgrad1x1_mfma_ok((long)N * H * W, Ci, Co, R, S, stride, pad, ldx, lddy, dtype, flags, txb))
        return umi_wgrad1x1_mfma(x, ldx, txa, dy, lddy, dW, s_co, s_ci, s_t, out_scale, (long)N * H * W, Ci, Co, ws, ws_bytes,
                                 (hipStream_t)stream);
    if (umi_wgradT_mfma_ok(H, W, Ci, Co, R, S, stride, pad, Ho, Wo, ldx, lddy, dtype, flags, txa))
        return umi_wgradT_mfma(x, ldx, dy, lddy, txb, dW, s_co, s_ci, s_t, out_scale, N, Ho, Wo, Ci, Co, ws, ws_bytes,
                               (hipStream_t)stream);
    if (umi_wgrad_gather_mfma_ok(N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, ldx, lddy, dtype, flags, txb))
        return umi_wgrad_gather_mfma(x, ldx, txa, dy, lddy, dW, s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, R, S, stride, pad,
                                     Ho, Wo, ws, ws_bytes, (hipStream_t)stream);
    if (umi_stem_wgrad_ok(Ci, Co, R, S, stride, pad, lddy, dtype, flags, txb))
        return umi_stem_wgrad(x, ldx, txa, dy, lddy, dW, s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, ws, ws_bytes,
                              (hipStream_t)stream);
    if (umi_head_wgrad_ok(Ci, Co, R, S, stride, pad, ldx, dtype, flags, txb))
        return umi_head_wgrad(x, ldx, txa, dy, lddy, dW, s_co, s_ci, s_t, out_scale, (long)N * H * W, Ci, Co, ws,
                              ws_bytes, (hipStream_t)stream);
    if (umi_root_wgrad_ok(Ci, Co, R, S, stride, pad, lddy, dtype, flags, txa, txb))
        return umi_root_wgrad(x, ldx, dy, lddy, dW, s_co, s_ci, s_t, out_scale, N, H, W, Ho, Wo, Co, ws, ws_bytes,
                              (hipStream_t)stream);
    if (umi_head3_wgrad_ok(Ci, Co, R, S, stride, pad, ldx, dtype, flags, txb))
        return umi_head3_wgrad(x, ldx, txa, dy, lddy, dW, s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, ws, ws_bytes,
                               (hipStream_t)stream);
    UMI_TRACE("wgrad");
    return umi_conv_wgrad_generic(x, ldx, txa, dy, lddy, txb, dW, s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, R, S,
                                  stride, pad, Ho, Wo, dtype, ws, ws_bytes, (hipStream_t)stream);
}
