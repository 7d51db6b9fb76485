/* unetmi.h -- C ABI of libunetmi.so: the MI355X (gfx950) U-Net / TransUNet hot path.
 *
 * The reference (caki35/UNet-Torch) is pure Python and has no FFI of its own: every
 * arithmetic op on its hot path is a torch.nn call.  The drop-in boundary
 * is therefore the reference's Python module API (`Model.UNet`, `Trainer.Trainer`, ...,
 * mirrored under unet-torch_amd/), and THIS header is the C ABI that mirror binds with
 * ctypes.  Each entry point cites the reference call site whose arithmetic it replaces.
 *
 * Conventions
 *  - Plain pointers and sizes only; device pointers are borrowed from the caller
 *    (torch tensor .data_ptr()); nothing is allocated, freed or synchronised inside.
 *  - Every function enqueues on the caller's `stream` and returns an int:
 *    0 = ok, <0 = UMI_ERR_*, >0 = hipError_t of the failed launch.  Nothing throws.
 *  - Activations are NHWC with an explicit pixel stride `ld` (elements), so a tensor may
 *    be a channel slice of a wider concat buffer.  dtype: UMI_F32 / UMI_F16 storage,
 *    accumulation is always fp32.
 *  - Input transform `tx` (nullable): one float4 per input channel {mean, scale, shift, lo};
 *    a conv/pool/etc. consumes  max(fma(x, scale, shift), lo)  of the stored value (`mean` is only
 *    read by the BatchNorm backward kernels),
 *    i.e. BatchNorm-apply + ReLU of the producer is fused into the consumer's load
 *    (lo = 0 for ReLU channels, -inf for pass-through channels of a concat buffer).
 *    Zero padding is applied AFTER the transform, as in the reference
 *    (Conv2d(padding=1) sees zeros of the activated tensor).
 *  - Workspaces: `*_ws_bytes()` returns the bytes the matching call needs.
 */
#ifndef UNETMI_H
#define UNETMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* umi_stream_t;          /* hipStream_t */

enum { UMI_F32 = 0, UMI_F16 = 1 };
enum { UMI_OK = 0, UMI_ERR_BADARG = -1, UMI_ERR_UNSUPPORTED = -2, UMI_ERR_WORKSPACE = -3 };
/* umi_conv_fwd flags */
enum { UMI_CONV_UPSAMPLE2 = 1,   /* ConvTranspose2d(k=2,s=2): tap t=(dy,dx) scatters to (2h+dy+off_h, 2w+dx+off_w) */
       UMI_CONV_FORCE_GENERIC = 2, /* never take the MFMA fast path (used by tests to cross-check it) */
       UMI_CONV_DGRAD_STRIDED = 4, /* data gradient of a stride>1 conv: x = dy [N,H,W,Ci:=Co_fwd], y = dx [N,Ho,Wo,Co:=Ci_fwd]
                                      with (R,S,stride,pad) of the FORWARD conv; weights packed [R*S][Co_fwd][Ci_fwd] unflipped */
       UMI_CONV_ACCUMULATE = 8     /* y += conv(...) instead of y = conv(...): the second gradient contribution of a tensor with two
                                      consumers (attention gate, reference Model.py:268-289: g and x each feed two branches).  Only
                                      the pointwise / tap-gather MFMA kernel implements it: umi_conv_fwd_plan and umi_conv_fwd return
                                      UMI_ERR_UNSUPPORTED for any other problem, nothing is written */ };

int umi_version(void);
const char* umi_arch(void);          /* "gfx950" */

/* Knob (process-wide) for same-process A/B timing of the conv3x3 matrix-core kernel's forms (csrc/conv_mfma.hip, tools/ab_conv.py):
 *   1 (default)  v_mfma_f32_16x16x32_f16 form where Co % 128 == 0 and Ci % 32 == 0, the 32x32x16 form elsewhere;
 *   2            the 32x32x16 form everywhere (the arithmetic order of rounds 1-2);
 *   3            the 16x16x32 form on the 64-output-channel tiles too (Ci % 32 == 0).
 * The forms differ by fp32 summation order only (<= 1 fp16 ulp on ~0.2 % of the outputs).  Returns the previous value; values
 * outside 1..8 only query.  Initial value: 1 or env UMI_CONV3X3_IMPL. */
int umi_tune_conv3x3_impl(int impl);

/* Re-layout of a weight tensor into the kernels' [T][K][N] packing (dtype storage):
 *   dst[(t*Kpad + k)*Npad + n] = src[t'*st + k*sk + n*sn],  t' = flip_t ? T-1-t : t,
 * zero for k >= K or n >= N.  Used for Conv2d OIHW weights (reference Model.py:15-20),
 * their 180-degree-rotated transpose (dgrad) and ConvTranspose2d [Cin,Cout,2,2]
 * (reference Model.py:56-57). */
int umi_pack_kn(const float* src, void* dst, int T, int K, int N, long st, long sk, long sn,
                int flip_t, int Kpad, int Npad, int dtype, umi_stream_t stream);

/* Same packing for the MFMA kernels: dst[t][k/8][n][k%8] (8 consecutive k contiguous). */
int umi_pack_kn8(const float* src, void* dst, int T, int K, int N, long st, long sk, long sn,
                 int flip_t, int Kpad, int Npad, int dtype, umi_stream_t stream);

/* Convolution forward, NHWC: replaces nn.Conv2d / F.conv2d on the hot path
 * (reference Model.py:15-16,19-20,89; vit_seg_modeling.py:145-148,265-272,321;
 *  vit_seg_modeling_resnet_skip.py:22-25) and, with UMI_CONV_UPSAMPLE2,
 * nn.ConvTranspose2d(k=2,s=2) (reference Model.py:56-57,67).
 *   y[n,ho,wo,co] = bias[co] + sum_{r,s,ci} tx(x[n,ho*stride-pad+r,wo*stride-pad+s,ci]) * wp[r*S+s][ci][co]
 * `stat_part` (nullable) receives per-pixel-tile partial sums for BatchNorm statistics,
 * layout [rows][2][Co] (sum, sum of squares of the *stored* y), rows = umi_conv_stat_rows().
 * Also used for dgrad (weights packed with flip_t, pad' = R-1-pad, stride 1). */
int umi_conv_fwd(const void* x, int ldx, const void* tx, const void* wp, const float* bias,
                 void* y, int ldy, float* stat_part,
                 int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad,
                 int Ho, int Wo, int off_h, int off_w, int out_H, int out_W,
                 int in_dtype, int out_dtype, int flags, umi_stream_t stream);
/* A ViT-block linear with its elementwise tail in the GEMM epilogue (reference TransUnet/vit_seg_modeling.py:113-119 Mlp.forward,
 * :177-187 Block.forward), fp16 storage:
 *   epi 1: y = x W + b, y2 = dropout(GELU(y)), mask        (fc1: y is kept for the GELU backward)
 *   epi 2: y = dropout(x W + b) + aux, mask                (fc2 / attention output projection, aux = the block's residual)
 * x [M, Ci], y / y2 / aux [M, Co] (ld in elements), wp8 = umi_pack_kn8 of the [Co, Ci] weight, mask: M*Co bytes (as umi_dropout).
 * Same values, random stream and roundings as umi_conv_fwd followed by umi_dropout_fused.  UMI_ERR_UNSUPPORTED where the
 * pointwise matrix-core kernel does not apply (run the two calls). */
int umi_linear_fused(const void* x, int ldx, const void* wp8, const float* bias, void* y, int ldy, long M, int Ci, int Co,
                     int epi, float p, unsigned seed, const unsigned* seed_dev, void* mask, const void* aux, int ldaux,
                     void* y2, int ldy2, int dtype, umi_stream_t stream);

/* Which kernel umi_conv_fwd will take for this problem: *layout = 0 -> weights packed with umi_pack_kn,
 * 1 -> umi_pack_kn8 (MFMA path: fp16, 3x3, stride 1, pad 1, Ci%16==0, Co%64==0, no bias, 16-B aligned
 * rows); *stat_rows = rows of `stat_part` the call will write. */
int umi_conv_fwd_plan(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad,
                      int ldx, int ldy, int in_dtype, int out_dtype, int flags, int has_bias,
                      int* layout, int* stat_rows);

/* BatchNorm2d training statistics -> consumer transform (reference Model.py:17,21 =
 * nn.BatchNorm2d: biased batch variance for normalisation, unbiased into running_var,
 * momentum 0.1).  tx_out[c] = {mean, gamma*rstd, beta - mean*gamma*rstd, 0}; running stats updated in place
 * when non-null. */
int umi_bn_finalize(const float* stat_part, int rows, int C, double count,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var,
                    void* tx_out, float* rstd_out, umi_stream_t stream);

/* MaxPool2d(2) of the transformed tensor (reference Model.py:36,42). Floor mode. */
int umi_pool2_fwd(const void* x, int ldx, const void* tx, void* y, int ldy,
                  int N, int H, int W, int C, int dtype, umi_stream_t stream);
/* Its backward: routes dpool to the first arg-max of tx(x) in each 2x2 window (PyTorch
 * tie rule); accumulate != 0 adds into `da` (skip-connection gradient already there). */
int umi_pool2_bwd(const void* dpool, int lddp, const void* x, int ldx, const void* tx,
                  void* da, int ldda, int accumulate,
                  int N, int H, int W, int C, int dtype, umi_stream_t stream);

/* BatchNorm+ReLU backward (autograd of reference Model.py:17-18,21-22).
 *   z = tx(y) before the clamp, dz = da * [z > 0],  xhat = (y - mean) * rstd
 *   reduce: sum_dz[c] = sum dz, sum_dzx[c] = sum dz*xhat      (d beta, d gamma)
 *   apply : dy = gamma*rstd * (dz - sum_dz/M - xhat*sum_dzx/M)   written over `da`. */
size_t umi_bn_bwd_ws_bytes(long M, int C);
int umi_bn_bwd_reduce(const void* da, int ldda, const void* y, int ldy, const void* tx,
                      const float* rstd, float* sum_dz, float* sum_dzx,
                      long M, int C, int dtype, void* ws, size_t ws_bytes, umi_stream_t stream);
int umi_bn_bwd_apply(void* da, int ldda, const void* y, int ldy, const void* tx,
                     const float* rstd, const float* sum_dz, const float* sum_dzx,
                     long M, int C, int dtype, umi_stream_t stream);

/* Fusion of the two previous steps for a DoubleConv's inner layer (reference Model.py:15-22): the 3x3 data gradient that
 * produces `da` = d(activated output) of a BatchNorm+ReLU layer also emits stage 1 of that layer's reduction
 * (`ybn`/`txbn`/`rstd` = that layer's raw output, transform and 1/std): part[rows][2][Co] per-tile sums of dz and
 * dz*xhat, rows = umi_conv_fwd_plan(...)'s stat_rows.  umi_bn_bwd_from_partials finishes the reduction; umi_bn_bwd_apply
 * follows as usual.  Returns UMI_ERR_UNSUPPORTED off the MFMA path (caller falls back to the separate calls). */
int umi_conv_dgrad_bnred(const void* dy, int lddy, const void* wp8, void* da, int ldda, const void* ybn, int ldybn,
                         const void* txbn, const float* rstd, float* part, int N, int H, int W, int Ci, int Co,
                         int dtype, umi_stream_t stream);

/* The same fusion for the data gradients that run on the pointwise / tap-gather matrix-core kernel -- ConvTranspose2d(2,2)'s
 * (a stride-2 2x2 conv over d(up), reference Model.py:56-57 under autograd), 1x1 convs', strided convs' (UMI_CONV_DGRAD_STRIDED):
 * y = conv(x, wp8) as umi_conv_fwd computes it, plus part[rows][2][Co] of the BatchNorm(+ReLU) layer whose activated output y is
 * the gradient of.  rows = umi_conv_gather_bnred_rows(...); 0 = the problem is not on that kernel (use the separate calls). */
int umi_conv_gather_bnred_rows(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo,
                               int ldx, int ldy, int dtype, int flags);
int umi_conv_gather_bnred(const void* x, int ldx, const void* wp8, void* y, int ldy, const void* ybn, int ldybn,
                          const void* txbn, const float* rstd, float* part, int N, int H, int W, int Ci, int Co, int R, int S,
                          int stride, int pad, int Ho, int Wo, int dtype, int flags, umi_stream_t stream);

/* The same fusion for the data gradient of the narrow pointwise head (`OutConv`, reference Model.py:89-93; Ci <= 8 logit channels):
 * da[p][c] = sum_k dl[p][k] * w[k][c] plus part[rows][2][Co], rows = umi_head_dgrad_bnred_rows(...) (0 = shape not taken, call
 * umi_conv_fwd and umi_bn_bwd_reduce).  wp: the generic [1][Ci][Co] fp16 packing (umi_pack_kn). */
int umi_head_dgrad_bnred_rows(long P, int Ci, int Co, int ldda, int dtype);
int umi_head_dgrad_bnred(const void* dl, int lddl, const void* wp, void* da, int ldda, const void* ybn, int ldybn,
                         const void* txbn, const float* rstd, float* part, long P, int Ci, int Co, int dtype,
                         umi_stream_t stream);
/* ... and the head's weight gradient from the same pass (its input is the activated ybn): dW[k * s_co + c * s_ci] <- out_scale *
 * sum_p tx(ybn[p][c]) * dl[p][k] (k < Ci logit channels, c < Co feature channels).  ws: umi_head_bwd_fused_ws_bytes. */
size_t umi_head_bwd_fused_ws_bytes(long P, int Ci, int Co);
int umi_head_bwd_fused(const void* dl, int lddl, const void* wp, void* da, int ldda, const void* ybn, int ldybn,
                       const void* txbn, const float* rstd, float* part, float* dW, long s_co, long s_ci, float out_scale,
                       void* ws, size_t ws_bytes, long P, int Ci, int Co, int dtype, umi_stream_t stream);

/* Inference form of Conv2d(3, pad 1, bias=False) -> BatchNorm2d (running statistics) -> ReLU (reference Model.py:15-22 under
 * model.eval(), the evaluation loop test_mc3serousv5.py:877-887): the layer's own transform out_tx[Co] = {mean, scale, shift,
 * lo} is applied to the fp32 accumulators in the epilogue, y = max(scale * conv(tx(x), w) + shift, lo) is stored ACTIVATED
 * and consumed without a transform; no statistics are produced.  fp16, matrix-core shapes only (UMI_ERR_UNSUPPORTED
 * otherwise: use umi_conv_fwd + the consumer-side transform). */
int umi_conv3x3_fwd_act(const void* x, int ldx, const void* tx, const void* wp8, const void* out_tx, void* y, int ldy,
                        int N, int H, int W, int Ci, int Co, int dtype, umi_stream_t stream);
/* The same fusion on the other producer of such a gradient: MaxPool2d(2) backward (reference Model.py:36) routes `dpool` into
 * `da` (accumulate != 0: adds to the skip-connection gradient already there) and, being the LAST contribution to `da`, also
 * emits stage 1 of the BatchNorm backward of the pooled layer (x = its raw output, tx / rstd its transform and 1/std):
 * part[rows][2][C], rows = umi_pool2_bwd_bnred_stat_rows().  fp16, even H and W, C % 8 == 0; else UMI_ERR_UNSUPPORTED
 * (caller uses umi_pool2_bwd + umi_bn_bwd_reduce). */
/* BatchNorm batch statistics of a stored fp16 tensor (for a producer without a statistics epilogue: the pointwise MFMA conv
 * of the attention gates, reference Model.py:268-289): part[rows][2][C] = per-block sums / sums of squares, the layout
 * umi_bn_finalize consumes; rows = umi_bn_stats_rows(M, C) (0 = unsupported: C % 8 != 0). */
int umi_bn_stats_rows(long M, int C);
int umi_bn_stats(const void* x, int ldx, float* part, long M, int C, int dtype, umi_stream_t stream);
int umi_pool2_bwd_bnred_stat_rows(int N, int H, int W, int C);
int umi_pool2_bwd_bnred(const void* dpool, int lddp, const void* x, int ldx, const void* tx, const float* rstd, void* da,
                        int ldda, int accumulate, float* part, int N, int H, int W, int C, int dtype, umi_stream_t stream);
int umi_bn_bwd_from_partials(const float* part, int rows, int C, float* sum_dz, float* sum_dzx, umi_stream_t stream);

/* Weight gradient of umi_conv_fwd (autograd of the reference convs):
 *   dW[co*s_co + ci*s_ci + t*s_t] = out_scale * sum_{n,ho,wo} txa(x[...,ci]) * txb(dy[n,ho,wo,co])
 * fp32 output in the parameter's own layout (OIHW: s_co=Ci*R*S, s_ci=R*S, s_t=1).
 * Deterministic: split-K partial slabs in `ws`, reduced in fixed order. */
size_t umi_conv_wgrad_ws_bytes(int N, int Ho, int Wo, int Ci, int Co, int R, int S, int dtype, int flags);
int umi_conv_wgrad(const void* x, int ldx, const void* txa, const void* dy, int lddy, const void* txb,
                   float* dW, long s_co, long s_ci, long s_t, float out_scale,
                   int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad,
                   int Ho, int Wo, int dtype, int flags, void* ws, size_t ws_bytes, umi_stream_t stream);
/* umi_conv_wgrad with its final split-K reduction recorded instead of launched: a reduction is a ~8-us launch over a few
 * hundred KB, a U-Net step has 22 and a TransUNet step 62; umi_wgrad_reduce_group runs 16 per launch (same arithmetic and
 * order, identical results).  `ws` must be the call's own and stay untouched until then.  out->part == NULL: the path taken
 * reduces inside its kernel, dW is already final. */
typedef struct umi_wgrad_pending {
    const float* part;
    float* dW;
    long s_co, s_ci, s_t;
    float scale;
    int splits, RS, Ci, Co;
} umi_wgrad_pending;
int umi_conv_wgrad_deferred(const void* x, int ldx, const void* txa, const void* dy, int lddy, const void* txb,
                            float* dW, long s_co, long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci, int Co,
                            int R, int S, int stride, int pad, int Ho, int Wo, int dtype, int flags, void* ws, size_t ws_bytes,
                            umi_wgrad_pending* out, umi_stream_t stream);
int umi_wgrad_reduce_group(int n, const void* items /* umi_wgrad_pending[n], host */, umi_stream_t stream);
/* Weight AND bias gradient of ConvTranspose2d(2,2) (reference Model.py:56-57 under autograd) in one pass over d(up): the arguments of
 * umi_conv_wgrad_deferred for that layer (x = d(up) [N,H,W,Ci], dy = the ConvT's input [N,Ho,Wo,Co] with its transform txb) plus
 * dbias[Ci] <- out_scale * column sums of x.  `out` NULL: the split-K reduction runs at once.  UMI_ERR_UNSUPPORTED (nothing
 * launched) where the 2x2 / stride-2 matrix-core kernel does not apply: umi_colsum + umi_conv_wgrad. */
int umi_conv_wgrad_bias(const void* x, int ldx, const void* dy, int lddy, const void* txb, float* dW, long s_co, long s_ci,
                        long s_t, float* dbias, float out_scale, int N, int H, int W, int Ci, int Co, int Ho, int Wo, int dtype,
                        int flags, void* ws, size_t ws_bytes, umi_wgrad_pending* out, umi_stream_t stream);
/* umi_conv_wgrad (R = S = 1, no transforms) for `n` layers of one shape in one launch: the per-layer weight gradients of a
 * ViT encoder (reference vit_seg_modeling.py:58-62,100-101, twelve Blocks), whose pixel dimension (tokens) is too short to
 * fill the chip one layer at a time without a deep split-K.  x / dy / dW: HOST arrays of n device pointers.  No workspace,
 * deterministic.  UMI_ERR_UNSUPPORTED where the pointwise matrix-core kernel does not apply (call umi_conv_wgrad per layer). */
int umi_conv_wgrad_group(int n, const void* const* x, int ldx, const void* const* dy, int lddy, float* const* dW, long s_co,
                         long s_ci, float out_scale, long M, int Ci, int Co, int dtype, umi_stream_t stream);
/* umi_conv_wgrad of a 3x3/stride-1/pad-1 conv fused with umi_bn_bwd_apply of the BatchNorm(+ReLU) that follows the conv
 * (reference Model.py:14-21 DoubleConv backward: autograd runs cudnn_batch_norm_backward, then convolution_backward):
 * `da` = gradient of the activated output (read only), `y`/`tx_bn`/`rstd`/`sum_dz`/`sum_dzx` as umi_bn_bwd_apply takes them;
 * `dz` (a separate tensor) receives exactly what umi_bn_bwd_apply would have left in `da` (dz == NULL: the layer's input takes no
 * gradient, nothing else reads dz -- accepted for the network's first conv, Ci <= 4, whose kernel then never stores it), dW what umi_conv_wgrad would
 * have produced from it.  UMI_ERR_UNSUPPORTED where the fused kernel does not apply (run the two calls instead). */
int umi_conv_wgrad_bnapply(const void* x, int ldx, const void* txa, const void* da, int ldda, const void* y, int ldy,
                           const void* tx_bn, const float* rstd, const float* sum_dz, const float* sum_dzx, void* dz,
                           int lddz, float* dW, long s_co, long s_ci, long s_t, float out_scale, int N, int H, int W,
                           int Ci, int Co, int R, int S, int stride, int pad, int dtype, int flags, void* ws,
                           size_t ws_bytes, umi_stream_t stream);

/* Per-channel sum over pixels (bias gradients): out[c] = out_scale * sum_p x[p, c]. */
size_t umi_colsum_ws_bytes(long M, int C);
int umi_colsum(const void* x, int ldx, float* out, float out_scale, long M, int C, int dtype,
               void* ws, size_t ws_bytes, umi_stream_t stream);
/* umi_colsum for n fp16 tensors of one shape (C % 8 == 0) in two launches per 16: the bias gradients of a ViT's twelve
 * encoder layers.  xs / outs: HOST arrays of device pointers; ws >= min(n,16) * umi_colsum_ws_bytes(M, C).
 * UMI_ERR_UNSUPPORTED where the vectorised kernel does not apply (call umi_colsum per tensor). */
int umi_colsum_group(int n, const void* const* xs, int ldx, float* const* outs, float out_scale, long M, int C, int dtype,
                     void* ws, size_t ws_bytes, umi_stream_t stream);

/* Materialise an activation (storage + consumer transform) as NCHW fp32: the tensor a
 * reference block returns (e.g. DoubleConv.forward, reference Model.py:25-26). */
int umi_materialize_nchw(const void* x, int ldx, const void* tx, float* y_nchw,
                         int N, int H, int W, int C, int dtype, umi_stream_t stream);

/* ---- TransUNet path (reference TransUnet/vit_seg_modeling.py, vit_seg_modeling_resnet_skip.py) -------------------- */

/* StdConv2d weight standardisation and its backward (resnet_skip.py:20-23): per output channel over K = Ci*R*S,
 * biased variance, w_std = (w - mean) / sqrt(var + eps). fp32 parameters. */
int umi_wstd_fwd(const float* w, float* wstd, float* rstd, int Co, int K, float eps, umi_stream_t stream);
int umi_wstd_bwd(const float* wstd, const float* rstd, const float* g, float* dw, int Co, int K, umi_stream_t stream);
/* The same for ALL StdConv2d layers of a model in one launch each way (a R50 hybrid has 52).  `descs`: DEVICE array sorted
 * by blk0; an entry owns Co workgroups.  Backward: the gradient w.r.t. the standardised weights of entry i is read at
 * g_base + off, the parameter gradient written at dw_base + off (flat per-step buffers, so the table never changes) or,
 * where the entry's `dw` is not NULL, there (e.g. the parameter's slot in a data-parallel gradient bucket). */
typedef struct umi_wstd_desc {
    const float* w;
    float* ws;
    float* rstd;
    long off;
    int Co, K;
    float eps;
    int blk0;
    float* dw;
} umi_wstd_desc;
int umi_wstd_fwd_multi(const void* descs, int n_desc, int total_rows, umi_stream_t stream);
int umi_wstd_bwd_multi(const void* descs, int n_desc, int total_rows, const float* g_base, float* dw_base, umi_stream_t stream);

/* GroupNorm (+ optional residual add, + optional ReLU) on NHWC, y = [relu](gn(x) [+ res]) (resnet_skip.py:47-58,68-73).
 * mean/rstd: [N*G] saved for backward.  Backward: dx (and dres = masked dy when dres != NULL), dgamma/dbeta scaled by
 * out_scale; `y` is the forward OUTPUT (ReLU mask).
 * part_out (backward): NULL, or [N][2][C] floats that receive the per-sample sums (dz*xhat, dz) per channel; dgamma / dbeta
 * may then be NULL and are formed later for many layers at once by umi_gn_param_grads_group (parts / dgammas / dbetas: HOST
 * arrays of n device pointers, Cs: n channel counts; out[c] = out_scale * sum over the N samples). */
size_t umi_gn_fwd_ws_bytes(int N, long HW, int C);
int umi_gn_fwd(const void* x, int ldx, const float* gamma, const float* beta, const void* res, int ldr, void* y, int ldy,
               float* mean, float* rstd, int relu, int N, long HW, int C, int G, float eps, int dtype,
               void* ws, size_t ws_bytes, umi_stream_t stream);
size_t umi_gn_bwd_ws_bytes(int N, long HW, int C, int G);
int umi_gn_bwd(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx, const float* mean,
               const float* rstd, const float* gamma, int relu, void* dx, int lddx, void* dres, int lddr,
               float* dgamma, float* dbeta, float out_scale, int N, long HW, int C, int G, int dtype,
               void* ws, size_t ws_bytes, float* part_out, umi_stream_t stream);
int umi_gn_param_grads_group(int n, const float* const* parts, const int* Cs, int N, float* const* dgammas,
                             float* const* dbetas, float out_scale, umi_stream_t stream);

/* MaxPool2d(kernel 3, stride 2, pad 0) (resnet_skip.py:147) and its backward (first-max tie rule).
 * idx: NULL, or N*Ho*Wo*C bytes (8-byte aligned, fp16 with C % 8 == 0 only: UMI_ERR_UNSUPPORTED otherwise) in which the forward
 * records the winning tap 0..8 of every output element; a backward given the same buffer reads it instead of re-deriving
 * every window's maximum from the input. */
int umi_pool3s2_fwd(const void* x, int ldx, void* y, int ldy, void* idx, int N, int H, int W, int C, int dtype,
                    umi_stream_t stream);
int umi_pool3s2_bwd(const void* dy, int lddy, const void* x, int ldx, const void* idx, void* dx, int lddx, int N, int H, int W,
                    int C, int dtype, umi_stream_t stream);

/* LayerNorm over the last dim (vit_seg_modeling.py:172-173,232; eps 1e-6) and backward.  * umi_ln_bwd with dgamma == dbeta == NULL leaves its partial rows [*rows_out][2][C] in `ws` (then a buffer of the caller's
 * that stays alive) for umi_gn_param_grads_group (N = *rows_out) to sum for many layers in one launch. */
int umi_ln_fwd(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* mean, float* rstd,
               long M, int C, float eps, int dtype, umi_stream_t stream);
size_t umi_ln_bwd_ws_bytes(long M, int C);
int umi_ln_bwd(const void* dy, int lddy, const void* x, int ldx, const float* gamma, const float* mean,
               const float* rstd, void* dx, int lddx, float* dgamma, float* dbeta, float out_scale, long M, int C,
               int dtype, void* ws, size_t ws_bytes, int* rows_out, umi_stream_t stream);

/* Elementwise: mode 0 y = gelu(x) (exact erf form, vit_seg_modeling.py:115); 1 y = g * gelu'(x); 2 y = x + g;
 * 3 y = x + g[row % bcast_rows] (position embedding, vit_seg_modeling.py:163). */
int umi_elementwise(int mode, const void* x, int ldx, const void* g, int ldg, void* y, int ldy, long M, int C,
                    long bcast_rows, int dtype, umi_stream_t stream);

/* Dropout (vit_seg_modeling.py:103,151; U-Net Down / Up, Model.py:37,80-81): forward writes a byte mask (own counter-based
 * RNG stream) and y = keep ? tx(x) / (1-p) : 0 (tx: nullable consumer transform of x, forward only); backward (x = dy)
 * reuses the mask.  seed_dev (nullable): device counter added into the stream seed inside the kernel, so a step replayed
 * from a captured HIP graph still draws a fresh mask every replay. */
int umi_dropout(const void* x, int ldx, void* y, int ldy, void* mask, int backward, float p, unsigned seed, long M, int C,
                int dtype, const void* tx, const unsigned* seed_dev, umi_stream_t stream);
/* umi_dropout fused with the GELU before it and / or the residual add after it (the MLP and the two residual joins of a
 * transformer Block, reference vit_seg_modeling.py:113-119,177-187); fp16 with C % 8 == 0 only (UMI_ERR_UNSUPPORTED otherwise):
 *   forward : y = dropout(gelu ? GELU(x) : x) + (aux ? aux : 0)
 *   backward: y = dropout'(x) * (gelu ? GELU'(aux) : 1), aux = the forward's x.  Same mask bytes / random stream as umi_dropout. */
int umi_dropout_fused(const void* x, int ldx, void* y, int ldy, void* mask, int backward, float p, unsigned seed, long M, int C,
                      int dtype, const unsigned* seed_dev, const void* aux, int ldaux, int gelu, umi_stream_t stream);

/* Multi-head softmax attention (vit_seg_modeling.py:73-91): q,k,v,o are [B, N, heads*D] token tensors (row stride ld),
 * head h = channels [h*D, (h+1)*D); softmax(q k^T / sqrt(D)) v.  lse/delta: [B*heads*N] fp32 scratch kept for backward. */
int umi_attn_fwd(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse, int B, int N, int heads,
                 int D, int dtype, umi_stream_t stream);
int umi_attn_bwd(const void* q, const void* k, const void* v, int ld, const void* o, const void* dO, int ldo,
                 const float* lse, void* dq, void* dk, void* dv, int ldd, float* delta, int B, int N, int heads, int D,
                 int dtype, umi_stream_t stream);

/* UpsamplingBilinear2d(scale_factor=2), align_corners=True (vit_seg_modeling.py:307): forward x[N,H,W,C] -> y[N,2H,2W,C];
 * backward (x = dy [N,2H,2W,C]) -> y = dx [N,H,W,C] (deterministic gather form). */
int umi_bilinear2x(const void* x, int ldx, const void* tx, void* y, int ldy, int backward, int N, int H, int W, int C,
                   int dtype, umi_stream_t stream);   /* tx: consumer transform of x, forward only (nullable) */

/* Fused training loss 'dice_bce_mc' (reference loss.py:488-500, DiceLoss loss.py:215-251) on NCHW fp32 logits [N,C,HW],
 * C <= 8: 0.5 * CrossEntropy + 0.5 * mean_c(1 - (2*sum(p*t)+1e-5)/(sum(p*p)+sum(t*t)+1e-5)), p = softmax(logits).
 * target [N,HW] class indices; target_dtype 0 = int64, 1 = float32, 2 = uint8, 3 = int32.
 * fwd fills stats[3*C + 2] = {sum p*t | sum p*p | sum t | CE sum | loss}; bwd writes dlogits = gout[0] * d loss / d logits
 * (gout: device pointer to the upstream scalar gradient, or NULL for 1). */
size_t umi_dice_ce_ws_bytes(int N, int C, long HW);
int umi_dice_ce_fwd(const float* logits, const void* target, int target_dtype, int N, int C, long HW, float* stats, void* ws,
                    size_t ws_bytes, umi_stream_t stream);
int umi_dice_ce_bwd(const float* logits, const void* target, int target_dtype, const float* stats, const float* gout, int N,
                    int C, long HW, float* dlogits, umi_stream_t stream);

/* Multi-tensor optimizer step: torch.optim.SGD / torch.optim.Adam arithmetic (reference train.py:341-347) on every parameter
 * tensor of a model in ONE launch.  `descs` is a DEVICE array of n_desc umi_optim_desc sorted by blk0; a tensor of n elements
 * occupies ceil(n / umi_optim_block_elems()) consecutive blocks starting at blk0; total_blocks = the sum.
 *   SGD:  g += wd*p; m = first_step ? g : momentum*m + (1-dampening)*g; g = nesterov ? g + momentum*m : m; p -= lr*g
 *         (s0 = momentum buffer, NULL when momentum == 0; s1 unused)
 *   Adam: g += wd*p; m += (1-beta1)*(g-m); v = beta2*v + (1-beta2)*g*g; p -= step_size * m / (sqrt(v)/bc2_sqrt + eps)
 *         (s0 = exp_avg, s1 = exp_avg_sq; step_size = lr / (1-beta1^t), bc2_sqrt = sqrt(1-beta2^t), computed by the host) */
typedef struct umi_optim_desc {
    float* p;            /* parameter (fp32 master), updated in place */
    const float* g;      /* gradient */
    float* s0;
    float* s1;
    long n;              /* elements */
    int blk0;
    int pad_;
} umi_optim_desc;
int umi_optim_block_elems(void);
/* Copies a descriptor table from PINNED (device-mapped) host memory to device memory with a kernel, so that the upload is a
 * plain kernel node inside a HIP-graph capture; nbytes and both pointers are multiples of 16. */
int umi_table_upload(const void* host_pinned, void* dev, size_t nbytes, umi_stream_t stream);
/* hyper-parameters are doubles (Python floats) and are rounded to fp32 where torch rounds them */
int umi_optim_sgd_multi(const void* descs, int n_desc, int total_blocks, double lr, double momentum, double dampening,
                        double weight_decay, int nesterov, int first_step, umi_stream_t stream);
int umi_optim_adam_multi(const void* descs, int n_desc, int total_blocks, double step_size, double beta1, double beta2,
                         double bc2_sqrt, double eps, double weight_decay, umi_stream_t stream);

/* Graph-safe hyper-parameters (reference Trainer.py:719-726: the poly learning-rate rule rewrites the LR after every step,
 * torch.optim.Adam advances `step` on the host).  A captured HIP graph freezes kernel ARGUMENTS, so the values that change
 * from step to step live in a device-resident umi_optim_hyper block per param group instead:
 *   umi_optim_hyper_pre   before the update: Adam t += 1, step_size_f = lr / (1 - beta1^t), bc2_sqrt_f = sqrt(1 - beta2^t)
 *                         (formed in double, rounded to fp32 as torch rounds its Python floats); lr_f = (float)lr
 *   umi_optim_*_multi_dev the same arithmetic as umi_optim_*_multi with lr / step_size / bc2_sqrt read from the block
 *   umi_optim_hyper_poly  after the update: lr = base_lr * (1 - iter / max_iter)^power; iter += 1  (pre-increment iter, as the
 *                         reference does)
 * The host fills the block once (umi_table_upload) and reads it back when it wants param_group['lr'] / state['step']. */
typedef struct umi_optim_hyper {
    double lr, base_lr, iter, max_iter, power, adam_t, beta1, beta2;
    float lr_f, step_size_f, bc2_sqrt_f, pad_;
    double pad2_[2];
} umi_optim_hyper;                         /* 96 bytes */
size_t umi_optim_hyper_bytes(void);
int umi_optim_hyper_pre(void* hyper, int adam, umi_stream_t stream);
int umi_optim_hyper_poly(void* hyper, umi_stream_t stream);
int umi_optim_sgd_multi_dev(const void* descs, int n_desc, int total_blocks, const void* hyper, double momentum,
                            double dampening, double weight_decay, int nesterov, int first_step, umi_stream_t stream);
int umi_optim_adam_multi_dev(const void* descs, int n_desc, int total_blocks, const void* hyper, double beta1, double beta2,
                             double eps, double weight_decay, umi_stream_t stream);

/* umi_pack_kn / umi_pack_kn8 of many weight tensors in one launch (all the convolution weights of a model after an
 * optimizer step).  `descs`: DEVICE array sorted by blk0; an entry owns ceil(T*Kpad*Npad / umi_pack_block_elems()) blocks. */
typedef struct umi_pack_desc {
    const float* src;
    void* dst;
    long st, sk, sn;
    int T, K, N, flip_t, Kpad, Npad, k8, blk0;
    int ldn, pad_;   /* ldn > Npad: the entry fills columns [0, Npad) of rows of length ldn (dst already offset to its first
                        column): several source matrices packed side by side into one operand (Q/K/V projections); 0 = Npad */
} umi_pack_desc;
int umi_pack_block_elems(void);
int umi_pack_kn_multi(const void* descs, int n_desc, int total_blocks, int dtype, umi_stream_t stream);

/* Inference pre-/post-processing, the steps either side of the network in the reference's test scripts.
 * umi_znorm_hwc (test_mc3serousv5.py:115-127 `preprocess`): one HWC image, src_dtype 0 = uint8 (cv2.imread) or
 *   1 = float32, C <= 4 -> out_chw[c'][p] = (img[p][c] - mean_c) / std_c as fp32, c' = reverse_channels ? C-1-c : c
 *   (BGR -> RGB); mean and population std per channel in fp64 (numpy semantics), two-pass variance.
 * umi_argmax_mask (test_mc3serousv5.py:883-885 softmax -> argmax -> uint8): mask[n][p] = argmax_c logits[n][c][p], first
 *   maximum wins; softmax is monotone and therefore skipped. */
/* Additive attention gate of UNet_attention (reference Model.py:265-305, Attention_block.forward :297-305); the 1x1
 * convolutions and BatchNorms of the block go through umi_conv_fwd / umi_bn_finalize, these are the fused elementwise steps:
 *   add2_relu: y = max(txa(a) + txb(b), 0)   (E = relu(Q1 + X1), Model.py:302);  bwd: da = db = dy * [y > 0]
 *   gate:      y[m][c] = txx(x[m][c]) * sigmoid(txp(p[m]))   (x * A, Model.py:303-304; p has ONE channel, txp one row)
 *              bwd: dx = dy * A,  dp[m] = A (1 - A) * sum_c dy[m][c] * txx(x[m][c])   (gradient w.r.t. txp(p), i.e. the
 *              BatchNorm output before the sigmoid). */
int umi_add2_relu_fwd(const void* a, int lda, const void* txa, const void* b, int ldb, const void* txb, void* y, int ldy,
                      long M, int C, int dtype, umi_stream_t stream);
int umi_add2_relu_bwd(const void* dy, int lddy, const void* y, int ldy, void* da, int ldda, void* db, int lddb, long M, int C,
                      int dtype, umi_stream_t stream);
int umi_gate_fwd(const void* x, int ldx, const void* txx, const void* p, const void* txp, void* y, int ldy, long M, int C,
                 int dtype, umi_stream_t stream);
int umi_gate_bwd(const void* dy, int lddy, const void* x, int ldx, const void* txx, const void* p, const void* txp, void* dx,
                 int lddx, void* dp, long M, int C, int dtype, umi_stream_t stream);

size_t umi_znorm_ws_bytes(void);
int umi_znorm_hwc(const void* img, int src_dtype, float* out_chw, long HW, int C, int reverse_channels, void* ws,
                  size_t ws_bytes, umi_stream_t stream);
int umi_argmax_mask(const float* logits, unsigned char* mask, int N, int C, long HW, umi_stream_t stream);
/* umi_zoom_cubic_hwc (test_mc3serousv5.py:100-113, the resize of `preprocess`): scipy.ndimage.zoom(img, (out_h / H, out_w / W[, 1]),
 *   order=3) of one HWC image, C <= 4, src_dtype as umi_znorm_hwc; `out` has the input's type and [out_h][out_w][C] elements.
 *   B-spline prefilter (float64, mirror boundaries), corner-aligned sampling, uint8 results rounded and clipped: SciPy's algorithm,
 *   restated and pinned against SciPy by oracle/ref_resize.py.  ws: umi_zoom_cubic_ws_bytes(H, W, C). */
size_t umi_zoom_cubic_ws_bytes(int H, int W, int C);
int umi_zoom_cubic_hwc(const void* img, int src_dtype, void* out, int H, int W, int C, int out_h, int out_w, void* ws,
                       size_t ws_bytes, umi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* UNETMI_H */
