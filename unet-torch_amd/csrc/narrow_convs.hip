// HBM-bound narrow convolutions at the two ends of TransUNet (fp16 storage), where one side has too few channels for
// the matrix cores to help:
//   * root : ResNetV2's first conv, StdConv2d(3, 64, kernel 7, stride 2, pad 3) (resnet_skip.py:120): forward and weight
//            gradient.  K = 147 per output value, input re-read from cache by every tap.
//   * head3: SegmentationHead Conv2d(16, n_classes, kernel 3, pad 1) (vit_seg_modeling.py:317-323): forward (fp32 logits,
//            bias, the producer's BN+ReLU applied on load) and weight gradient.  (Its data gradient has Ci = n_classes
//            <= 4 and runs on the stem kernel.)
// Same scheme as stem_head.hip: thread = pixel x 8 wide-side channels (16-B accesses), per-channel reductions through
// registers -> LDS -> one deterministic partial row per workgroup, reduced in fixed order by wgrad_reduce_kernel.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

void umi_launch_wgrad_reduce(const float* part, int splits, int RS, int Ci, int Co, float* dW, long s_co, long s_ci,
                             long s_t, float scale, hipStream_t st);

namespace {

constexpr int ROOT_ROWS = 4;         // output rows per workgroup (root forward: amortises the 37-KB weight staging)
constexpr int ROOT_WROWS = 8;        // output rows per workgroup (root weight gradient)
constexpr int WG_PPB = 2048;         // output pixels per workgroup (head weight-gradient kernel)

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
bool groups_ok(int C) { return C % 8 == 0 && C / 8 <= 64 && (256 % (C / 8)) == 0; }

// ---------------------------------------------------------------------------------------------------------------------
// root forward: y[n][ho][wo][co] = sum_{ty,tx,ci} x[n][S*ho+ty-PAD][S*wo+tx-PAD][ci] * w[(ty*R+tx)*CI + ci][co]
// Workgroup = ROOT_ROWS output rows of one image.  The R input rows an output row reads are staged in LDS once (zero-padded
// left / right / outside the image: no bounds checks in the tap loop), the weights once per workgroup as fp32; thread =
// 8 output channels x 4 output pixels (wo = lane, lane + PL, ...): per tap 2 weight reads + 4 input reads for 32 FMAs.
// (The first version read every input value from global memory, 2 bytes at a time, inside the tap loop: one memory round trip
// per tap row, 207 us for the 24 x 112 x 112 x 64 output of the R50 root.)
template <int CI, int R, int STRIDE, int PAD>
__global__ __launch_bounds__(256) void root_fwd_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ wp,
                                                       half_t* __restrict__ y, int ldy, int N, int H, int W, int Ho, int Wo,
                                                       int Co) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];      // [R*R*CI][Co] fp32, then xs [R][Wp][CI] fp16
    const int Wp = W + 2 * PAD + STRIDE;                              // padded row (+ slack for the pixels past Wo of the last lane)
    half_t* xs = reinterpret_cast<half_t*>(wsm + R * R * CI * Co);
    const int tid = threadIdx.x;
    const int G = Co >> 3, PL = 256 / G;
    const int cg = tid % G, pl = tid / G;
    for (int i = tid; i < R * R * CI * Co; i += 256) wsm[i] = (float)wp[i];
    const int rows_per_img = (Ho + ROOT_ROWS - 1) / ROOT_ROWS;
    const int n = blockIdx.x / rows_per_img, ho0 = (blockIdx.x % rows_per_img) * ROOT_ROWS;
    for (int ho = ho0; ho < ho0 + ROOT_ROWS && ho < Ho; ++ho) {
        __syncthreads();                                              // (weights staged / previous row's readers done)
        for (int i = tid; i < R * Wp * CI; i += 256) {
            const int ci = i % CI, col = (i / CI) % Wp, ty = i / (CI * Wp);
            const int hi = ho * STRIDE + ty - PAD, wi = col - PAD;
            xs[i] = (hi >= 0 && hi < H && wi >= 0 && wi < W) ? x[((long)((long)n * H + hi) * W + wi) * ldx + ci] : (half_t)0.f;
        }
        __syncthreads();
        for (int w0 = 0; w0 < Wo; w0 += 4 * PL) {
            float acc[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[u][j] = 0.f;
            int col[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int wo = w0 + pl + u * PL;
                col[u] = (wo < Wo ? wo : 0) * STRIDE;                 // (lanes past the row compute pixel 0 and do not store)
            }
            for (int ty = 0; ty < R; ++ty)
#pragma unroll
                for (int tx_ = 0; tx_ < R; ++tx_)
#pragma unroll
                    for (int ci = 0; ci < CI; ++ci) {
                        const float4* wrow = reinterpret_cast<const float4*>(wsm + ((ty * R + tx_) * CI + ci) * Co + cg * 8);
                        const float4 w0v = wrow[0], w1v = wrow[1];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float v = (float)xs[(ty * Wp + col[u] + tx_) * CI + ci];
                            acc[u][0] = fmaf(v, w0v.x, acc[u][0]); acc[u][1] = fmaf(v, w0v.y, acc[u][1]);
                            acc[u][2] = fmaf(v, w0v.z, acc[u][2]); acc[u][3] = fmaf(v, w0v.w, acc[u][3]);
                            acc[u][4] = fmaf(v, w1v.x, acc[u][4]); acc[u][5] = fmaf(v, w1v.y, acc[u][5]);
                            acc[u][6] = fmaf(v, w1v.z, acc[u][6]); acc[u][7] = fmaf(v, w1v.w, acc[u][7]);
                        }
                    }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int wo = w0 + pl + u * PL;
                if (wo >= Wo) continue;
                half8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[u][j];
                *reinterpret_cast<half8*>(y + ((long)((long)n * Ho + ho) * Wo + wo) * ldy + cg * 8) = o;
            }
        }
    }
}

// root weight gradient partials, one tap row (ty) per blockIdx.y, ROOT_WROWS output rows of one image per blockIdx.x:
// part[((blk*R*R + ty*R + tx)*CI + ci)*Co + co].  The one input row a (output row, ty) pair reads is staged in LDS
// (zero-padded), the gradient rows of 4 pixels per thread are loaded up front: the first version took both from global memory
// pixel by pixel, one round trip each (446 us on the R50 root).
template <int CI, int R, int STRIDE, int PAD>
__global__ __launch_bounds__(256) void root_wgrad_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ dy,
                                                         int lddy, float* __restrict__ part, int N, int H, int W, int Ho,
                                                         int Wo, int Co) {
    extern __shared__ __attribute__((aligned(16))) float red_[];     // red [256][9] floats, then xs [Wp][CI] fp16
    float (*red)[9] = reinterpret_cast<float (*)[9]>(red_);
    half_t* xs = reinterpret_cast<half_t*>(red_ + 256 * 9);
    const int Wp = W + 2 * PAD + STRIDE;
    const int tid = threadIdx.x, ty = blockIdx.y;
    const int G = Co >> 3, PL = 256 / G;
    const int cg = tid % G, pl = tid / G;
    const int rows_per_img = (Ho + ROOT_WROWS - 1) / ROOT_WROWS;
    const int n = blockIdx.x / rows_per_img, ho0 = (blockIdx.x % rows_per_img) * ROOT_WROWS;
    float acc[R][CI][8];
#pragma unroll
    for (int t = 0; t < R; ++t)
#pragma unroll
        for (int c = 0; c < CI; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[t][c][j] = 0.f;
    for (int ho = ho0; ho < ho0 + ROOT_WROWS && ho < Ho; ++ho) {
        const int hi = ho * STRIDE + ty - PAD;
        if (hi < 0 || hi >= H) continue;                              // (uniform over the workgroup)
        __syncthreads();
        for (int i = tid; i < Wp * CI; i += 256) {
            const int ci = i % CI, wi = i / CI - PAD;
            xs[i] = (wi >= 0 && wi < W) ? x[((long)((long)n * H + hi) * W + wi) * ldx + ci] : (half_t)0.f;
        }
        __syncthreads();
        for (int w0 = 0; w0 < Wo; w0 += 4 * PL) {
            half8 g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int wo = w0 + pl + u * PL;
#pragma unroll
                for (int j = 0; j < 8; ++j) g[u][j] = (half_t)0.f;
                if (wo < Wo) g[u] = *reinterpret_cast<const half8*>(dy + ((long)((long)n * Ho + ho) * Wo + wo) * lddy + cg * 8);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int wo = w0 + pl + u * PL;
                const int col = (wo < Wo ? wo : 0) * STRIDE;          // (past the row: zero gradient, any column)
                float gf[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) gf[j] = (float)g[u][j];
#pragma unroll
                for (int tx_ = 0; tx_ < R; ++tx_)
#pragma unroll
                    for (int c = 0; c < CI; ++c) {
                        const float v = (float)xs[(col + tx_) * CI + c];
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[tx_][c][j] = fmaf(v, gf[j], acc[tx_][c][j]);
                    }
            }
        }
    }
    for (int t = 0; t < R; ++t)
        for (int c = 0; c < CI; ++c) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) red[tid][j] = acc[t][c][j];
            __syncthreads();
            if (tid < Co) {
                float a = 0.f;
                for (int k = 0; k < PL; ++k) a += red[k * G + (tid >> 3)][tid & 7];
                part[(((long)blockIdx.x * R * R + ty * R + t) * CI + c) * Co + tid] = a;
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// head3 forward: logits[p][k] = bias[k] + sum_{tap,c} tx(x[p + tap][c]) * w[(tap*C + c)*NC + k]   (fp32 out)
template <int NC>
__global__ __launch_bounds__(256) void head3x3_fwd_kernel(const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx,
                                                          const half_t* __restrict__ wp, const float* __restrict__ bias,
                                                          float* __restrict__ y, int ldy, int N, int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];      // [9*C][NC]
    const int G = C >> 3;                                   // lanes per pixel (power of two <= 64)
    for (int i = threadIdx.x; i < 9 * C * NC; i += 256) wsm[i] = (float)wp[i];
    __syncthreads();
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gt % G);
    const long stride_p = ((long)gridDim.x * 256) / G;
    float4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = tx ? tx[cg * 8 + j] : make_float4(0.f, 1.f, 0.f, -INFINITY);
    const long P = (long)N * H * W;
    const long Pr = ((P + stride_p - 1) / stride_p) * stride_p;      // keep whole pixel groups in the shuffle
    for (long p = gt / G; p < Pr; p += stride_p) {
        float acc[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[k] = 0.f;
        if (p < P) {
            const int n = (int)(p / ((long)H * W));
            const int r = (int)(p - (long)n * H * W);
            const int yy = r / W, xx = r - yy * W;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int hi = yy + tap / 3 - 1, wi = xx + tap % 3 - 1;
                if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;          // zero padding AFTER the transform
                const half8 v = *reinterpret_cast<const half8*>(x + ((long)((long)n * H + hi) * W + wi) * ldx + cg * 8);
                const float* wt = wsm + (tap * C + cg * 8) * NC;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = umi_tx((float)v[j], t[j]);
#pragma unroll
                    for (int k = 0; k < NC; ++k) acc[k] = fmaf(a, wt[j * NC + k], acc[k]);
                }
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
            for (int k = 0; k < NC; ++k) acc[k] += __shfl_xor(acc[k], o);
        if (cg == 0 && p < P) {
#pragma unroll
            for (int k = 0; k < NC; ++k) y[p * ldy + k] = acc[k] + (bias ? bias[k] : 0.f);
        }
    }
}

// head3 weight gradient partials, one tap per blockIdx.y: part[((blk*9 + tap)*C + c)*NC + k] = sum_p tx(x[p+tap][c]) * dl[p][k]
template <int NC>
__global__ __launch_bounds__(256) void head3x3_wgrad_kernel(const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx,
                                                            const half_t* __restrict__ dl, int lddl, float* __restrict__ part,
                                                            int N, int H, int W, int C) {
    __shared__ float red[256][9];
    const int tid = threadIdx.x, tap = blockIdx.y;
    const int G = C >> 3, PL = 256 / G;
    const int cg = tid % G, pl = tid / G;
    const long P = (long)N * H * W;
    const long p0 = (long)blockIdx.x * WG_PPB;
    float4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = tx ? tx[cg * 8 + j] : make_float4(0.f, 1.f, 0.f, -INFINITY);
    float acc[NC][8];
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    for (long p = p0 + pl; p < p0 + WG_PPB && p < P; p += PL) {
        const int n = (int)(p / ((long)H * W));
        const int r = (int)(p - (long)n * H * W);
        const int yy = r / W, xx = r - yy * W;
        const int hi = yy + tap / 3 - 1, wi = xx + tap % 3 - 1;
        if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
        const half8 v = *reinterpret_cast<const half8*>(x + ((long)((long)n * H + hi) * W + wi) * ldx + cg * 8);
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = umi_tx((float)v[j], t[j]);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const float d = (float)dl[p * lddl + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(a[j], d, acc[k][j]);
        }
    }
    for (int k = 0; k < NC; ++k) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid][j] = acc[k][j];
        __syncthreads();
        if (tid < C) {
            float s = 0.f;
            for (int q = 0; q < PL; ++q) s += red[q * G + (tid >> 3)][tid & 7];
            part[(((long)blockIdx.x * 9 + tap) * C + tid) * NC + k] = s;
        }
    }
}

int grid_for(long items) {
    long g = (items + 255) / 256;
    if (g > 16384) g = 16384;
    return g < 1 ? 1 : (int)g;
}

}  // namespace

// ---- dispatch helpers used by api.hip -------------------------------------------------------------------------------
bool umi_root_fwd_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldy, int in_dtype, int out_dtype, int flags,
                     const void* tx, const float* bias) {
    if (flags & (UMI_CONV_UPSAMPLE2 | UMI_CONV_FORCE_GENERIC | UMI_CONV_DGRAD_STRIDED)) return false;
    return in_dtype == UMI_F16 && out_dtype == UMI_F16 && !tx && !bias && R == 7 && S == 7 && stride == 2 && pad == 3 &&
           Ci == 3 && groups_ok(Co) && ldy % 8 == 0 && (size_t)49 * 3 * Co * 4 <= 48 * 1024;
}
int umi_root_fwd(const void* x, int ldx, const void* wp, void* y, int ldy, int N, int H, int W, int Ho, int Wo, int Co,
                 hipStream_t s) {
    if (!al16(y)) return UMI_ERR_BADARG;
    const int blocks = N * ((Ho + ROOT_ROWS - 1) / ROOT_ROWS);
    const size_t smem = (size_t)49 * 3 * Co * 4 + (size_t)7 * (W + 8) * 3 * 2;
    if (smem > 64 * 1024) return UMI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((root_fwd_kernel<3, 7, 2, 3>), dim3(blocks), dim3(256), smem, s, (const half_t*)x, ldx,
                       (const half_t*)wp, (half_t*)y, ldy, N, H, W, Ho, Wo, Co);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

bool umi_root_wgrad_ok(int Ci, int Co, int R, int S, int stride, int pad, int lddy, int dtype, int flags, const void* txa,
                       const void* txb) {
    if (flags & UMI_CONV_FORCE_GENERIC) return false;
    return dtype == UMI_F16 && !txa && !txb && R == 7 && S == 7 && stride == 2 && pad == 3 && Ci == 3 && groups_ok(Co) &&
           Co <= 256 && lddy % 8 == 0;
}
size_t umi_root_wgrad_ws_bytes(int N, int Ho, int Wo, int Co) {
    const long blocks = (long)N * ((Ho + ROOT_WROWS - 1) / ROOT_WROWS);
    return (size_t)blocks * 49 * 3 * Co * sizeof(float);
}
int umi_root_wgrad(const void* x, int ldx, const void* dy, int lddy, float* dW, long s_co, long s_ci, long s_t, float out_scale,
                   int N, int H, int W, int Ho, int Wo, int Co, void* ws, size_t ws_bytes, hipStream_t s) {
    if (ws_bytes < umi_root_wgrad_ws_bytes(N, Ho, Wo, Co)) return UMI_ERR_WORKSPACE;
    if (!al16(dy)) return UMI_ERR_BADARG;
    const int blocks = N * ((Ho + ROOT_WROWS - 1) / ROOT_WROWS);
    const size_t smem = (size_t)256 * 9 * 4 + (size_t)(W + 8) * 3 * 2;
    hipLaunchKernelGGL((root_wgrad_kernel<3, 7, 2, 3>), dim3(blocks, 7), dim3(256), smem, s, (const half_t*)x, ldx,
                       (const half_t*)dy, lddy, (float*)ws, N, H, W, Ho, Wo, Co);
    UMI_LAUNCH_CHECK();
    umi_launch_wgrad_reduce((const float*)ws, blocks, 49, 3, Co, dW, s_co, s_ci, s_t, out_scale, s);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

bool umi_head3_fwd_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldx, int in_dtype, int out_dtype, int flags) {
    if (flags & (UMI_CONV_UPSAMPLE2 | UMI_CONV_FORCE_GENERIC | UMI_CONV_DGRAD_STRIDED)) return false;
    return in_dtype == UMI_F16 && out_dtype == UMI_F32 && R == 3 && S == 3 && stride == 1 && pad == 1 && Co >= 1 && Co <= 4 &&
           groups_ok(Ci) && Ci <= 64 && ldx % 8 == 0;
}
int umi_head3_fwd(const void* x, int ldx, const void* tx, const void* wp, const float* bias, void* y, int ldy, int N, int H,
                  int W, int Ci, int Co, hipStream_t s) {
    if (!al16(x)) return UMI_ERR_BADARG;
    const int grid = grid_for((long)N * H * W * (Ci / 8));
    const size_t smem = (size_t)9 * Ci * Co * 4;
#define GO(NC) hipLaunchKernelGGL(head3x3_fwd_kernel<NC>, dim3(grid), dim3(256), smem, s, (const half_t*)x, ldx, (const float4*)tx, (const half_t*)wp, bias, (float*)y, ldy, N, H, W, Ci)
    switch (Co) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; default: GO(4); }
#undef GO
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

bool umi_head3_wgrad_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldx, int dtype, int flags, const void* txb) {
    if (flags & UMI_CONV_FORCE_GENERIC) return false;
    return dtype == UMI_F16 && !txb && R == 3 && S == 3 && stride == 1 && pad == 1 && Co >= 1 && Co <= 4 && groups_ok(Ci) &&
           Ci <= 256 && ldx % 8 == 0;
}
size_t umi_head3_wgrad_ws_bytes(long P, int Ci, int Co) {
    return (size_t)((P + WG_PPB - 1) / WG_PPB) * 9 * Ci * Co * sizeof(float);
}
int umi_head3_wgrad(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co, long s_ci, long s_t,
                    float out_scale, int N, int H, int W, int Ci, int Co, void* ws, size_t ws_bytes, hipStream_t s) {
    const long P = (long)N * H * W;
    if (ws_bytes < umi_head3_wgrad_ws_bytes(P, Ci, Co)) return UMI_ERR_WORKSPACE;
    if (!al16(x)) return UMI_ERR_BADARG;
    const int blocks = (int)((P + WG_PPB - 1) / WG_PPB);
#define GO(NC) hipLaunchKernelGGL(head3x3_wgrad_kernel<NC>, dim3(blocks, 9), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)txa, (const half_t*)dy, lddy, (float*)ws, N, H, W, Ci)
    switch (Co) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; default: GO(4); }
#undef GO
    UMI_LAUNCH_CHECK();
    umi_launch_wgrad_reduce((const float*)ws, blocks, 9, Ci, Co, dW, s_co, s_ci, s_t, out_scale, s);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
