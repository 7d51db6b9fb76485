// HBM-bound fp16 kernels, vectorised to 16 B per lane (8 channels): BatchNorm+ReLU backward (reduce / apply)
// and MaxPool2d(2) forward / backward.  Taken when C % 8 == 0, pixel strides % 8 == 0, 16-B aligned bases
// and (C/8) divides 256; anything else stays on the scalar generic kernels.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

// pixel rows per stage-1 block of the BN-backward reduction: sized so that every layer launches ~2,048 workgroups
// (a fixed 512 left the 32x32 / 64x64 layers with 32-128 workgroups on 256 CUs)
static int rpb_for(long M) {
    long r = (M + 2047) / 2048;
    r = (r + 63) / 64 * 64;
    return (int)(r < 64 ? 64 : (r > 2048 ? 2048 : r));
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- BN + ReLU backward, stage 1: per-block partial sums of dz and dz*xhat ------------------------------
__global__ __launch_bounds__(256) void bn_bwd_reduce1_v8(const half_t* __restrict__ da, int ldda,
                                                         const half_t* __restrict__ y, int ldy,
                                                         const float4* __restrict__ tx, const float* __restrict__ rstd,
                                                         float* __restrict__ ws, long M, int C, int RPB) {
    __shared__ float red[2][256][9];               // [which][thread][8 channels] (+1 pad: conflict-free column sums)
    const int tid = threadIdx.x;
    const int G = C >> 3;                          // channel groups of 8; G divides 256
    const int PL = 256 / G;
    const int cg = tid % G, pl = tid / G;
    const long r0 = (long)blockIdx.x * RPB;
    long r1 = r0 + RPB;
    if (r1 > M) r1 = M;
    float4 t[8];
    float rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { t[j] = tx[cg * 8 + j]; rs[j] = rstd[cg * 8 + j]; }
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
#define UMI_BNB_ACC(yv_, gv_)                                          \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                    \
        float yy = (float)yv_[j];                                      \
        float z = umi_tx_pre(yy, t[j]);                                \
        float dz = z > t[j].w ? (float)gv_[j] : 0.f;                   \
        s[j] += dz;                                                    \
        q[j] = fmaf(dz, (yy - t[j].x) * rs[j], q[j]);                  \
    }
    long r = r0 + pl;
    for (; r + PL < r1; r += 2 * PL) {             // two rows per trip: four 16-B loads in flight per thread
        half8 yv0 = *reinterpret_cast<const half8*>(y + r * ldy + cg * 8);
        half8 gv0 = *reinterpret_cast<const half8*>(da + r * ldda + cg * 8);
        half8 yv1 = *reinterpret_cast<const half8*>(y + (r + PL) * ldy + cg * 8);
        half8 gv1 = *reinterpret_cast<const half8*>(da + (r + PL) * ldda + cg * 8);
        UMI_BNB_ACC(yv0, gv0)
        UMI_BNB_ACC(yv1, gv1)
    }
    if (r < r1) {
        half8 yv = *reinterpret_cast<const half8*>(y + r * ldy + cg * 8);
        half8 gv = *reinterpret_cast<const half8*>(da + r * ldda + cg * 8);
        UMI_BNB_ACC(yv, gv)
    }
#undef UMI_BNB_ACC
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][tid][j] = s[j]; red[1][tid][j] = q[j]; }
    __syncthreads();
    // thread (which, c) sums the PL pixel lanes of its channel in fixed order
    for (int i = tid; i < 2 * C; i += 256) {
        int which = i / C, c = i - which * C;
        int g = c >> 3, j = c & 7;
        float a = 0.f;
        for (int k = 0; k < PL; ++k) a += red[which][k * G + g][j];
        ws[((long)blockIdx.x * 2 + which) * C + c] = a;
    }
}

// ---- per-channel column sums (bias gradients: ConvTranspose2d / Linear biases), stage 1 -------------------------------
// SQ: also the per-channel sums of squares (slot 1 of the partial rows): BatchNorm statistics of a tensor whose producer has
// no statistics epilogue (the pointwise MFMA convolution of the attention gates)
template <bool SQ = false>
__device__ __forceinline__ void colsum_v8_body(const half_t* __restrict__ x, int ldx, float* __restrict__ ws, long M, int C,
                                               int RPB) {
    __shared__ float red[256][9];
    float q0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int tid = threadIdx.x;
    // channel groups of 8: this workgroup covers groups [gb, gb + Gb) (any C % 8 == 0: blockIdx.y walks 256 groups at a time,
    // a group count that does not divide 256 leaves the tail threads idle)
    const int Gt = C >> 3, gb = blockIdx.y * 256;
    const int Gb = Gt - gb < 256 ? Gt - gb : 256;
    const int PL = 256 / Gb;
    const int cgl = tid % Gb, pl = tid / Gb, cg = gb + cgl;
    const long r0 = (long)blockIdx.x * RPB;
    long r1 = r0 + RPB;
    if (r1 > M) r1 = M;
    float s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s0[j] = s1[j] = 0.f;
    if (pl < PL) {
        long r = r0 + pl;
        for (; r + 3 * PL < r1; r += 4 * PL) {         // four independent 16-B loads in flight per thread
            half8 a = *reinterpret_cast<const half8*>(x + r * ldx + cg * 8);
            half8 b = *reinterpret_cast<const half8*>(x + (r + PL) * ldx + cg * 8);
            half8 c = *reinterpret_cast<const half8*>(x + (r + 2 * PL) * ldx + cg * 8);
            half8 d = *reinterpret_cast<const half8*>(x + (r + 3 * PL) * ldx + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s0[j] += (float)a[j] + (float)c[j]; s1[j] += (float)b[j] + (float)d[j];
                if (SQ) {
                    const float fa = (float)a[j], fb = (float)b[j], fc = (float)c[j], fd = (float)d[j];
                    q0[j] = fmaf(fa, fa, fmaf(fb, fb, fmaf(fc, fc, fmaf(fd, fd, q0[j]))));
                }
            }
        }
        for (; r < r1; r += PL) {
            half8 a = *reinterpret_cast<const half8*>(x + r * ldx + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0[j] += (float)a[j]; if (SQ) q0[j] = fmaf((float)a[j], (float)a[j], q0[j]); }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[tid][j] = s0[j] + s1[j];
    __syncthreads();
    for (int cl = tid; cl < Gb * 8; cl += 256) {
        float a = 0.f;
        for (int k = 0; k < PL; ++k) a += red[k * Gb + (cl >> 3)][cl & 7];
        ws[((long)blockIdx.x * 2 + 0) * C + gb * 8 + cl] = a;
    }
    if (SQ) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid][j] = q0[j];
        __syncthreads();
        for (int cl = tid; cl < Gb * 8; cl += 256) {
            float a = 0.f;
            for (int k = 0; k < PL; ++k) a += red[k * Gb + (cl >> 3)][cl & 7];
            ws[((long)blockIdx.x * 2 + 1) * C + gb * 8 + cl] = a;
        }
    }
}

template <bool SQ = false>
__global__ __launch_bounds__(256) void colsum_v8(const half_t* __restrict__ x, int ldx, float* __restrict__ ws, long M, int C,
                                                 int RPB) {
    colsum_v8_body<SQ>(x, ldx, ws, M, C, RPB);
}
// up to 16 tensors of one shape per launch (blockIdx.z): the bias gradients of the twelve encoder layers of a ViT
struct CsGroup { const half_t* x[16]; float* out[16]; };
__global__ __launch_bounds__(256) void colsum_group_v8(CsGroup grp, int ldx, float* __restrict__ ws, long ws_stride, long M, int C,
                                                       int RPB) {
    colsum_v8_body<false>(grp.x[blockIdx.z], ldx, ws + blockIdx.z * ws_stride, M, C, RPB);
}
// second stage of the grouped launch: out[g][c] = scale * sum over the partial rows (slot 0) of group g, fp64, fixed order.
// Thread = channel (coalesced across the row), 4 row lanes per channel combined through LDS: the rows are few (tens), a
// workgroup per channel left 256 threads with a handful of loads each (94 us per launch for 12 x 3,072 channels).
__global__ __launch_bounds__(256) void colsum_group_reduce(CsGroup grp, const float* __restrict__ ws, long ws_stride, int rows, int C,
                                                           float scale) {
    __shared__ double sh[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const float* w = ws + blockIdx.y * ws_stride;
    double a = 0.0;
    if (c < C)
        for (int r = rl; r < rows; r += 4) a += (double)w[((long)r * 2 + 0) * C + c];
    sh[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && c < C) grp.out[blockIdx.y][c] = (float)((sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]) * (double)scale);
}

// ---- BN + ReLU backward, stage 3: dy = gamma*rstd * (dz - sum_dz/M - xhat*sum_dzx/M), in place ---------------
__global__ __launch_bounds__(256) void bn_bwd_apply_v8(half_t* __restrict__ da, int ldda, const half_t* __restrict__ y,
                                                       int ldy, const float4* __restrict__ tx,
                                                       const float* __restrict__ rstd,
                                                       const float* __restrict__ sum_dz,
                                                       const float* __restrict__ sum_dzx, long M, int C) {
    const int G = C >> 3;
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;     // gridDim*256 is a multiple of G => cg fixed per thread
    const int cg = (int)(gt % G);
    const long stride_rows = ((long)gridDim.x * 256) / G;
    const float invM = 1.f / (float)M;
    float4 t[8];
    float rs[8], c1[8], c2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        int c = cg * 8 + j;
        t[j] = tx[c];
        rs[j] = rstd[c];
        c1[j] = sum_dz[c] * invM;
        c2[j] = sum_dzx[c] * invM;
    }
    for (long r = gt / G; r < M; r += stride_rows) {
        half8 yv = *reinterpret_cast<const half8*>(y + r * ldy + cg * 8);
        half8 gv = *reinterpret_cast<const half8*>(da + r * ldda + cg * 8);
        *reinterpret_cast<half8*>(da + r * ldda + cg * 8) = umi_bn_dz8(yv, gv, t, rs, c1, c2);
    }
}

// ---- MaxPool2d(2) forward on the transformed tensor -------------------------------------------------------
template <bool HAS_TX>
__global__ __launch_bounds__(256) void pool2_fwd_v8(const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx,
                                                    half_t* __restrict__ y, int ldy, int N, int H, int W, int C) {
    const int G = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gt % G);
    const long stride_p = ((long)gridDim.x * 256) / G;
    const long P = (long)N * Ho * Wo;
    float4 t[8];
    if (HAS_TX) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = tx[cg * 8 + j];
    }
    for (long p = gt / G; p < P; p += stride_p) {
        int wo = (int)(p % Wo);
        long r = p / Wo;
        int ho = (int)(r % Ho);
        int n = (int)(r / Ho);
        const half_t* base = x + ((long)((long)n * H + 2 * ho) * W + 2 * wo) * ldx + cg * 8;
        half8 v[4];
        v[0] = *reinterpret_cast<const half8*>(base);
        v[1] = *reinterpret_cast<const half8*>(base + ldx);
        v[2] = *reinterpret_cast<const half8*>(base + (long)W * ldx);
        v[3] = *reinterpret_cast<const half8*>(base + (long)W * ldx + ldx);
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m = -INFINITY;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float f = (float)v[d][j];
                if (HAS_TX) f = umi_tx(f, t[j]);
                m = f > m ? f : m;
            }
            o[j] = (half_t)m;
        }
        *reinterpret_cast<half8*>(y + p * ldy + cg * 8) = o;
    }
}

// ---- MaxPool2d(2) backward: first arg-max of tx(x) gets dpool (even H and W only) -----------------------------
// RED: the pooled tensor is the output of a BatchNorm+ReLU layer (tx, rstd) and this is the LAST contribution to its
// gradient: the kernel then also emits stage 1 of that layer's BatchNorm backward reduction (per-workgroup sums of dz and
// dz*xhat over the values it stores, ws[block][2][C]) -- the separate reduction pass over da and x disappears.
template <bool HAS_TX, bool ACC, bool RED = false>
__global__ __launch_bounds__(256) void pool2_bwd_v8(const half_t* __restrict__ dp, int lddp, const half_t* __restrict__ x,
                                                    int ldx, const float4* __restrict__ tx, half_t* __restrict__ da,
                                                    int ldda, int N, int H, int W, int C,
                                                    const float* __restrict__ rstd = nullptr, float* __restrict__ ws = nullptr) {
    __shared__ float red[RED ? 2 : 1][RED ? 256 : 1][9];
    float rs[8], s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rs[j] = s[j] = q[j] = 0.f;
    const int G = C >> 3, Ho = H >> 1, Wo = W >> 1;
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gt % G);
    const long stride_p = ((long)gridDim.x * 256) / G;
    const long P = (long)N * Ho * Wo;
    float4 t[8];
    if (HAS_TX) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = tx[cg * 8 + j];
    }
    if (RED) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rs[j] = rstd[cg * 8 + j];
    }
    for (long p = gt / G; p < P; p += stride_p) {
        int wo = (int)(p % Wo);
        long r = p / Wo;
        int ho = (int)(r % Ho);
        int n = (int)(r / Ho);
        const long pix = ((long)((long)n * H + 2 * ho) * W + 2 * wo);
        const long off[4] = {0, 1, (long)W, (long)W + 1};
        half8 v[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) v[d] = *reinterpret_cast<const half8*>(x + (pix + off[d]) * ldx + cg * 8);
        half8 g = *reinterpret_cast<const half8*>(dp + p * lddp + cg * 8);
        half8 o[4];
        if (ACC) {
#pragma unroll
            for (int d = 0; d < 4; ++d) o[d] = *reinterpret_cast<const half8*>(da + (pix + off[d]) * ldda + cg * 8);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m = -INFINITY;
            int best = 0;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float f = (float)v[d][j];
                if (HAS_TX) f = umi_tx(f, t[j]);
                if (d == 0 || f > m) { m = f; best = d; }
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float add = (d == best) ? (float)g[j] : 0.f;
                o[d][j] = ACC ? (half_t)((float)o[d][j] + add) : (half_t)add;
                if (RED) {                      // sums over the STORED (fp16) gradient, like the apply pass will read it
                    const float yy = (float)v[d][j];
                    const float dz = umi_tx_pre(yy, t[j]) > t[j].w ? (float)o[d][j] : 0.f;
                    s[j] += dz;
                    q[j] = fmaf(dz, (yy - t[j].x) * rs[j], q[j]);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) *reinterpret_cast<half8*>(da + (pix + off[d]) * ldda + cg * 8) = o[d];
    }
    if (RED) {
        const int tid = threadIdx.x, PL = 256 / G;         // 256 % G == 0: this thread's group is tid % G (as gt % G)
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[0][tid][j] = s[j]; red[1][tid][j] = q[j]; }
        __syncthreads();
        for (int i = tid; i < 2 * C; i += 256) {
            const int which = i / C, c = i - which * C;
            float a = 0.f;
            for (int k = 0; k < PL; ++k) a += red[which][k * G + (c >> 3)][c & 7];
            ws[((long)blockIdx.x * 2 + which) * C + c] = a;
        }
    }
}

bool vec_ok(int C, int lda, int ldb, const void* a, const void* b) {
    if (C % 8 || lda % 8 || ldb % 8) return false;
    const int G = C / 8;
    if (G > 256 || 256 % G) return false;
    return al16(a) && al16(b);
}

int grid_for(long items) {
    long g = (items + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

int umi_bn_bwd_rpb_f16v(long M) { return rpb_for(M); }

// rows of partial sums the vectorised column-sum writes (layout [rows][2][C], slot 0 used), or 0 when it does not apply
// rows per column-sum workgroup: ~1,024 workgroups, at least two rows per pixel lane
static int rpb_colsum(long M, int C) {
    const int G = C / 8 < 256 ? C / 8 : 256, PL = 256 / G;
    long r = (M + 1023) / 1024;
    if (r < 2 * PL) r = 2 * PL;
    r = (r + PL - 1) / PL * PL;
    return (int)(r > 4096 ? 4096 : r);
}
int umi_colsum_rows_f16v(long M, int C) { return (C % 8 == 0) ? (int)((M + rpb_colsum(M, C) - 1) / rpb_colsum(M, C)) : 0; }
bool umi_colsum_f16v(const void* x, int ldx, float* ws, long M, int C, hipStream_t s) {
    if (!umi_colsum_rows_f16v(M, C) || ldx % 8 || !al16(x)) return false;
    const int rpb = rpb_colsum(M, C);
    hipLaunchKernelGGL(colsum_v8<false>, dim3((unsigned)((M + rpb - 1) / rpb), (C / 8 + 255) / 256), dim3(256), 0, s, (const half_t*)x, ldx, ws,
                       M, C, rpb);
    return true;
}
// n tensors of one shape: partial rows of tensor g at ws + g * rows * 2 * C, then outs[g] <- scale * column sums
bool umi_colsum_group_f16v(int n, const void* const* xs, int ldx, float* const* outs, float scale, float* ws, long M, int C,
                           hipStream_t s) {
    if (!umi_colsum_rows_f16v(M, C) || ldx % 8) return false;
    for (int i = 0; i < n; ++i)
        if (!al16(xs[i])) return false;
    // ~1,024 workgroups per LAUNCH: the tensors of a group share them, so each gets fewer, longer row blocks than alone
    // (12 tensors x 941 rows of partial sums made the second stage 94 us)
    int rpb = rpb_colsum(M, C);
    {
        const int cnt0 = n < 16 ? n : 16, G = C / 8 < 256 ? C / 8 : 256, PL = 256 / G;
        long r = (M * cnt0 * ((C / 8 + 255) / 256) + 1023) / 1024;
        r = (r + PL - 1) / PL * PL;
        if (r > rpb) rpb = (int)(r > 4096 ? 4096 : r);
    }
    const int rows = (int)((M + rpb - 1) / rpb);
    const long stride = (long)rows * 2 * C;
    for (int g0 = 0; g0 < n; g0 += 16) {
        const int cnt = n - g0 < 16 ? n - g0 : 16;
        CsGroup grp;
        for (int i = 0; i < 16; ++i) { const int j = g0 + (i < cnt ? i : 0); grp.x[i] = (const half_t*)xs[j]; grp.out[i] = outs[j]; }
        hipLaunchKernelGGL(colsum_group_v8, dim3((unsigned)rows, (C / 8 + 255) / 256, cnt), dim3(256), 0, s, grp, ldx, ws, stride, M, C, rpb);
        hipLaunchKernelGGL(colsum_group_reduce, dim3((C + 63) / 64, cnt), dim3(256), 0, s, grp, (const float*)ws, stride, rows, C, scale);
    }
    return true;
}
// BatchNorm statistics pass: part[rows][2][C] = per-block sums and sums of squares of x (rows = umi_colsum_rows_f16v)
bool umi_bn_stats_f16v(const void* x, int ldx, float* part, long M, int C, hipStream_t s) {
    if (!umi_colsum_rows_f16v(M, C) || ldx % 8 || !al16(x)) return false;
    const int rpb = rpb_colsum(M, C);
    hipLaunchKernelGGL(colsum_v8<true>, dim3((unsigned)((M + rpb - 1) / rpb), (C / 8 + 255) / 256), dim3(256), 0, s, (const half_t*)x, ldx,
                       part, M, C, rpb);
    return true;
}

bool umi_bn_bwd_reduce1_f16v(const void* da, int ldda, const void* y, int ldy, const void* tx, const float* rstd, float* ws,
                             long M, int C, hipStream_t s) {
    if (!vec_ok(C, ldda, ldy, da, y)) return false;
    const int rpb = rpb_for(M);
    int rows = (int)((M + rpb - 1) / rpb);
    hipLaunchKernelGGL(bn_bwd_reduce1_v8, dim3(rows), dim3(256), 0, s, (const half_t*)da, ldda, (const half_t*)y, ldy,
                       (const float4*)tx, rstd, ws, M, C, rpb);
    return true;
}

bool umi_bn_bwd_apply_f16v(void* da, int ldda, const void* y, int ldy, const void* tx, const float* rstd,
                           const float* sum_dz, const float* sum_dzx, long M, int C, hipStream_t s) {
    if (!vec_ok(C, ldda, ldy, da, y)) return false;
    hipLaunchKernelGGL(bn_bwd_apply_v8, dim3(grid_for(M * (C / 8))), dim3(256), 0, s, (half_t*)da, ldda, (const half_t*)y, ldy,
                       (const float4*)tx, rstd, sum_dz, sum_dzx, M, C);
    return true;
}

bool umi_pool2_fwd_f16v(const void* x, int ldx, const void* tx, void* y, int ldy, int N, int H, int W, int C,
                        hipStream_t s) {
    if (!vec_ok(C, ldx, ldy, x, y)) return false;
    int grid = grid_for((long)N * (H / 2) * (W / 2) * (C / 8));
    if (tx) hipLaunchKernelGGL(pool2_fwd_v8<true>, dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)tx, (half_t*)y, ldy, N, H, W, C);
    else hipLaunchKernelGGL(pool2_fwd_v8<false>, dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)tx, (half_t*)y, ldy, N, H, W, C);
    return true;
}

// pool backward + BatchNorm-backward stage 1 of the pooled layer: rows of partial sums written, or 0 when it does not apply
int umi_pool2_bwd_bnred_rows(int N, int H, int W, int C) {
    if (((H | W) & 1) || C % 8 || C / 8 > 256 || 256 % (C / 8)) return 0;
    return grid_for((long)N * (H / 2) * (W / 2) * (C / 8));
}
bool umi_pool2_bwd_bnred_f16v(const void* dp, int lddp, const void* x, int ldx, const void* tx, const float* rstd, void* da,
                              int ldda, int accumulate, float* part, int N, int H, int W, int C, hipStream_t s) {
    if (!tx || !rstd || !part || !umi_pool2_bwd_bnred_rows(N, H, W, C)) return false;
    if (!vec_ok(C, ldx, ldda, x, da) || lddp % 8 || !al16(dp)) return false;
    const int grid = umi_pool2_bwd_bnred_rows(N, H, W, C);
    if (accumulate)
        hipLaunchKernelGGL((pool2_bwd_v8<true, true, true>), dim3(grid), dim3(256), 0, s, (const half_t*)dp, lddp, (const half_t*)x,
                           ldx, (const float4*)tx, (half_t*)da, ldda, N, H, W, C, rstd, part);
    else
        hipLaunchKernelGGL((pool2_bwd_v8<true, false, true>), dim3(grid), dim3(256), 0, s, (const half_t*)dp, lddp, (const half_t*)x,
                           ldx, (const float4*)tx, (half_t*)da, ldda, N, H, W, C, rstd, part);
    return true;
}

bool umi_pool2_bwd_f16v(const void* dp, int lddp, const void* x, int ldx, const void* tx, void* da, int ldda,
                        int accumulate, int N, int H, int W, int C, hipStream_t s) {
    if ((H | W) & 1) return false;                        // odd tails stay on the generic kernel
    if (!vec_ok(C, ldx, ldda, x, da) || lddp % 8 || !al16(dp)) return false;
    int grid = grid_for((long)N * (H / 2) * (W / 2) * (C / 8));
#define GO(T, A) hipLaunchKernelGGL((pool2_bwd_v8<T, A>), dim3(grid), dim3(256), 0, s, (const half_t*)dp, lddp, (const half_t*)x, ldx, (const float4*)tx, (half_t*)da, ldda, N, H, W, C, (const float*)nullptr, (float*)nullptr)
    if (tx) { if (accumulate) GO(true, true); else GO(true, false); }
    else    { if (accumulate) GO(false, true); else GO(false, false); }
#undef GO
    return true;
}
