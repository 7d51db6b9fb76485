// 16-byte vectorised fp16 variants of the TransUNet elementwise / normalisation kernels (thread = 8 channels of one row).
// Arithmetic is the same fp32 expression per element as the scalar kernels in transformer_kernels.hip (the dropout
// stream is keyed by the same element index), so either path gives the same values; these exist because the scalar
// `i % C, i / C` form reaches only ~20 % of HBM bandwidth.  Every launcher returns false when the shape or alignment
// does not qualify and the caller falls back to the scalar kernel.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

namespace {

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
inline int grid8(long vecs) {
    long g = (vecs + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float u) {
    return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
}
__device__ __forceinline__ unsigned hash32(unsigned a, unsigned b) {
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void ew8_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ g, int ldg,
                                                  half_t* __restrict__ y, int ldy, long M, int C8, long bcast_rows) {
    const long total = M * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long r; int c;
        umi_divmod(i, C8, r, c);
        c *= 8;
        half8 xv = *reinterpret_cast<const half8*>(x + r * ldx + c), gv, o;
        if (MODE == 1 || MODE == 2) gv = *reinterpret_cast<const half8*>(g + r * ldg + c);
        if (MODE == 3) gv = *reinterpret_cast<const half8*>(g + (r % bcast_rows) * ldg + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)xv[j], v;
            if (MODE == 0) v = gelu_f(f);
            else if (MODE == 1) v = (float)gv[j] * gelu_df(f);
            else v = f + (float)gv[j];
            o[j] = (half_t)v;
        }
        *reinterpret_cast<half8*>(y + r * ldy + c) = o;
    }
}

__global__ __launch_bounds__(256) void dropout8_kernel(const half_t* __restrict__ x, int ldx, half_t* __restrict__ y, int ldy,
                                                       unsigned char* __restrict__ mask, int bwd, float p, unsigned seed,
                                                       long M, int C8, const float4* __restrict__ tx,
                                                       const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B9u;          // same offset rule as the scalar kernel
    const long total = M * C8;
    const float scale = 1.f / (1.f - p);
    const int C = C8 * 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long r; int c;
        umi_divmod(i, C8, r, c);
        c *= 8;
        const long e0 = r * C + c;                                       // element index of the scalar kernel
        half8 xv = *reinterpret_cast<const half8*>(x + r * ldx + c), o;
        unsigned long long mk = 0;
        if (bwd) mk = *reinterpret_cast<const unsigned long long*>(mask + e0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned k;
            if (bwd) k = (unsigned)(mk >> (8 * j)) & 0xFFu;
            else {
                const long e = e0 + j;
                float u = (hash32((unsigned)e, seed ^ (unsigned)(e >> 32)) >> 8) * (1.f / 16777216.f);
                k = u >= p;
                mk |= (unsigned long long)k << (8 * j);
            }
            float v = (float)xv[j];
            if (tx && !bwd) v = umi_tx(v, tx[c + j]);
            o[j] = (half_t)(k ? v * scale : 0.f);
        }
        if (!bwd) *reinterpret_cast<unsigned long long*>(mask + e0) = mk;
        *reinterpret_cast<half8*>(y + r * ldy + c) = o;
    }
}

// Dropout fused with its neighbours in a transformer block (vit_seg_modeling.py:113-119,177-187):
//   forward : y = dropout(GELU ? gelu(x) : x) + (AUX ? aux : 0)       (MLP: fc1 -> GELU -> dropout;  residual: dropout(f(x)) + h)
//   backward: y = dropout'(x) * (GELU ? gelu'(aux) : 1)                (aux = the forward's pre-activation)
// Same random stream (element index, seed) and mask bytes as dropout8_kernel; one rounding to fp16 instead of two or three.
template <bool GELU, bool AUX>
__global__ __launch_bounds__(256) void dropout8_fused_kernel(const half_t* __restrict__ x, int ldx, half_t* __restrict__ y, int ldy,
                                                             unsigned char* __restrict__ mask, int bwd, float p, unsigned seed,
                                                             long M, int C8, const unsigned* __restrict__ seed_dev,
                                                             const half_t* __restrict__ aux, int ldaux) {
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B9u;
    const long total = M * C8;
    const float scale = 1.f / (1.f - p);
    const int C = C8 * 8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long r; int c;
        umi_divmod(i, C8, r, c);
        c *= 8;
        const long e0 = r * C + c;
        half8 xv = *reinterpret_cast<const half8*>(x + r * ldx + c), av, o;
        if (AUX || (GELU && bwd)) av = *reinterpret_cast<const half8*>(aux + r * ldaux + c);
        unsigned long long mk = 0;
        if (bwd) mk = *reinterpret_cast<const unsigned long long*>(mask + e0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned k;
            if (bwd) k = (unsigned)(mk >> (8 * j)) & 0xFFu;
            else {
                const long e = e0 + j;
                float u = (hash32((unsigned)e, seed ^ (unsigned)(e >> 32)) >> 8) * (1.f / 16777216.f);
                k = u >= p;
                mk |= (unsigned long long)k << (8 * j);
            }
            float v = (float)xv[j];
            if (!bwd) {
                if (GELU) v = gelu_f(v);
                v = k ? v * scale : 0.f;
                if (AUX) v += (float)av[j];
            } else {
                v = k ? v * scale : 0.f;
                if (GELU) v *= gelu_df((float)av[j]);
            }
            o[j] = (half_t)v;
        }
        if (!bwd) *reinterpret_cast<unsigned long long*>(mask + e0) = mk;
        *reinterpret_cast<half8*>(y + r * ldy + c) = o;
    }
}

// LayerNorm backward: one wave per row, lane owns K4 groups of 4 channels (C <= 4*64*K4; C = 768 -> K4 = 3); dgamma/dbeta
// stay in registers across the rows of a wave and leave as one partial row per workgroup (fixed order => deterministic)
constexpr int LNV_ROWS = 16;       // rows per backward workgroup (4 per wave)
template <int K4>
__global__ __launch_bounds__(256) void ln_bwd_v4_kernel(const half_t* __restrict__ dy, int lddy, const half_t* __restrict__ x,
                                                        int ldx, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, half_t* __restrict__ dx, int lddx,
                                                        float* __restrict__ part, long M, int C) {
    __shared__ float red[3][2][K4 * 256];                       // waves 1..3 -> wave 0
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float gm[K4][4], dg[K4][4], db[K4][4];
#pragma unroll
    for (int k = 0; k < K4; ++k) {
        const int c = (lane + 64 * k) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) { gm[k][j] = c < C ? gamma[c + j] : 0.f; dg[k][j] = 0.f; db[k][j] = 0.f; }
    }
    const float invC = 1.f / C;
    const long r0 = (long)blockIdx.x * LNV_ROWS;
    for (int rr = wave; rr < LNV_ROWS; rr += 4) {
        const long row = r0 + rr;
        if (row >= M) break;
        const float m = mean[row], r = rstd[row];
        float d[K4][4], xh[K4][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < K4; ++k) {
            const int c = (lane + 64 * k) * 4;
            half4 hd = {0, 0, 0, 0}, hx = {0, 0, 0, 0};
            if (c < C) {
                hd = *reinterpret_cast<const half4*>(dy + row * lddy + c);
                hx = *reinterpret_cast<const half4*>(x + row * ldx + c);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[k][j] = (float)hd[j];
                xh[k][j] = c < C ? ((float)hx[j] - m) * r : 0.f;
                const float g = d[k][j] * gm[k][j];
                s1 += g;
                s2 = fmaf(g, xh[k][j], s2);
            }
        }
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
#pragma unroll
        for (int k = 0; k < K4; ++k) {
            const int c = (lane + 64 * k) * 4;
            half4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = (half_t)(r * (d[k][j] * gm[k][j] - s1 - xh[k][j] * s2));
                dg[k][j] = fmaf(d[k][j], xh[k][j], dg[k][j]);
                db[k][j] += d[k][j];
            }
            if (c < C) *reinterpret_cast<half4*>(dx + row * lddx + c) = o;
        }
    }
    if (wave) {
#pragma unroll
        for (int k = 0; k < K4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[wave - 1][0][(k * 4 + j) * 64 + lane] = dg[k][j];
                red[wave - 1][1][(k * 4 + j) * 64 + lane] = db[k][j];
            }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int k = 0; k < K4; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c >= C) continue;
            float ap[4], bp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sa = dg[k][j], sb = db[k][j];
#pragma unroll
                for (int w = 0; w < 3; ++w) { sa += red[w][0][(k * 4 + j) * 64 + lane]; sb += red[w][1][(k * 4 + j) * 64 + lane]; }
                ap[j] = sa;
                bp[j] = sb;
            }
            *reinterpret_cast<float4*>(part + ((long)blockIdx.x * 2 + 0) * C + c) = make_float4(ap[0], ap[1], ap[2], ap[3]);
            *reinterpret_cast<float4*>(part + ((long)blockIdx.x * 2 + 1) * C + c) = make_float4(bp[0], bp[1], bp[2], bp[3]);
        }
    }
}

// MaxPool2d(3, stride 2, pad 0), 8 channels per thread.  The forward also stores WHICH of the 9 taps won (first maximum in scan
// order, the rule of the scalar kernels and of torch's max_pool2d backward); the backward then reads one byte per (window,
// channel) instead of re-deriving every window's argmax from 9 input taps (36 two-byte loads per input element: 325 us for the
// R50 root's 24 x 112 x 112 x 64 tensor).
__global__ __launch_bounds__(256) void pool3s2_fwd8_kernel(const half_t* __restrict__ x, int ldx, half_t* __restrict__ y, int ldy,
                                                           unsigned char* __restrict__ idx, int N, int H, int W, int C8, int Ho,
                                                           int Wo) {
    const long total = (long)N * Ho * Wo * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        const long p = i / C8;
        const int wo = (int)(p % Wo);
        const long r = p / Wo;
        const int ho = (int)(r % Ho), n = (int)(r / Ho);
        float m[8];
        unsigned char b[8];
#pragma unroll
        for (int d = 0; d < 9; ++d) {
            const half8 v = *reinterpret_cast<const half8*>(x + ((long)((long)n * H + 2 * ho + d / 3) * W + 2 * wo + d % 3) * ldx + c);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (d == 0 || (float)v[j] > m[j]) { m[j] = (float)v[j]; b[j] = (unsigned char)d; }
        }
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)m[j];
        *reinterpret_cast<half8*>(y + p * ldy + c) = o;
        if (idx) {
            uint2 pk;
            pk.x = b[0] | (b[1] << 8) | (b[2] << 16) | ((unsigned)b[3] << 24);
            pk.y = b[4] | (b[5] << 8) | (b[6] << 16) | ((unsigned)b[7] << 24);
            *reinterpret_cast<uint2*>(idx + p * (C8 * 8) + c) = pk;
        }
    }
}

__global__ __launch_bounds__(256) void pool3s2_bwd8_kernel(const half_t* __restrict__ dy, int lddy, const unsigned char* __restrict__ idx,
                                                           half_t* __restrict__ dx, int lddx, int N, int H, int W, int C8, int Ho,
                                                           int Wo) {
    const long total = (long)N * H * W * C8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        const long p = i / C8;
        const int w = (int)(p % W);
        const long r = p / W;
        const int h = (int)(r % H), n = (int)(r / H);
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        // windows (ho, wo) with 2*ho <= h <= 2*ho + 2: ascending, the accumulation order of the scalar kernel
        for (int ho = h < 2 ? 0 : (h - 1) / 2; ho <= h / 2 && ho < Ho; ++ho)
            for (int wo = w < 2 ? 0 : (w - 1) / 2; wo <= w / 2 && wo < Wo; ++wo) {
                const long q = ((long)n * Ho + ho) * Wo + wo;
                const uint2 pk = *reinterpret_cast<const uint2*>(idx + q * (C8 * 8) + c);
                const half8 g = *reinterpret_cast<const half8*>(dy + q * lddy + c);
                const unsigned me = (unsigned)((h - 2 * ho) * 3 + (w - 2 * wo));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned b = ((j < 4 ? pk.x : pk.y) >> (8 * (j & 3))) & 0xffu;
                    if (b == me) acc[j] += (float)g[j];
                }
            }
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
        *reinterpret_cast<half8*>(dx + (long)p * lddx + c) = o;
    }
}

}  // namespace

bool umi_ew_f16v(int mode, const void* x, int ldx, const void* g, int ldg, void* y, int ldy, long M, int C, long bcast_rows,
                 hipStream_t s) {
    if (C % 8 || ldx % 8 || ldy % 8 || (mode && ldg % 8) || !al16(x) || !al16(y) || (mode && !al16(g))) return false;
    const int C8 = C / 8;
    const int grid = grid8(M * C8);
    const half_t* xp = (const half_t*)x;
    const half_t* gp = (const half_t*)g;
    half_t* yp = (half_t*)y;
    if (mode == 0) hipLaunchKernelGGL(ew8_kernel<0>, dim3(grid), dim3(256), 0, s, xp, ldx, gp, ldg, yp, ldy, M, C8, bcast_rows);
    else if (mode == 1) hipLaunchKernelGGL(ew8_kernel<1>, dim3(grid), dim3(256), 0, s, xp, ldx, gp, ldg, yp, ldy, M, C8, bcast_rows);
    else if (mode == 2) hipLaunchKernelGGL(ew8_kernel<2>, dim3(grid), dim3(256), 0, s, xp, ldx, gp, ldg, yp, ldy, M, C8, bcast_rows);
    else hipLaunchKernelGGL(ew8_kernel<3>, dim3(grid), dim3(256), 0, s, xp, ldx, gp, ldg, yp, ldy, M, C8, bcast_rows);
    return true;
}

bool umi_dropout_f16v(const void* x, int ldx, void* y, int ldy, void* mask, int backward, float p, unsigned seed, long M, int C,
                      const void* tx, const unsigned* seed_dev, hipStream_t s) {
    if (C % 8 || ldx % 8 || ldy % 8 || !al16(x) || !al16(y) || (((uintptr_t)mask) & 7)) return false;
    const int C8 = C / 8;
    hipLaunchKernelGGL(dropout8_kernel, dim3(grid8(M * C8)), dim3(256), 0, s, (const half_t*)x, ldx, (half_t*)y, ldy,
                       (unsigned char*)mask, backward, p, seed, M, C8, (const float4*)tx, seed_dev);
    return true;
}

bool umi_dropout_fused_f16v(const void* x, int ldx, void* y, int ldy, void* mask, int backward, float p, unsigned seed, long M,
                            int C, const unsigned* seed_dev, const void* aux, int ldaux, int gelu, hipStream_t s) {
    const bool need_aux = backward ? gelu != 0 : aux != nullptr;
    if (C % 8 || ldx % 8 || ldy % 8 || !al16(x) || !al16(y) || (((uintptr_t)mask) & 7) || (need_aux && (!aux || ldaux % 8 || !al16(aux))))
        return false;
    const int C8 = C / 8;
    const dim3 grid(grid8(M * C8));
#define GO(G_, A_)                                                                                                    \
    hipLaunchKernelGGL((dropout8_fused_kernel<G_, A_>), grid, dim3(256), 0, s, (const half_t*)x, ldx, (half_t*)y, ldy,  \
                       (unsigned char*)mask, backward, p, seed, M, C8, seed_dev, (const half_t*)aux, ldaux)
    if (gelu) { if (!backward && aux) GO(true, true); else GO(true, false); }
    else { if (!backward && aux) GO(false, true); else GO(false, false); }
#undef GO
    return true;
}

int umi_ln_bwd_rows_f16v() { return LNV_ROWS; }

// part rows = ceil(M / LNV_ROWS), layout [rows][2][C] as the scalar kernel's
bool umi_ln_bwd_f16v(const void* dy, int lddy, const void* x, int ldx, const float* gamma, const float* mean, const float* rstd,
                     void* dx, int lddx, float* part, long M, int C, hipStream_t s) {
    if (C % 4 || C > 1024 || lddy % 4 || ldx % 4 || lddx % 4 || (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx) & 7) ||
        (((uintptr_t)part) & 15))
        return false;
    const int rows = (int)((M + LNV_ROWS - 1) / LNV_ROWS);
    const int K4 = (C / 4 + 63) / 64;
#define GO(K)                                                                                                            \
    hipLaunchKernelGGL(ln_bwd_v4_kernel<K>, dim3(rows), dim3(256), 0, s, (const half_t*)dy, lddy, (const half_t*)x, ldx, gamma, \
                       mean, rstd, (half_t*)dx, lddx, part, M, C)
    if (K4 == 1) GO(1); else if (K4 == 2) GO(2); else if (K4 == 3) GO(3); else GO(4);
#undef GO
    return true;
}

// ---- UpsamplingBilinear2d(x2, align_corners=True) forward / adjoint, 8 channels per thread (same per-element float
// expressions as the scalar kernels in transformer_kernels.hip) ------------------------------------------------------------
namespace {

// A thread keeps ONE 8-channel group for all its pixels (256 % C8 == 0, as in pool2_fwd_v8): its transform rows are loaded once,
// so a pixel costs four loads and a store instead of twelve loads and a store (the kernel was bound by vector-memory issue:
// 1.2 TB/s of output); the pixel index is split with 32-bit divisions (IDX = unsigned wherever the pixel count allows).
template <typename IDX>
__global__ __launch_bounds__(256) void bilinear2x_fwd8_kernel(const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx,
                                                              half_t* __restrict__ y, int ldy, int N, int H, int W, int C8) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const IDX gt = (IDX)blockIdx.x * 256 + threadIdx.x;
    const int c = (int)(gt % (IDX)C8) * 8;
    const IDX stride_p = ((IDX)gridDim.x * 256) / (IDX)C8, P = (IDX)N * Ho * Wo;
    float4 t[8];
    if (tx) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = tx[c + j];
    }
    for (IDX p = gt / (IDX)C8; p < P; p += stride_p) {
        const int wo = (int)(p % (IDX)Wo);
        const IDX r = p / (IDX)Wo;
        const int ho = (int)(r % (IDX)Ho), n = (int)(r / (IDX)Ho);
        const float fy = ho * sy, fx = wo * sx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
        const float ly = fy - y0, lx = fx - x0;
        const half_t* b = x + (long)n * H * W * ldx + c;
        const half8 a00 = *reinterpret_cast<const half8*>(b + ((long)y0 * W + x0) * ldx);
        const half8 a01 = *reinterpret_cast<const half8*>(b + ((long)y0 * W + x1) * ldx);
        const half8 a10 = *reinterpret_cast<const half8*>(b + ((long)y1 * W + x0) * ldx);
        const half8 a11 = *reinterpret_cast<const half8*>(b + ((long)y1 * W + x1) * ldx);
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v00 = (float)a00[j], v01 = (float)a01[j], v10 = (float)a10[j], v11 = (float)a11[j];
            if (tx) { v00 = umi_tx(v00, t[j]); v01 = umi_tx(v01, t[j]); v10 = umi_tx(v10, t[j]); v11 = umi_tx(v11, t[j]); }
            o[j] = (half_t)((1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11));
        }
        *reinterpret_cast<half8*>(y + (long)p * ldy + c) = o;
    }
}

template <typename IDX>
__global__ __launch_bounds__(256) void bilinear2x_bwd8_kernel(const half_t* __restrict__ dy, int lddy, half_t* __restrict__ dx,
                                                              int lddx, int N, int H, int W, int C8) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const IDX total = (IDX)N * H * W * C8;
    for (IDX i = (IDX)blockIdx.x * 256 + threadIdx.x; i < total; i += (IDX)gridDim.x * 256) {
        const int c = (int)(i % (IDX)C8) * 8;
        const IDX p = i / (IDX)C8;
        const int w = (int)(p % (IDX)W);
        const IDX r = p / (IDX)W;
        const int h = (int)(r % (IDX)H), n = (int)(r / (IDX)H);
        int ho_lo = sy > 0.f ? (int)floorf((h - 1) / sy) : 0, ho_hi = sy > 0.f ? (int)ceilf((h + 1) / sy) : Ho - 1;
        int wo_lo = sx > 0.f ? (int)floorf((w - 1) / sx) : 0, wo_hi = sx > 0.f ? (int)ceilf((w + 1) / sx) : Wo - 1;
        ho_lo = ho_lo < 0 ? 0 : ho_lo; wo_lo = wo_lo < 0 ? 0 : wo_lo;
        ho_hi = ho_hi > Ho - 1 ? Ho - 1 : ho_hi; wo_hi = wo_hi > Wo - 1 ? Wo - 1 : wo_hi;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            const float fy = ho * sy;
            const int y0 = (int)fy;
            const int y1 = y0 + 1 < H ? y0 + 1 : H - 1;
            const float ly = fy - y0;
            const float wy = (y0 == h ? 1.f - ly : 0.f) + (y1 == h ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const float fx = wo * sx;
                const int x0 = (int)fx;
                const int x1 = x0 + 1 < W ? x0 + 1 : W - 1;
                const float lx = fx - x0;
                const float wx = (x0 == w ? 1.f - lx : 0.f) + (x1 == w ? lx : 0.f);
                if (wx == 0.f) continue;
                const half8 g = *reinterpret_cast<const half8*>(dy + ((long)((long)n * Ho + ho) * Wo + wo) * lddy + c);
                const float ww = wy * wx;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(ww, (float)g[j], acc[j]);
            }
        }
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
        *reinterpret_cast<half8*>(dx + (long)p * lddx + c) = o;
    }
}

}  // namespace

bool umi_bilinear2x_f16v(const void* x, int ldx, const void* tx, void* y, int ldy, int backward, int N, int H, int W, int C,
                         hipStream_t s) {
    if (C % 8 || ldx % 8 || ldy % 8 || !al16(x) || !al16(y)) return false;
    const int C8 = C / 8;
    if (!backward && (C8 > 256 || 256 % C8)) return false;       // (forward: one channel group per thread; other widths take the scalar kernel)
    const bool small = (long)N * 4 * H * W * C8 + 16384L * 256 < 0xFFFFFFFFL;      // (the grid-stride loop's last increment must not wrap)
    if (!backward) {
        if (small)
            hipLaunchKernelGGL(bilinear2x_fwd8_kernel<unsigned>, dim3(grid8((long)N * 4 * H * W * C8)), dim3(256), 0, s, (const half_t*)x,
                               ldx, (const float4*)tx, (half_t*)y, ldy, N, H, W, C8);
        else
            hipLaunchKernelGGL(bilinear2x_fwd8_kernel<long>, dim3(grid8((long)N * 4 * H * W * C8)), dim3(256), 0, s, (const half_t*)x,
                               ldx, (const float4*)tx, (half_t*)y, ldy, N, H, W, C8);
    } else {
        if (small)
            hipLaunchKernelGGL(bilinear2x_bwd8_kernel<unsigned>, dim3(grid8((long)N * H * W * C8)), dim3(256), 0, s, (const half_t*)x, ldx,
                               (half_t*)y, ldy, N, H, W, C8);
        else
            hipLaunchKernelGGL(bilinear2x_bwd8_kernel<long>, dim3(grid8((long)N * H * W * C8)), dim3(256), 0, s, (const half_t*)x, ldx,
                               (half_t*)y, ldy, N, H, W, C8);
    }
    return true;
}

bool umi_pool3s2_fwd_f16v(const void* x, int ldx, void* y, int ldy, void* idx, int N, int H, int W, int C, hipStream_t s) {
    if (C % 8 || ldx % 8 || ldy % 8 || !al16(x) || !al16(y) || (idx && (((uintptr_t)idx) & 7))) return false;
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    hipLaunchKernelGGL(pool3s2_fwd8_kernel, dim3(grid8((long)N * Ho * Wo * (C / 8))), dim3(256), 0, s, (const half_t*)x, ldx,
                       (half_t*)y, ldy, (unsigned char*)idx, N, H, W, C / 8, Ho, Wo);
    return true;
}
bool umi_pool3s2_bwd_f16v(const void* dy, int lddy, const void* idx, void* dx, int lddx, int N, int H, int W, int C, hipStream_t s) {
    if (!idx || C % 8 || lddy % 8 || lddx % 8 || !al16(dy) || !al16(dx) || (((uintptr_t)idx) & 7)) return false;
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    hipLaunchKernelGGL(pool3s2_bwd8_kernel, dim3(grid8((long)N * H * W * (C / 8))), dim3(256), 0, s, (const half_t*)dy, lddy,
                       (const unsigned char*)idx, (half_t*)dx, lddx, N, H, W, C / 8, Ho, Wo);
    return true;
}
