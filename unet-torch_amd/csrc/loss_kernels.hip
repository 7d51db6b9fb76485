// Fused 'dice_bce_mc' training loss (reference loss.py:488-500 with DiceLoss loss.py:215-251) on NCHW fp32 logits:
//   loss = 0.5 * CrossEntropy(logits, target) + 0.5 * (1/C) * sum_c (1 - (2*A_c + eps) / (B_c + T_c + eps)),
//   p = softmax(logits), A_c = sum p_c*[t==c], B_c = sum p_c^2, T_c = sum [t==c]   (sums over batch and pixels), eps = 1e-5.
// Forward: one streaming pass (softmax in registers) -> per-block partial rows -> fp64 fixed-order finalize (deterministic).
// Backward: one streaming pass that recomputes the softmax and writes d loss / d logits.
// HBM-bound: 4*C bytes read per pixel per pass (+ the target), 4*C written by the backward.
#include "common.h"

namespace {

constexpr float DICE_EPS = 1e-5f;

__device__ __forceinline__ int load_target(const void* t, int tdtype, long i) {
    if (tdtype == 0) return (int)reinterpret_cast<const long long*>(t)[i];
    if (tdtype == 1) return (int)reinterpret_cast<const float*>(t)[i];
    if (tdtype == 2) return (int)reinterpret_cast<const unsigned char*>(t)[i];
    return reinterpret_cast<const int*>(t)[i];
}

// part row layout: [A_0..A_{NC-1}, B_0.., T_0.., ce_sum]
template <int NC>
__global__ __launch_bounds__(256) void dice_ce_stats_kernel(const float* __restrict__ logits, const void* __restrict__ target,
                                                            int tdtype, long HW, long total, float* __restrict__ part) {
    __shared__ float red[4][3 * NC + 1];
    float a[NC], b[NC], tt[NC], ce = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) a[c] = b[c] = tt[c] = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, hw = i - n * HW;
        const float* x = logits + n * NC * HW + hw;
        float v[NC], m = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c) { v[c] = x[(long)c * HW]; m = fmaxf(m, v[c]); }
        const int t = load_target(target, tdtype, i);
        float s = 0.f, dt = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float d = v[c] - m;
            if (c == t) dt = d;
            v[c] = expf(d);
            s += v[c];
        }
        const float inv = 1.f / s;
        ce += logf(s) - dt;                             // -log_softmax[t], the stable form F.cross_entropy uses
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float p = v[c] * inv;
            b[c] = fmaf(p, p, b[c]);
            if (c == t) { a[c] += p; tt[c] += 1.f; }
        }
    }
    // wave reduction, then the 4 waves through LDS in fixed order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a[c] += __shfl_xor(a[c], o);
            b[c] += __shfl_xor(b[c], o);
            tt[c] += __shfl_xor(tt[c], o);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ce += __shfl_xor(ce, o);
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) { red[wave][c] = a[c]; red[wave][NC + c] = b[c]; red[wave][2 * NC + c] = tt[c]; }
        red[wave][3 * NC] = ce;
    }
    __syncthreads();
    if (threadIdx.x < 3 * NC + 1)
        part[(long)blockIdx.x * (3 * NC + 1) + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// stats out: [A | B | T | ce_sum | loss]  (3*NC + 2 floats).  1,024 threads = 32 columns x 32 row lanes; lane l sums rows
// l, l+32, ... in fp64, then the 32 lane sums are added in lane order (deterministic).
__global__ __launch_bounds__(1024) void dice_ce_finalize_kernel(const float* __restrict__ part, int rows, int NC, long total,
                                                                float* __restrict__ stats) {
    __shared__ double lanes[32][33];
    __shared__ double sums[32];
    const int K = 3 * NC + 1;
    const int col = threadIdx.x & 31, lane = threadIdx.x >> 5;
    double s = 0.0;
    if (col < K)
        for (int r = lane; r < rows; r += 32) s += (double)part[(long)r * K + col];
    lanes[lane][col] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
        for (int l = 0; l < 32; ++l) t += lanes[l][threadIdx.x];
        sums[threadIdx.x] = t;
        if ((int)threadIdx.x < K) stats[threadIdx.x] = (float)t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double dice = 0.0;
        for (int c = 0; c < NC; ++c)
            dice += 1.0 - (2.0 * sums[c] + (double)DICE_EPS) / (sums[NC + c] + sums[2 * NC + c] + (double)DICE_EPS);
        stats[K] = (float)(0.5 * sums[3 * NC] / (double)total + 0.5 * dice / NC);
    }
}

template <int NC>
__global__ __launch_bounds__(256) void dice_ce_bwd_kernel(const float* __restrict__ logits, const void* __restrict__ target,
                                                          int tdtype, const float* __restrict__ stats,
                                                          const float* __restrict__ gout, long HW, long total,
                                                          float* __restrict__ dlogits) {
    const float g = gout ? gout[0] : 1.f;
    float k1[NC], k2[NC];                              // d loss_dice / d p_c = k1_c * [t==c] + k2_c * p_c
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float D = stats[NC + c] + stats[2 * NC + c] + DICE_EPS;
        k1[c] = -0.5f / NC * 2.f / D;
        k2[c] = 0.5f / NC * (2.f * stats[c] + DICE_EPS) * 2.f / (D * D);
    }
    const float ce_w = 0.5f / (float)total;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long n = i / HW, hw = i - n * HW;
        const float* x = logits + n * NC * HW + hw;
        float p[NC], m = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; ++c) { p[c] = x[(long)c * HW]; m = fmaxf(m, p[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) { p[c] = expf(p[c] - m); s += p[c]; }
        const float inv = 1.f / s;
        const int t = load_target(target, tdtype, i);
        float h[NC], dot = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            p[c] *= inv;
            h[c] = (c == t ? k1[c] : 0.f) + k2[c] * p[c];
            dot = fmaf(h[c], p[c], dot);
        }
        float* o = dlogits + n * NC * HW + hw;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            o[(long)c * HW] = g * (ce_w * (p[c] - (c == t ? 1.f : 0.f)) + p[c] * (h[c] - dot));
    }
}

int blocks_for(long total) {
    long b = (total + 255) / 256 / 4;                   // ~4 pixels per thread
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace

extern "C" size_t umi_dice_ce_ws_bytes(int N, int C, long HW) {
    return (size_t)blocks_for((long)N * HW) * (3 * C + 1) * sizeof(float);
}

extern "C" int umi_dice_ce_fwd(const float* logits, const void* target, int target_dtype, int N, int C, long HW, float* stats,
                               void* ws, size_t ws_bytes, umi_stream_t st) {
    if (!logits || !target || !stats || !ws || N <= 0 || HW <= 0 || target_dtype < 0 || target_dtype > 3) return UMI_ERR_BADARG;
    if (C < 1 || C > 8) return UMI_ERR_UNSUPPORTED;
    if (ws_bytes < umi_dice_ce_ws_bytes(N, C, HW)) return UMI_ERR_WORKSPACE;
    const long total = (long)N * HW;
    const int rows = blocks_for(total);
    hipStream_t s = (hipStream_t)st;
#define GO(NC) hipLaunchKernelGGL(dice_ce_stats_kernel<NC>, dim3(rows), dim3(256), 0, s, logits, target, target_dtype, HW, total, (float*)ws)
    switch (C) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
                 case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break; default: GO(8); }
#undef GO
    UMI_LAUNCH_CHECK();
    hipLaunchKernelGGL(dice_ce_finalize_kernel, dim3(1), dim3(1024), 0, s, (const float*)ws, rows, C, total, stats);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_dice_ce_bwd(const float* logits, const void* target, int target_dtype, const float* stats, const float* gout,
                               int N, int C, long HW, float* dlogits, umi_stream_t st) {
    if (!logits || !target || !stats || !dlogits || N <= 0 || HW <= 0 || target_dtype < 0 || target_dtype > 3) return UMI_ERR_BADARG;
    if (C < 1 || C > 8) return UMI_ERR_UNSUPPORTED;
    const long total = (long)N * HW;
    const int grid = blocks_for(total);
    hipStream_t s = (hipStream_t)st;
#define GO(NC) hipLaunchKernelGGL(dice_ce_bwd_kernel<NC>, dim3(grid), dim3(256), 0, s, logits, target, target_dtype, stats, gout, HW, total, dlogits)
    switch (C) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
                 case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break; default: GO(8); }
#undef GO
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
