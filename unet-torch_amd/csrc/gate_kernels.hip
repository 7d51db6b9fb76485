// Additive attention gate of UNet_attention (reference Model.py:265-305, Attention_block):
//     E = relu(BN(Wq q_up) + BN(Wx x));   A = sigmoid(BN(psi E));   out = x * A
// The convolutions / BatchNorms run on the common conv + statistics kernels; this file holds the two fused elementwise
// steps and their backward.  All HBM-bound: one pass over the operands, consumer transforms applied on load.
//
//   umi_add2_relu_fwd / _bwd   y = max(tx_a(a) + tx_b(b), 0);   da = db = dy * [y > 0]
//   umi_gate_fwd / _bwd        y[m][c] = tx_x(x[m][c]) * sigmoid(tx_p(p[m]))      (p: one channel, broadcast over c)
//                              dx = dy * A;   dp[m] = A (1 - A) * sum_c dy[m][c] * tx_x(x[m][c])
//
// One wave owns a pixel at a time: lanes stride over the channels (coalesced), the per-pixel reduction of the backward
// is a wave shuffle.  Pixels are independent -> grid-stride over pixels, >> 256 workgroups at the sizes of interest.
#include "common.h"
#include <stdint.h>

namespace {

__device__ inline float txf(float v, const float4 t) { return fmaxf(fmaf(v, t.y, t.z), t.w); }
__device__ inline float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

template <typename T>
__global__ __launch_bounds__(256) void add2_relu_fwd_kernel(const T* __restrict__ a, int lda, const float4* __restrict__ txa,
                                                            const T* __restrict__ b, int ldb, const float4* __restrict__ txb,
                                                            T* __restrict__ y, int ldy, long M, int C) {
    const long total = M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        float va = (float)a[m * lda + c], vb = (float)b[m * ldb + c];
        if (txa) va = txf(va, txa[c]);
        if (txb) vb = txf(vb, txb[c]);
        y[m * ldy + c] = (T)fmaxf(va + vb, 0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add2_relu_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                                            T* __restrict__ da, int ldda, T* __restrict__ db, int lddb, long M,
                                                            int C) {
    const long total = M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        const T g = (float)y[m * ldy + c] > 0.f ? dy[m * lddy + c] : (T)0.f;
        da[m * ldda + c] = g;
        db[m * lddb + c] = g;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const T* __restrict__ x, int ldx, const float4* __restrict__ txx,
                                                       const T* __restrict__ p, const float4* __restrict__ txp,
                                                       T* __restrict__ y, int ldy, long M, int C) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * 256) >> 6;
    const float4 tp = txp ? txp[0] : make_float4(0.f, 1.f, 0.f, -INFINITY);
    for (long m = wave; m < M; m += nwaves) {
        const float A = sigmoidf_(txf((float)p[m], tp));
        for (int c = lane; c < C; c += 64) {
            float v = (float)x[m * ldx + c];
            if (txx) v = txf(v, txx[c]);
            y[m * ldy + c] = (T)(v * A);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                       const float4* __restrict__ txx, const T* __restrict__ p,
                                                       const float4* __restrict__ txp, T* __restrict__ dx, int lddx,
                                                       T* __restrict__ dp, long M, int C) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((long)gridDim.x * 256) >> 6;
    const float4 tp = txp ? txp[0] : make_float4(0.f, 1.f, 0.f, -INFINITY);
    for (long m = wave; m < M; m += nwaves) {
        const float A = sigmoidf_(txf((float)p[m], tp));
        float acc = 0.f;
        for (int c = lane; c < C; c += 64) {
            const float g = (float)dy[m * lddy + c];
            float v = (float)x[m * ldx + c];
            if (txx) v = txf(v, txx[c]);
            acc = fmaf(g, v, acc);
            dx[m * lddx + c] = (T)(g * A);
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) dp[m] = (T)(acc * A * (1.f - A));
    }
}

// 16-B vectorised variants (fp16, C % 8 == 0, C/8 a power of two <= 64): a pixel's channels sit on C/8 consecutive lanes,
// 64 / (C/8) pixels per wave; the backward's per-pixel sum is a butterfly over those lanes.
typedef _Float16 half8g __attribute__((ext_vector_type(8)));

template <bool HAS_TX>
__global__ __launch_bounds__(256) void gate_fwd_v8(const half_t* __restrict__ x, int ldx, const float4* __restrict__ txx,
                                                   const half_t* __restrict__ p, const float4* __restrict__ txp,
                                                   half_t* __restrict__ y, int ldy, long M, int G) {
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gt % G);
    const long stride = ((long)gridDim.x * 256) / G;
    const float4 tp = txp ? txp[0] : make_float4(0.f, 1.f, 0.f, -INFINITY);
    float4 t[8];
    if (HAS_TX) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = txx[cg * 8 + j];
    }
    for (long m = gt / G; m < M; m += stride) {
        const float A = sigmoidf_(txf((float)p[m], tp));
        const half8g v = *reinterpret_cast<const half8g*>(x + m * ldx + cg * 8);
        half8g o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)v[j];
            if (HAS_TX) f = txf(f, t[j]);
            o[j] = (half_t)(f * A);
        }
        *reinterpret_cast<half8g*>(y + m * ldy + cg * 8) = o;
    }
}

template <bool HAS_TX>
__global__ __launch_bounds__(256) void gate_bwd_v8(const half_t* __restrict__ dy, int lddy, const half_t* __restrict__ x, int ldx,
                                                   const float4* __restrict__ txx, const half_t* __restrict__ p,
                                                   const float4* __restrict__ txp, half_t* __restrict__ dx, int lddx,
                                                   half_t* __restrict__ dp, long M, int G) {
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gt % G);
    const long stride = ((long)gridDim.x * 256) / G;
    const float4 tp = txp ? txp[0] : make_float4(0.f, 1.f, 0.f, -INFINITY);
    float4 t[8];
    if (HAS_TX) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = txx[cg * 8 + j];
    }
    // every lane of a pixel's group runs the same number of trips (M and the stride are the same for all of them), so the
    // butterfly below always finds its partners
    for (long m = gt / G; m < M; m += stride) {
        const float A = sigmoidf_(txf((float)p[m], tp));
        const half8g g = *reinterpret_cast<const half8g*>(dy + m * lddy + cg * 8);
        const half8g v = *reinterpret_cast<const half8g*>(x + m * ldx + cg * 8);
        half8g o;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)v[j];
            if (HAS_TX) f = txf(f, t[j]);
            acc = fmaf((float)g[j], f, acc);
            o[j] = (half_t)((float)g[j] * A);
        }
        *reinterpret_cast<half8g*>(dx + m * lddx + cg * 8) = o;
        for (int w = G >> 1; w > 0; w >>= 1) acc += __shfl_xor(acc, w, 64);
        if (cg == 0) dp[m] = (half_t)(acc * A * (1.f - A));
    }
}

// add2_relu, 16 B per lane (same arithmetic as the scalar kernels: fp32 transform, one rounding of the result)
__global__ __launch_bounds__(256) void add2_relu_fwd_v8(const half_t* __restrict__ a, int lda, const float4* __restrict__ txa,
                                                        const half_t* __restrict__ b, int ldb, const float4* __restrict__ txb,
                                                        half_t* __restrict__ y, int ldy, long M, int G) {
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gt % G);
    const long stride = ((long)gridDim.x * 256) / G;
    const float4 ident = make_float4(0.f, 1.f, 0.f, -INFINITY);
    float4 ta[8], tb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        ta[j] = txa ? txa[cg * 8 + j] : ident;
        tb[j] = txb ? txb[cg * 8 + j] : ident;
    }
    for (long m = gt / G; m < M; m += stride) {
        const half8g va = *reinterpret_cast<const half8g*>(a + m * lda + cg * 8);
        const half8g vb = *reinterpret_cast<const half8g*>(b + m * ldb + cg * 8);
        half8g o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float fa = (float)va[j], fb = (float)vb[j];
            if (txa) fa = txf(fa, ta[j]);
            if (txb) fb = txf(fb, tb[j]);
            o[j] = (half_t)fmaxf(fa + fb, 0.f);
        }
        *reinterpret_cast<half8g*>(y + m * ldy + cg * 8) = o;
    }
}

__global__ __launch_bounds__(256) void add2_relu_bwd_v8(const half_t* __restrict__ dy, int lddy, const half_t* __restrict__ y,
                                                        int ldy, half_t* __restrict__ da, int ldda, half_t* __restrict__ db,
                                                        int lddb, long M, int G) {
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gt % G);
    const long stride = ((long)gridDim.x * 256) / G;
    for (long m = gt / G; m < M; m += stride) {
        const half8g g = *reinterpret_cast<const half8g*>(dy + m * lddy + cg * 8);
        const half8g v = *reinterpret_cast<const half8g*>(y + m * ldy + cg * 8);
        half8g o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (float)v[j] > 0.f ? g[j] : (half_t)0.f;
        *reinterpret_cast<half8g*>(da + m * ldda + cg * 8) = o;
        *reinterpret_cast<half8g*>(db + m * lddb + cg * 8) = o;
    }
}

inline bool gate_vec_ok(int C, int l0, int l1, int l2, const void* a, const void* b, const void* c) {
    const int G = C / 8;
    if (C % 8 || G > 64 || (G & (G - 1)) || l0 % 8 || l1 % 8 || l2 % 8) return false;
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}

inline int ew_grid(long work_items, int per_block) {
    long g = (work_items + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

}  // namespace

extern "C" int umi_add2_relu_fwd(const void* a, int lda, const void* txa, const void* b, int ldb, const void* txb, void* y,
                                 int ldy, long M, int C, int dtype, umi_stream_t stream) {
    if (!a || !b || !y || M <= 0 || C <= 0 || lda < C || ldb < C || ldy < C) return UMI_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UMI_F16 && gate_vec_ok(C, lda, ldb, ldy, a, b, y)) {
        const int G = C / 8;
        hipLaunchKernelGGL(add2_relu_fwd_v8, dim3(ew_grid(M * G, 256 * 4)), dim3(256), 0, s, (const half_t*)a, lda,
                           (const float4*)txa, (const half_t*)b, ldb, (const float4*)txb, (half_t*)y, ldy, M, G);
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    const int grid = ew_grid(M * C, 256 * 4);
    if (dtype == UMI_F16)
        hipLaunchKernelGGL(add2_relu_fwd_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)a, lda, (const float4*)txa,
                           (const half_t*)b, ldb, (const float4*)txb, (half_t*)y, ldy, M, C);
    else if (dtype == UMI_F32)
        hipLaunchKernelGGL(add2_relu_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)a, lda, (const float4*)txa,
                           (const float*)b, ldb, (const float4*)txb, (float*)y, ldy, M, C);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_add2_relu_bwd(const void* dy, int lddy, const void* y, int ldy, void* da, int ldda, void* db, int lddb,
                                 long M, int C, int dtype, umi_stream_t stream) {
    if (!dy || !y || !da || !db || M <= 0 || C <= 0 || lddy < C || ldy < C || ldda < C || lddb < C) return UMI_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UMI_F16 && gate_vec_ok(C, lddy, ldy, ldda, dy, y, da) && lddb % 8 == 0 && ((uintptr_t)db & 15) == 0) {
        const int G = C / 8;
        hipLaunchKernelGGL(add2_relu_bwd_v8, dim3(ew_grid(M * G, 256 * 4)), dim3(256), 0, s, (const half_t*)dy, lddy,
                           (const half_t*)y, ldy, (half_t*)da, ldda, (half_t*)db, lddb, M, G);
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    const int grid = ew_grid(M * C, 256 * 4);
    if (dtype == UMI_F16)
        hipLaunchKernelGGL(add2_relu_bwd_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)dy, lddy, (const half_t*)y,
                           ldy, (half_t*)da, ldda, (half_t*)db, lddb, M, C);
    else if (dtype == UMI_F32)
        hipLaunchKernelGGL(add2_relu_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, lddy, (const float*)y,
                           ldy, (float*)da, ldda, (float*)db, lddb, M, C);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_gate_fwd(const void* x, int ldx, const void* txx, const void* p, const void* txp, void* y, int ldy, long M,
                            int C, int dtype, umi_stream_t stream) {
    if (!x || !p || !y || M <= 0 || C <= 0 || ldx < C || ldy < C) return UMI_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UMI_F16 && gate_vec_ok(C, ldx, ldy, 8, x, y, x)) {
        const int G = C / 8, g2 = ew_grid(M * G, 256 * 4);
        if (txx) hipLaunchKernelGGL(gate_fwd_v8<true>, dim3(g2), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)txx,
                                    (const half_t*)p, (const float4*)txp, (half_t*)y, ldy, M, G);
        else hipLaunchKernelGGL(gate_fwd_v8<false>, dim3(g2), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)txx,
                                (const half_t*)p, (const float4*)txp, (half_t*)y, ldy, M, G);
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    const int grid = ew_grid(M, 4 * 4);                  // 4 waves per block, ~4 pixels per wave
    if (dtype == UMI_F16)
        hipLaunchKernelGGL(gate_fwd_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)txx,
                           (const half_t*)p, (const float4*)txp, (half_t*)y, ldy, M, C);
    else if (dtype == UMI_F32)
        hipLaunchKernelGGL(gate_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, (const float4*)txx,
                           (const float*)p, (const float4*)txp, (float*)y, ldy, M, C);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_gate_bwd(const void* dy, int lddy, const void* x, int ldx, const void* txx, const void* p, const void* txp,
                            void* dx, int lddx, void* dp, long M, int C, int dtype, umi_stream_t stream) {
    if (!dy || !x || !p || !dx || !dp || M <= 0 || C <= 0 || lddy < C || ldx < C || lddx < C) return UMI_ERR_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UMI_F16 && gate_vec_ok(C, lddy, ldx, lddx, dy, x, dx)) {
        const int G = C / 8, g2 = ew_grid(M * G, 256 * 4);
        if (txx) hipLaunchKernelGGL(gate_bwd_v8<true>, dim3(g2), dim3(256), 0, s, (const half_t*)dy, lddy, (const half_t*)x, ldx,
                                    (const float4*)txx, (const half_t*)p, (const float4*)txp, (half_t*)dx, lddx, (half_t*)dp, M, G);
        else hipLaunchKernelGGL(gate_bwd_v8<false>, dim3(g2), dim3(256), 0, s, (const half_t*)dy, lddy, (const half_t*)x, ldx,
                                (const float4*)txx, (const half_t*)p, (const float4*)txp, (half_t*)dx, lddx, (half_t*)dp, M, G);
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    const int grid = ew_grid(M, 4 * 4);
    if (dtype == UMI_F16)
        hipLaunchKernelGGL(gate_bwd_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)dy, lddy, (const half_t*)x, ldx,
                           (const float4*)txx, (const half_t*)p, (const float4*)txp, (half_t*)dx, lddx, (half_t*)dp, M, C);
    else if (dtype == UMI_F32)
        hipLaunchKernelGGL(gate_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, lddy, (const float*)x, ldx,
                           (const float4*)txx, (const float*)p, (const float4*)txp, (float*)dx, lddx, (float*)dp, M, C);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
