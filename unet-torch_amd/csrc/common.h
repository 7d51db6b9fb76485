// Shared device helpers for libunetmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/unetmi.h"

typedef _Float16 half_t;

#define UMI_LAUNCH_CHECK()                                    \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

// Consumer-side transform of a stored activation: max(fma(x, scale, shift), lo) -- ONE fp32 fma + max per
// element, evaluated identically by every kernel that needs the activation or its ReLU mask.
// t = {mean, scale, shift, lo}: for BatchNorm scale = gamma*rstd, shift = beta - mean*scale; `mean` is only
// read by the BatchNorm backward kernels (xhat).  `pre` is the value before the clamp (the ReLU mask).
__device__ __forceinline__ float umi_tx_pre(float v, const float4 t) {
    return fmaf(v, t.y, t.z);
}
__device__ __forceinline__ float umi_tx(float v, const float4 t) {
    return fmaxf(umi_tx_pre(v, t), t.w);
}

// Stage 3 of the BatchNorm(+ReLU) backward for one element: the gradient of the raw conv output from the gradient `g` of
// the activated value, dz = gamma*rstd * (relu'(z)*g - c1 - xhat*c2), c1 = sum_dz/M, c2 = sum(dz*xhat)/M.  One definition
// with every rounding spelled out, because the standalone passes (bn_bwd_apply_v8, bn_bwd_apply_kernel) and the
// weight-gradient kernel that does it while staging (wgrad3x3_ws_kernel<.., BNA>) must agree bit for bit and the compiler
// picks contractions -- and whether the final product is rounded to fp16 once (v_fma_mix*_f16) or twice -- per call site.
// umi_bn_dz_inner: everything but the final multiplication by gamma*rstd (= t.y).
__device__ __forceinline__ float umi_bn_dz_inner(float y, float g, const float4 t, float rstd, float c1, float c2) {
#pragma clang fp contract(off)
    const float z = umi_tx_pre(y, t);
    const float dz = z > t.w ? g : 0.f;
    const float xh = (y - t.x) * rstd;
    return fmaf(-xh, c2, dz - c1);
}
// round_f16(a0 * b0) | round_f16(a1 * b1) << 16, each product rounded ONCE from the exact value
__device__ __forceinline__ unsigned umi_mul2_f16(float a0, float b0, float a1, float b1) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0\n\tv_fma_mixhi_f16 %0, %3, %4, 0" : "=&v"(r) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
    return r;
}
__device__ __forceinline__ _Float16 umi_mul_f16(float a, float b) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return __builtin_bit_cast(_Float16, (unsigned short)(r & 0xffffu));
}
template <typename T>
__device__ __forceinline__ T umi_bn_dz(float y, float g, const float4 t, float rstd, float c1, float c2);
template <>
__device__ __forceinline__ _Float16 umi_bn_dz<_Float16>(float y, float g, const float4 t, float rstd, float c1, float c2) {
    return umi_mul_f16(t.y, umi_bn_dz_inner(y, g, t, rstd, c1, c2));
}
template <>
__device__ __forceinline__ float umi_bn_dz<float>(float y, float g, const float4 t, float rstd, float c1, float c2) {
    return t.y * umi_bn_dz_inner(y, g, t, rstd, c1, c2);
}

// Same transform on 8 fp16 channels at once for the MFMA kernels' staging: fp32 fma on the fp16 input rounded straight
// to fp16 (v_fma_mixlo/hi_f16), then the clamp as a packed fp16 max -- max(round(z), lo) == round(max(z, lo)) for
// lo in {0, -inf}, so the result is bit-identical to (half)umi_tx((float)x, t) at ~1.5 instead of ~3 VALU ops/element.
typedef _Float16 umi_half8 __attribute__((ext_vector_type(8)));
typedef unsigned umi_uint4v __attribute__((ext_vector_type(4)));
// 8 channels of umi_bn_dz<_Float16>: t / rstd / c1 / c2 are the 8 channels' constants
__device__ __forceinline__ umi_half8 umi_bn_dz8(umi_half8 y, umi_half8 g, const float4* t, const float* rstd, const float* c1,
                                                const float* c2) {
    umi_uint4v r;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        r[j] = umi_mul2_f16(t[2 * j].y, umi_bn_dz_inner((float)y[2 * j], (float)g[2 * j], t[2 * j], rstd[2 * j], c1[2 * j], c2[2 * j]),
                            t[2 * j + 1].y, umi_bn_dz_inner((float)y[2 * j + 1], (float)g[2 * j + 1], t[2 * j + 1], rstd[2 * j + 1],
                                                            c1[2 * j + 1], c2[2 * j + 1]));
    return __builtin_bit_cast(umi_half8, r);
}
__device__ __forceinline__ umi_half8 umi_tx8(umi_half8 v, const float4* t) {
#ifdef UMI_EXP_TX_PK
    // timing-only ablation (tools/exp_stamp_wgrad.py): packed fp16 fma with fp16 scale/shift -- NOT the shipped numerics
    umi_half8 sc, sh, lo_;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = (half_t)t[j].y; sh[j] = (half_t)t[j].z; lo_[j] = (half_t)t[j].w; }
    return __builtin_elementwise_max(v * sc + sh, lo_);
#endif
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 in = __builtin_bit_cast(u32x4, v), out;
    umi_half8 lo;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        unsigned d;
        // fp32 fma with the fp16 operand taken from the low / high half of the packed register, result rounded to
        // fp16 into the low / high half of d (plain VALU->VALU dependency: interlocked by hardware)
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(d)
            : "v"(in[p]), "v"(t[2 * p].y), "v"(t[2 * p].z), "v"(t[2 * p + 1].y), "v"(t[2 * p + 1].z));
        out[p] = d;
        lo[2 * p] = (half_t)t[2 * p].w;
        lo[2 * p + 1] = (half_t)t[2 * p + 1].w;
    }
    return __builtin_elementwise_max(__builtin_bit_cast(umi_half8, out), lo);
}

template <typename T> __device__ __forceinline__ float umi_ld(const T* p) { return (float)(*p); }
template <typename T> __device__ __forceinline__ void umi_st(T* p, float v) { *p = (T)v; }

static inline int umi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// i = q * d + r for a non-negative element index: 32-bit arithmetic whenever i fits (always, in practice) -- a 64-bit division
// expands to ~200 instructions on gfx950, more than the rest of an 8-channel elementwise body.
__device__ __forceinline__ void umi_divmod(long i, int d, long& q, int& r) {
    if ((unsigned long)i <= 0xFFFFFFFFul) {
        const unsigned ui = (unsigned)i, qq = ui / (unsigned)d;
        q = (long)qq;
        r = (int)(ui - qq * (unsigned)d);
    } else {
        q = i / d;
        r = (int)(i - q * d);
    }
}
