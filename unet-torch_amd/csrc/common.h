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

template <typename T> __device__ __forceinline__ float umi_ld(const T* p) { return (float)(*p); }
template <typename T> __device__ __forceinline__ void umi_st(T* p, float v) { *p = (T)v; }

static inline int umi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
