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

// Consumer-side transform of a stored activation: max((x - sub) * scale + shift, lo).
// t = {sub, scale, shift, lo}.  `pre` is the value before the clamp (needed by the
// ReLU mask in backward kernels).
__device__ __forceinline__ float umi_tx_pre(float v, const float4 t) {
    return fmaf(v - t.x, t.y, t.z);
}
__device__ __forceinline__ float umi_tx(float v, const float4 t) {
    return fmaxf(umi_tx_pre(v, t), t.w);
}

template <typename T> __device__ __forceinline__ float umi_ld(const T* p) { return (float)(*p); }
template <typename T> __device__ __forceinline__ void umi_st(T* p, float v) { *p = (T)v; }

static inline int umi_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
