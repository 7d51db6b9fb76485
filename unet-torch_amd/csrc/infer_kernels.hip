// The steps either side of the network at inference time (reference test_mc3serousv5.py:100-127 `preprocess`,
// :877-887 softmax -> argmax -> uint8 mask).  Both are one-pass HBM-bound byte/float streams.
//
//   umi_znorm_hwc     HWC image (uint8 as cv2.imread returns it, or float32) -> per-channel z-normalised CHW fp32,
//                     optional channel reversal (BGR -> RGB); mean / population std in fp64 like numpy.
//   umi_argmax_mask   NCHW fp32 logits -> uint8 class mask; the first maximum wins (torch.argmax).  softmax is monotone,
//                     so the reference's softmax before the argmax is skipped.
#include "common.h"

namespace {

constexpr int ZN_BLOCKS = 512;
constexpr int ZN_MAXC = 4;

template <typename T>
__device__ inline double zn_load(const T* p, long i) { return (double)p[i]; }

__device__ inline double block_sum(double v, double* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;                                    // valid in thread 0
}

// pass 0: partial sums of x; pass 1: partial sums of (x - mean)^2 (two-pass variance, as numpy.std)
template <typename T, int PASS>
__global__ __launch_bounds__(256) void znorm_reduce_kernel(const T* __restrict__ img, long HW, int C,
                                                            const double* __restrict__ stats, double* __restrict__ part) {
    __shared__ double sh[4];
    double acc[ZN_MAXC] = {0.0, 0.0, 0.0, 0.0};
    double mean[ZN_MAXC] = {0.0, 0.0, 0.0, 0.0};
    if (PASS == 1)
        for (int c = 0; c < C; ++c) mean[c] = stats[c];
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256)
        for (int c = 0; c < C; ++c) {
            const double v = zn_load(img, p * C + c) - mean[c];
            acc[c] += PASS == 0 ? v : v * v;
        }
    for (int c = 0; c < C; ++c) {
        const double s = block_sum(acc[c], sh);
        if (threadIdx.x == 0) part[(long)blockIdx.x * ZN_MAXC + c] = s;
    }
}

// stats[c] = mean (PASS 0) / stats[ZN_MAXC + c] = population std (PASS 1): fixed-order sum of the block partials
template <int PASS>
__global__ __launch_bounds__(64) void znorm_finalize_kernel(const double* __restrict__ part, int nblk, long HW, int C,
                                                            double* __restrict__ stats) {
    const int c = threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += part[(long)b * ZN_MAXC + c];
    if (PASS == 0) stats[c] = s / (double)HW;
    else stats[ZN_MAXC + c] = sqrt(s / (double)HW);
}

template <typename T>
__global__ __launch_bounds__(256) void znorm_apply_kernel(const T* __restrict__ img, float* __restrict__ out, long HW, int C,
                                                          int reverse, const double* __restrict__ stats) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    for (int c = 0; c < C; ++c) {
        const double v = (zn_load(img, p * C + c) - stats[c]) / stats[ZN_MAXC + c];     // fp64, then one rounding to fp32
        out[(long)(reverse ? C - 1 - c : c) * HW + p] = (float)v;
    }
}

template <int C>
__global__ __launch_bounds__(256) void argmax_mask_kernel(const float* __restrict__ logits, unsigned char* __restrict__ mask,
                                                          long HW, int Cdyn) {
    const int n = blockIdx.y;
    const long p4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (p4 >= HW) return;
    const float* base = logits + (long)n * (C ? C : Cdyn) * HW;
    const int nc = C ? C : Cdyn;
    if (p4 + 4 <= HW && (HW & 3) == 0) {
        float4 best = *reinterpret_cast<const float4*>(base + p4);
        uchar4 bi = make_uchar4(0, 0, 0, 0);
        for (int c = 1; c < nc; ++c) {
            const float4 v = *reinterpret_cast<const float4*>(base + (long)c * HW + p4);
            if (v.x > best.x) { best.x = v.x; bi.x = (unsigned char)c; }
            if (v.y > best.y) { best.y = v.y; bi.y = (unsigned char)c; }
            if (v.z > best.z) { best.z = v.z; bi.z = (unsigned char)c; }
            if (v.w > best.w) { best.w = v.w; bi.w = (unsigned char)c; }
        }
        *reinterpret_cast<uchar4*>(mask + (long)n * HW + p4) = bi;
    } else {
        for (long p = p4; p < HW && p < p4 + 4; ++p) {
            float best = base[p];
            unsigned char bi = 0;
            for (int c = 1; c < nc; ++c) {
                const float v = base[(long)c * HW + p];
                if (v > best) { best = v; bi = (unsigned char)c; }
            }
            mask[(long)n * HW + p] = bi;
        }
    }
}

}  // namespace

extern "C" size_t umi_znorm_ws_bytes(void) { return (size_t)(ZN_BLOCKS + 2) * ZN_MAXC * sizeof(double); }

extern "C" int umi_znorm_hwc(const void* img, int src_dtype, float* out_chw, long HW, int C, int reverse_channels, void* ws,
                             size_t ws_bytes, umi_stream_t stream) {
    if (!img || !out_chw || HW <= 0 || C < 1 || C > ZN_MAXC || (src_dtype != 0 && src_dtype != 1)) return UMI_ERR_BADARG;
    if (!ws || ws_bytes < umi_znorm_ws_bytes()) return UMI_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    double* stats = (double*)ws;                          // [2][ZN_MAXC]: mean, std
    double* part = stats + 2 * ZN_MAXC;                   // [ZN_BLOCKS][ZN_MAXC]
    long want = (HW + 255) / 256;
    const int nblk = (int)(want < ZN_BLOCKS ? want : ZN_BLOCKS);
    const int ablk = (int)((HW + 255) / 256);
#define ZN_RUN(T)                                                                                                      \
    do {                                                                                                               \
        hipLaunchKernelGGL((znorm_reduce_kernel<T, 0>), dim3(nblk), dim3(256), 0, s, (const T*)img, HW, C, stats, part); \
        hipLaunchKernelGGL((znorm_finalize_kernel<0>), dim3(1), dim3(64), 0, s, part, nblk, HW, C, stats);             \
        hipLaunchKernelGGL((znorm_reduce_kernel<T, 1>), dim3(nblk), dim3(256), 0, s, (const T*)img, HW, C, stats, part); \
        hipLaunchKernelGGL((znorm_finalize_kernel<1>), dim3(1), dim3(64), 0, s, part, nblk, HW, C, stats);             \
        hipLaunchKernelGGL((znorm_apply_kernel<T>), dim3(ablk), dim3(256), 0, s, (const T*)img, out_chw, HW, C,        \
                           reverse_channels, stats);                                                                   \
    } while (0)
    if (src_dtype == 0) ZN_RUN(unsigned char); else ZN_RUN(float);
#undef ZN_RUN
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_argmax_mask(const float* logits, unsigned char* mask, int N, int C, long HW, umi_stream_t stream) {
    if (!logits || !mask || N <= 0 || C < 1 || C > 256 || HW <= 0) return UMI_ERR_BADARG;
    if (((unsigned long)logits & 15) || ((unsigned long)mask & 3)) return UMI_ERR_BADARG;
    dim3 grid((unsigned)((HW + 1023) / 1024), N);
    hipStream_t s = (hipStream_t)stream;
    if (C == 2) hipLaunchKernelGGL((argmax_mask_kernel<2>), grid, dim3(256), 0, s, logits, mask, HW, C);
    else if (C == 4) hipLaunchKernelGGL((argmax_mask_kernel<4>), grid, dim3(256), 0, s, logits, mask, HW, C);
    else hipLaunchKernelGGL((argmax_mask_kernel<0>), grid, dim3(256), 0, s, logits, mask, HW, C);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
