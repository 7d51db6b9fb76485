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


// ---- cubic resize: scipy.ndimage.zoom(img, (oh / H, ow / W[, 1]), order=3) as `preprocess` calls it (test_mc3serousv5.py:100-113)
// SciPy's algorithm (ndimage/_interpolation.py zoom; ni_splines.c, ni_interpolation.c NI_ZoomShift), restated by
// oracle/ref_resize.py and pinned against SciPy itself there:
//   1. cubic B-spline prefilter in float64 along H, then W: gain 6, pole sqrt(3) - 2, MIRROR initialisation (mode 'constant'
//      filters as 'mirror');
//   2. corner-aligned sampling x = i * (in - 1) / (out - 1), 4 x 4 coefficients around floor(x) - 1, mirrored indices;
//   3. uint8 images: floor(v + 0.5) clipped to [0, 255] (the output has the input's type).
// All arithmetic in float64 without contraction (SciPy's generic x86-64 build has no FMA): uint8 results are identical to SciPy's,
// float32 ones to the last bit of the float64 -> float32 rounding except where summation order differs by an ulp of float64.
#pragma clang fp contract(off)
constexpr double ZC_POLE = -0.26794919243112270647;       // sqrt(3) - 2

// one thread = one line (a column of one channel for AXIS 0, a row of one channel for AXIS 1) of the float64 coefficient image
// c[H][W][C]; AXIS 0 also converts the source image into it
template <typename T, int AXIS>
__global__ __launch_bounds__(256) void zoom_prefilter_kernel(const T* __restrict__ img, double* __restrict__ c, int H, int W, int C) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long lines = AXIS == 0 ? (long)W * C : (long)H * C;
    if (t >= lines) return;
    const int n = AXIS == 0 ? H : W;
    const long stride = AXIS == 0 ? (long)W * C : C;
    const long base = AXIS == 0 ? t : (t / C) * (long)W * C + (t % C);
    double* p = c + base;
    const double z = ZC_POLE;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    if (AXIS == 0)
        for (int i = 0; i < n; ++i) p[i * stride] = (double)img[base + i * stride];
    if (n < 2) return;
    for (int i = 0; i < n; ++i) p[i * stride] *= gain;
    const double z_n_1 = pow(z, (double)(n - 1));
    double c0 = p[0] + z_n_1 * p[(n - 1) * stride];
    double z_i = z;
    for (int i = 1; i < n - 1; ++i) {
        c0 = c0 + z_i * (p[i * stride] + z_n_1 * p[(n - 1 - i) * stride]);
        z_i *= z;
    }
    p[0] = c0 / (1.0 - z_n_1 * z_n_1);
    for (int i = 1; i < n; ++i) p[i * stride] += z * p[(i - 1) * stride];
    p[(n - 1) * stride] = (z * p[(n - 2) * stride] + p[(n - 1) * stride]) * z / (z * z - 1.0);
    for (int i = n - 2; i >= 0; --i) p[i * stride] = z * (p[(i + 1) * stride] - p[i * stride]);
}

__device__ inline int zc_mirror(int idx, int n) {
    if (n <= 1) return 0;
    const int s2 = 2 * n - 2;
    idx = (idx < 0 ? -idx : idx) % s2;
    return idx >= n ? s2 - idx : idx;
}
__device__ inline void zc_plan(int i, int n_in, int n_out, int idx[4], double w[4]) {
    const double x = (double)i * (n_out > 1 ? (double)(n_in - 1) / (double)(n_out - 1) : 0.0);
    const double f = floor(x), t = x - f, z = 1.0 - t;
    w[1] = (t * t * (t - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = z * z * z / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
    for (int k = 0; k < 4; ++k) idx[k] = zc_mirror((int)f - 1 + k, n_in);
}

template <typename T>
__global__ __launch_bounds__(256) void zoom_sample_kernel(const double* __restrict__ c, T* __restrict__ out, int H, int W, int C,
                                                          int oh, int ow) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)oh * ow * C) return;
    const int ch = (int)(t % C), ox = (int)((t / C) % ow), oy = (int)(t / ((long)C * ow));
    int iy[4], ix[4];
    double wy[4], wx[4];
    zc_plan(oy, H, oh, iy, wy);
    zc_plan(ox, W, ow, ix, wx);
    double v = 0.0;
    for (int ky = 0; ky < 4; ++ky) {
        double acc = 0.0;
        for (int kx = 0; kx < 4; ++kx) acc += wx[kx] * c[((long)iy[ky] * W + ix[kx]) * C + ch];
        v += wy[ky] * acc;
    }
    if (sizeof(T) == 1) {
        v = floor(v + 0.5);
        v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
        out[t] = (T)(int)v;
    } else out[t] = (T)v;
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

// scipy.ndimage.zoom(order=3) of one HWC image (C <= 4; src_dtype 0 = uint8, 1 = float32; the result has the same type), see the
// kernels above.  ws: umi_zoom_cubic_ws_bytes(H, W, C) bytes (the float64 coefficient image).
extern "C" size_t umi_zoom_cubic_ws_bytes(int H, int W, int C) { return (size_t)H * W * C * sizeof(double); }

extern "C" int umi_zoom_cubic_hwc(const void* img, int src_dtype, void* out, int H, int W, int C, int out_h, int out_w, void* ws,
                                  size_t ws_bytes, umi_stream_t stream) {
    if (!img || !out || !ws || H <= 0 || W <= 0 || C <= 0 || C > ZN_MAXC || out_h <= 0 || out_w <= 0) return UMI_ERR_BADARG;
    if (src_dtype != 0 && src_dtype != 1) return UMI_ERR_BADARG;
    if (ws_bytes < umi_zoom_cubic_ws_bytes(H, W, C)) return UMI_ERR_WORKSPACE;
    if ((long)H * W * C >= (1L << 31) || (long)out_h * out_w * C >= (1L << 31)) return UMI_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    double* c = (double*)ws;
    const unsigned g0 = (unsigned)(((long)W * C + 255) / 256), g1 = (unsigned)(((long)H * C + 255) / 256);
    const unsigned gs = (unsigned)(((long)out_h * out_w * C + 255) / 256);
    if (src_dtype == 0) {
        hipLaunchKernelGGL((zoom_prefilter_kernel<unsigned char, 0>), dim3(g0), dim3(256), 0, s, (const unsigned char*)img, c, H, W, C);
        hipLaunchKernelGGL((zoom_prefilter_kernel<unsigned char, 1>), dim3(g1), dim3(256), 0, s, (const unsigned char*)img, c, H, W, C);
        hipLaunchKernelGGL((zoom_sample_kernel<unsigned char>), dim3(gs), dim3(256), 0, s, c, (unsigned char*)out, H, W, C, out_h, out_w);
    } else {
        hipLaunchKernelGGL((zoom_prefilter_kernel<float, 0>), dim3(g0), dim3(256), 0, s, (const float*)img, c, H, W, C);
        hipLaunchKernelGGL((zoom_prefilter_kernel<float, 1>), dim3(g1), dim3(256), 0, s, (const float*)img, c, H, W, C);
        hipLaunchKernelGGL((zoom_sample_kernel<float>), dim3(gs), dim3(256), 0, s, c, (float*)out, H, W, C, out_h, out_w);
    }
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
