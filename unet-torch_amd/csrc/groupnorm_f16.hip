// GroupNorm reductions for fp16 NHWC tensors with channel-coalesced 16-B accesses (thread = pixel lane x 8 channels).
// The generic kernels in transformer_kernels.hip walk one (sample, group) per workgroup with a pixel-strided access
// pattern; here every (sample, row-split) workgroup streams whole pixels and produces per-channel partial sums, from
// which the per-group statistics follow in a tiny second stage (fp64 combine, fixed order => deterministic).
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

// pixel rows per stage-1 workgroup: ~1,024 workgroups per launch (a fixed 512 left the 14x14 / 28x28 stages of the
// ResNet with one workgroup per sample: 24 of 256 CUs busy)
static int gn_rows(int N, long HW) {
    long r = ((long)N * HW + 1023) / 1024;
    r = (r + 7) / 8 * 8;
    return (int)(r < 16 ? 16 : (r > 512 ? 512 : r));
}

// MODE 0: out = (sum x, sum x^2) per channel.   MODE 1: out = (sum dz*xhat, sum dz) per channel, dz = dy*[y>0 | 1]
template <int MODE>
__global__ __launch_bounds__(256) void gn_rowsum_v8(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ dy,
                                                    int lddy, const half_t* __restrict__ y, int ldy,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd, int relu,
                                                    long HW, int C, int G, int S, float* __restrict__ ws, int ROWS) {
    __shared__ float red[2][256][9];
    const int tid = threadIdx.x;
    const int G8 = C >> 3, PL = 256 / G8;
    const int cg = tid % G8, pl = tid / G8;
    const int n = blockIdx.x, sp = blockIdx.y;
    const long r0 = (long)sp * ROWS;
    long r1 = r0 + ROWS;
    if (r1 > HW) r1 = HW;
    float a[8], b[8], mu[8], rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = b[j] = 0.f;
        if (MODE == 1) {
            int g = (cg * 8 + j) / (C / G);
            mu[j] = mean[n * G + g];
            rs[j] = rstd[n * G + g];
        }
    }
    for (long r = r0 + pl; r < r1; r += PL) {
        const long row = (long)n * HW + r;
        half8 xv = *reinterpret_cast<const half8*>(x + row * ldx + cg * 8);
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float f = (float)xv[j]; a[j] += f; b[j] = fmaf(f, f, b[j]); }
        } else {
            half8 gv = *reinterpret_cast<const half8*>(dy + row * lddy + cg * 8);
            half8 yv;
            if (relu) yv = *reinterpret_cast<const half8*>(y + row * ldy + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float dz = (float)gv[j];
                if (relu && !((float)yv[j] > 0.f)) dz = 0.f;
                a[j] = fmaf(dz, ((float)xv[j] - mu[j]) * rs[j], a[j]);
                b[j] += dz;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][tid][j] = a[j]; red[1][tid][j] = b[j]; }
    __syncthreads();
    for (int i = tid; i < 2 * C; i += 256) {
        int which = i / C, c = i - which * C;
        float s = 0.f;
        for (int k = 0; k < PL; ++k) s += red[which][k * G8 + (c >> 3)][c & 7];
        ws[(((long)n * S + sp) * 2 + which) * C + c] = s;
    }
}

// forward stage 2: one thread per (sample, group): fp64 combine over splits and the group's channels
__global__ void gn_stats_finalize(const float* __restrict__ ws, int N, int S, int C, int G, long HW, float eps,
                                  float* __restrict__ mean, float* __restrict__ rstd) {
    // one wave per (sample, group): lanes stride over the S x Cg partial sums (fixed assignment), butterfly in fp64
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N * G) return;
    const int n = i / G, g = i % G, Cg = C / G;
    double s = 0.0, q = 0.0;
    for (int k = lane; k < S * Cg; k += 64) {
        const int sp = k / Cg, c = g * Cg + (k - sp * Cg);
        s += (double)ws[(((long)n * S + sp) * 2 + 0) * C + c];
        q += (double)ws[(((long)n * S + sp) * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (lane) return;
    const double cnt = (double)HW * Cg;
    const double m = s / cnt;
    double var = q / cnt - m * m;
    if (var < 0.0) var = 0.0;
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
}

// backward stage 2a: part[n][which][c] = sum over splits
__global__ void gn_bwd_part(const float* __restrict__ ws, int N, int S, int C, float* __restrict__ part) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * 2 * C) return;
    const int c = (int)(i % C);
    const int which = (int)((i / C) % 2), n = (int)(i / (2L * C));
    float s = 0.f;
    for (int sp = 0; sp < S; ++sp) s += ws[(((long)n * S + sp) * 2 + which) * C + c];
    part[i] = s;
}
// backward stage 2b: gsum[n*G+g] = (sum_c gamma_c * dbeta_c, sum_c gamma_c * dgamma_c) over the group's channels
__global__ void gn_bwd_gsum(const float* __restrict__ part, const float* __restrict__ gamma, int N, int C, int G,
                            float* __restrict__ gsum) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * G) return;
    const int n = i / G, g = i % G, Cg = C / G;
    float t1 = 0.f, t2 = 0.f;
    for (int c = g * Cg; c < (g + 1) * Cg; ++c) {
        t1 += part[((long)n * 2 + 1) * C + c] * gamma[c];
        t2 += part[((long)n * 2 + 0) * C + c] * gamma[c];
    }
    gsum[i * 2 + 0] = t1;
    gsum[i * 2 + 1] = t2;
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
bool shape_ok(int C, int G) {
    if (C % 8 || C % G) return false;
    const int G8 = C / 8;
    return G8 <= 256 && 256 % G8 == 0;
}

}  // namespace

int umi_gn_splits(int N, long HW) { const int r = gn_rows(N, HW); return (int)((HW + r - 1) / r); }

// returns false when the shape does not qualify (caller falls back to the generic kernels)
bool umi_gn_stats_f16v(const void* x, int ldx, int N, long HW, int C, int G, float eps, float* mean, float* rstd, float* ws,
                       hipStream_t s) {
    if (!shape_ok(C, G) || ldx % 8 || !al16(x)) return false;
    const int S = umi_gn_splits(N, HW);
    hipLaunchKernelGGL(gn_rowsum_v8<0>, dim3(N, S), dim3(256), 0, s, (const half_t*)x, ldx, (const half_t*)nullptr, 0,
                       (const half_t*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, 0, HW, C, G, S, ws, gn_rows(N, HW));
    hipLaunchKernelGGL(gn_stats_finalize, dim3((N * G + 3) / 4), dim3(256), 0, s, (const float*)ws, N, S, C, G, HW, eps,
                       mean, rstd);
    return true;
}

// fills gsum [N*G][2] and part [N][2][C] (the layouts gn_bwd_apply_kernel / reduce_rows2 expect)
bool umi_gn_bwd_reduce_f16v(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx, const float* mean,
                            const float* rstd, const float* gamma, int relu, int N, long HW, int C, int G, float* gsum,
                            float* part, float* ws, hipStream_t s) {
    if (!shape_ok(C, G) || ldx % 8 || lddy % 8 || ldy % 8 || !al16(x) || !al16(dy) || !al16(y)) return false;
    const int S = umi_gn_splits(N, HW);
    hipLaunchKernelGGL(gn_rowsum_v8<1>, dim3(N, S), dim3(256), 0, s, (const half_t*)x, ldx, (const half_t*)dy, lddy,
                       (const half_t*)y, ldy, mean, rstd, relu, HW, C, G, S, ws, gn_rows(N, HW));
    const long np = (long)N * 2 * C;
    hipLaunchKernelGGL(gn_bwd_part, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, (const float*)ws, N, S, C, part);
    hipLaunchKernelGGL(gn_bwd_gsum, dim3((N * G + 255) / 256), dim3(256), 0, s, (const float*)part, gamma, N, C, G, gsum);
    return true;
}
