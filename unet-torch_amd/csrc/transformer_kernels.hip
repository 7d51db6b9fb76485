// Kernels of the TransUNet path that are not convolutions (reference TransUnet/vit_seg_modeling*.py):
// weight standardisation, GroupNorm(+ReLU/+residual), 3x3/s2 max-pool, LayerNorm, exact GELU, dropout, softmax
// attention, bilinear x2 (align_corners) upsampling, broadcast adds -- forward and backward, fp32 math on
// fp32/fp16 NHWC storage ([B,1,N,C] for token tensors).  Correctness-first generic kernels: every reduction is
// fixed-order (no atomics); they are HBM- or latency-bound and small next to the convolutions / GEMMs.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum of two values, result broadcast to every thread (256 threads)
__device__ __forceinline__ void block_sum2(float& a, float& b, float* sh /*[16]*/) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh[w] = a; sh[8 + w] = b; }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    float x = 0.f, y = 0.f;
    for (int i = 0; i < nw; ++i) { x += sh[i]; y += sh[8 + i]; }
    a = x;
    b = y;
}

int grid_for(long items, int cap = 16384) {
    long g = (items + 255) / 256;
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}

// ---------------------------------------------------------------------------------------------------------
// StdConv2d weight standardisation (resnet_skip.py:20-23): per output channel over K = Ci*R*S, biased var, eps in sqrt
__global__ __launch_bounds__(256) void wstd_fwd_kernel(const float* __restrict__ w, float* __restrict__ ws,
                                                       float* __restrict__ rstd, int K, float eps) {
    __shared__ float sh[16];
    const int co = blockIdx.x;
    const float* p = w + (long)co * K;
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < K; i += 256) { float v = p[i]; s += v; q = fmaf(v, v, q); }
    block_sum2(s, q, sh);
    const float mean = s / K;
    float var = q / K - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float r = rsqrtf(var + eps);
    for (int i = threadIdx.x; i < K; i += 256) ws[(long)co * K + i] = (p[i] - mean) * r;
    if (threadIdx.x == 0) rstd[co] = r;
}

// dw = rstd * (g - mean(g) - what * mean(g * what))
__global__ __launch_bounds__(256) void wstd_bwd_kernel(const float* __restrict__ ws, const float* __restrict__ rstd,
                                                       const float* __restrict__ g, float* __restrict__ dw, int K) {
    __shared__ float sh[16];
    const int co = blockIdx.x;
    const float* wh = ws + (long)co * K;
    const float* gp = g + (long)co * K;
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < K; i += 256) { float gv = gp[i]; s += gv; q = fmaf(gv, wh[i], q); }
    block_sum2(s, q, sh);
    const float m1 = s / K, m2 = q / K, r = rstd[co];
    for (int i = threadIdx.x; i < K; i += 256) dw[(long)co * K + i] = r * (gp[i] - m1 - wh[i] * m2);
}

// All StdConv2d weights of a model in ONE launch each way (a R50 hybrid has 52 of them; one launch per conv costs more in
// launch gaps than in work): workgroup = one output channel of one conv, found by binary search over the descriptor table.
struct WstdDesc {          // mirrors umi_wstd_desc
    const float* w;
    float* ws;
    float* rstd;
    long off;              // element offset of this conv in the flat gradient buffers of the backward launch
    int Co, K;
    float eps;
    int blk0;              // first workgroup (= output-channel row) of this conv
    float* dw;             // backward: where this conv's parameter gradient goes (NULL: dw_base + off)
};

__device__ inline int wstd_find(const WstdDesc* d, int n, int blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (d[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ __launch_bounds__(256) void wstd_fwd_multi_kernel(const WstdDesc* __restrict__ descs, int n_desc) {
    __shared__ float sh[16];
    const WstdDesc d = descs[wstd_find(descs, n_desc, blockIdx.x)];
    const int co = blockIdx.x - d.blk0, K = d.K;
    const float* p = d.w + (long)co * K;                    // same arithmetic, in the same order, as wstd_fwd_kernel
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < K; i += 256) { float v = p[i]; s += v; q = fmaf(v, v, q); }
    block_sum2(s, q, sh);
    const float mean = s / K;
    float var = q / K - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float r = rsqrtf(var + d.eps);
    for (int i = threadIdx.x; i < K; i += 256) d.ws[(long)co * K + i] = (p[i] - mean) * r;
    if (threadIdx.x == 0) d.rstd[co] = r;
}

__global__ __launch_bounds__(256) void wstd_bwd_multi_kernel(const WstdDesc* __restrict__ descs, int n_desc,
                                                             const float* __restrict__ g_base, float* __restrict__ dw_base) {
    __shared__ float sh[16];
    const WstdDesc d = descs[wstd_find(descs, n_desc, blockIdx.x)];
    const int co = blockIdx.x - d.blk0, K = d.K;
    const float* wh = d.ws + (long)co * K;
    const float* gp = g_base + d.off + (long)co * K;
    float* dw = (d.dw ? d.dw : dw_base + d.off) + (long)co * K;
    float s = 0.f, q = 0.f;
    for (int i = threadIdx.x; i < K; i += 256) { float gv = gp[i]; s += gv; q = fmaf(gv, wh[i], q); }
    block_sum2(s, q, sh);
    const float m1 = s / K, m2 = q / K, r = d.rstd[co];
    for (int i = threadIdx.x; i < K; i += 256) dw[i] = r * (gp[i] - m1 - wh[i] * m2);
}

// ---------------------------------------------------------------------------------------------------------
// GroupNorm on NHWC: statistics per (sample, group); group = Cg contiguous channels.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int ldx, long HW, int C, int G, float eps,
                                                       float* __restrict__ mean, float* __restrict__ rstd) {
    __shared__ float sh[16];
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G;
    const T* base = x + (long)n * HW * ldx + g * Cg;
    const long total = HW * Cg;
    // two passes (mean, then centred variance) for accuracy: instance-norm-like groups can have |mean| >> std
    float s = 0.f, dummy = 0.f;
    for (long i = threadIdx.x; i < total; i += 256) s += (float)base[(i / Cg) * ldx + (i % Cg)];
    block_sum2(s, dummy, sh);
    const float m = s / (float)total;
    float q = 0.f;
    dummy = 0.f;
    for (long i = threadIdx.x; i < total; i += 256) { float d = (float)base[(i / Cg) * ldx + (i % Cg)] - m; q = fmaf(d, d, q); }
    block_sum2(q, dummy, sh);
    if (threadIdx.x == 0) { mean[blockIdx.x] = m; rstd[blockIdx.x] = rsqrtf(q / (float)total + eps); }
}

// y = [relu]( (x-mean)*rstd*gamma + beta [+ res] )
template <typename T>
__global__ void gn_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, const T* __restrict__ res, int ldr, T* __restrict__ y,
                                int ldy, int relu, int N, long HW, int C, int G) {
    const int Cg = C / G;
    const long total = (long)N * HW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long p = i / C;
        int n = (int)(p / HW);
        int sg = n * G + c / Cg;
        float v = ((float)x[p * ldx + c] - mean[sg]) * rstd[sg] * gamma[c] + beta[c];
        if (res) v += (float)res[p * ldr + c];
        if (relu) v = fmaxf(v, 0.f);
        y[p * ldy + c] = (T)v;
    }
}

// backward stage 1: per (sample, group): s1 = sum dz*gamma, s2 = sum dz*gamma*xhat ; per (sample, channel): dgamma, dbeta partials.
// dz = dy * (relu ? y > 0 : 1).  part layout [N][2][C] (dgamma, dbeta), gsum [N*G][2].
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y,
                                                            int ldy, const T* __restrict__ x, int ldx,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, int relu, long HW, int C, int G,
                                                            float* __restrict__ gsum, float* __restrict__ part) {
    __shared__ float sh[16];
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int Cg = C / G;
    const float m = mean[blockIdx.x], r = rstd[blockIdx.x];
    float t1 = 0.f, t2 = 0.f;
    for (int cc = 0; cc < Cg; ++cc) {
        const int c = g * Cg + cc;
        float a = 0.f, b = 0.f;                      // sum dz*xhat, sum dz for this channel
        for (long p = threadIdx.x; p < HW; p += 256) {
            long row = (long)n * HW + p;
            float dz = (float)dy[row * lddy + c];
            if (relu && !((float)y[row * ldy + c] > 0.f)) dz = 0.f;
            float xh = ((float)x[row * ldx + c] - m) * r;
            a = fmaf(dz, xh, a);
            b += dz;
        }
        block_sum2(a, b, sh);
        if (threadIdx.x == 0) { part[((long)n * 2 + 0) * C + c] = a; part[((long)n * 2 + 1) * C + c] = b; }
        t1 += b * gamma[c];
        t2 += a * gamma[c];
    }
    if (threadIdx.x == 0) { gsum[blockIdx.x * 2 + 0] = t1; gsum[blockIdx.x * 2 + 1] = t2; }
}

// dx = rstd * (dz*gamma - s1/m - xhat*s2/m) ; dres = dz (written to dres when non-null)
template <typename T>
__global__ void gn_bwd_apply_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ y, int ldy,
                                    const T* __restrict__ x, int ldx, const float* __restrict__ mean,
                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                    const float* __restrict__ gsum, int relu, T* __restrict__ dx, int lddx,
                                    T* __restrict__ dres, int lddr, int N, long HW, int C, int G) {
    const int Cg = C / G;
    const float invm = 1.f / (float)(HW * Cg);
    const long total = (long)N * HW * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long p = i / C;
        int n = (int)(p / HW);
        int sg = n * G + c / Cg;
        float dz = (float)dy[p * lddy + c];
        if (relu && !((float)y[p * ldy + c] > 0.f)) dz = 0.f;
        float xh = ((float)x[p * ldx + c] - mean[sg]) * rstd[sg];
        float v = rstd[sg] * (dz * gamma[c] - gsum[sg * 2 + 0] * invm - xh * gsum[sg * 2 + 1] * invm);
        dx[p * lddx + c] = (T)v;
        if (dres) dres[p * lddr + c] = (T)dz;
    }
}

// ---------------------------------------------------------------------------------------------------------
// MaxPool2d(3, stride 2, pad 0)  (resnet_skip.py:147)
template <typename T>
__global__ void pool3s2_fwd_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, int N, int H, int W, int C,
                                   int Ho, int Wo) {
    const long total = (long)N * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long p = i / C;
        int wo = (int)(p % Wo);
        long r = p / Wo;
        int ho = (int)(r % Ho), n = (int)(r / Ho);
        float m = -INFINITY;
        for (int d = 0; d < 9; ++d) {
            float v = (float)x[((long)((long)n * H + 2 * ho + d / 3) * W + 2 * wo + d % 3) * ldx + c];
            m = v > m ? v : m;
        }
        y[p * ldy + c] = (T)m;
    }
}

// gather form: input pixel (h,w) receives dy of every window whose FIRST max (scan order) it is
template <typename T>
__global__ void pool3s2_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx, T* __restrict__ dx,
                                   int lddx, int N, int H, int W, int C, int Ho, int Wo) {
    const long total = (long)N * H * W * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long p = i / C;
        int w = (int)(p % W);
        long r = p / W;
        int h = (int)(r % H), n = (int)(r / H);
        float acc = 0.f;
        for (int ho = (h - 2 + 1) / 2 < 0 ? 0 : (h - 1) / 2; ho <= h / 2 && ho < Ho; ++ho) {
            if (2 * ho > h || 2 * ho + 2 < h) continue;
            for (int wo = (w - 1) / 2 < 0 ? 0 : (w - 1) / 2; wo <= w / 2 && wo < Wo; ++wo) {
                if (2 * wo > w || 2 * wo + 2 < w) continue;
                float m = -INFINITY;
                int best = 0;
                for (int d = 0; d < 9; ++d) {
                    float v = (float)x[((long)((long)n * H + 2 * ho + d / 3) * W + 2 * wo + d % 3) * ldx + c];
                    if (d == 0 || v > m) { m = v; best = d; }
                }
                if (2 * ho + best / 3 == h && 2 * wo + best % 3 == w)
                    acc += (float)dy[((long)((long)n * Ho + ho) * Wo + wo) * lddy + c];
            }
        }
        dx[p * lddx + c] = (T)acc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (one wave per row), vit_seg_modeling.py:172-173 (eps 1e-6)
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y, int ldy,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long M, int C,
                                                     float eps) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const T* xp = x + row * ldx;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += (float)xp[c];
    s = wave_sum(s);
    const float m = s / C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { float d = (float)xp[c] - m; q = fmaf(d, d, q); }
    q = wave_sum(q);
    const float r = rsqrtf(q / C + eps);
    for (int c = lane; c < C; c += 64) y[row * ldy + c] = (T)(((float)xp[c] - m) * r * gamma[c] + beta[c]);
    if (lane == 0) { mean[row] = m; rstd[row] = r; }
}

// dx per row; dgamma/dbeta partial rows [nblk][2][C] (each block handles RPB rows, fixed order)
constexpr int LN_RPB = 64;
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, int lddy, const T* __restrict__ x, int ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, T* __restrict__ dx, int lddx,
                                                     float* __restrict__ part, long M, int C) {
    extern __shared__ float lsm[];                  // [4 waves][2][C]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* mydg = lsm + (wave * 2 + 0) * C;
    float* mydb = lsm + (wave * 2 + 1) * C;
    for (int c = lane; c < C; c += 64) { mydg[c] = 0.f; mydb[c] = 0.f; }
    const long r0 = (long)blockIdx.x * LN_RPB;
    for (int k = wave; k < LN_RPB; k += 4) {
        long row = r0 + k;
        if (row >= M) break;
        const float m = mean[row], r = rstd[row];
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < C; c += 64) {
            float g = (float)dy[row * lddy + c] * gamma[c];
            float xh = ((float)x[row * ldx + c] - m) * r;
            s1 += g;
            s2 = fmaf(g, xh, s2);
        }
        s1 = wave_sum(s1) / C;
        s2 = wave_sum(s2) / C;
        for (int c = lane; c < C; c += 64) {
            float d = (float)dy[row * lddy + c];
            float xh = ((float)x[row * ldx + c] - m) * r;
            dx[row * lddx + c] = (T)(r * (d * gamma[c] - s1 - xh * s2));
            mydg[c] = fmaf(d, xh, mydg[c]);
            mydb[c] += d;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 4; ++w) { a += lsm[(w * 2 + 0) * C + c]; b += lsm[(w * 2 + 1) * C + c]; }
        part[((long)blockIdx.x * 2 + 0) * C + c] = a;
        part[((long)blockIdx.x * 2 + 1) * C + c] = b;
    }
}

// ---------------------------------------------------------------------------------------------------------
// elementwise: exact GELU (F.gelu default, vit_seg_modeling.py:102,115), add, broadcast add, dropout
__device__ __forceinline__ float gelu_f(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float u) {
    return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * __expf(-0.5f * u * u);
}

// mode 0: y = gelu(x) ; 1: y = g * gelu'(x) (x = pre-activation, g = upstream grad) ; 2: y = x + g ; 3: y = x + g[bcast over rows]
template <typename T>
__global__ void ew_kernel(int mode, const T* __restrict__ x, int ldx, const T* __restrict__ g, int ldg, T* __restrict__ y,
                          int ldy, long M, int C, long bcast_rows) {
    const long total = M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long r = i / C;
        float xv = (float)x[r * ldx + c], o;
        if (mode == 0) o = gelu_f(xv);
        else if (mode == 1) o = (float)g[r * ldg + c] * gelu_df(xv);
        else if (mode == 2) o = xv + (float)g[r * ldg + c];
        else o = xv + (float)g[(r % bcast_rows) * ldg + c];
        y[r * ldy + c] = (T)o;
    }
}

__device__ __forceinline__ unsigned hash32(unsigned a, unsigned b) {           // counter-based RNG (own stream)
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
// fwd (g == nullptr): keep = u >= p ; y = keep ? x/(1-p) : 0 ; mask byte written.   bwd: y = g * mask/(1-p)
template <typename T>
__global__ void dropout_kernel(const T* __restrict__ x, int ldx, T* __restrict__ y, int ldy, unsigned char* __restrict__ mask,
                               int bwd, float p, unsigned seed, long M, int C, const float4* __restrict__ tx,
                               const unsigned* __restrict__ seed_dev) {
    if (seed_dev) seed += seed_dev[0] * 0x9E3779B9u;          // per-replay stream offset when the step runs from a HIP graph
    const long total = M * C;
    const float scale = 1.f / (1.f - p);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long r = i / C;
        unsigned char k;
        if (bwd) k = mask[i];
        else {
            float u = (hash32((unsigned)i, seed ^ (unsigned)(i >> 32)) >> 8) * (1.f / 16777216.f);
            k = u >= p;
            mask[i] = k;
        }
        float v = (float)x[r * ldx + c];
        if (tx && !bwd) v = umi_tx(v, tx[c]);                // forward on a lazily-activated tensor (U-Net Down / Up dropout)
        y[r * ldy + c] = (T)(k ? v * scale : 0.f);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Softmax attention, heads = channel slices of width D (vit_seg_modeling.py:73-91).  One thread = one query row of
// one (batch, head); K/V tiles of 64 keys staged in LDS; online softmax; log-sum-exp saved for backward.
template <typename T, int D>
__global__ __launch_bounds__(64) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                      int ld, T* __restrict__ o, int ldo, float* __restrict__ lse, int B,
                                                      int N, int Hh, float scale) {
    __shared__ float ks[64][D + 1], vs[64][D + 1];
    const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
    const int qi = blockIdx.x * 64 + threadIdx.x;
    const bool act = qi < N;
    float qr[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) { qr[d] = act ? (float)q[((long)b * N + qi) * ld + h * D + d] * scale : 0.f; acc[d] = 0.f; }
    float mx = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < N; k0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * D; i += 64) {
            int kk = i / D, d = i % D;
            bool in = k0 + kk < N;
            ks[kk][d] = in ? (float)k[((long)b * N + k0 + kk) * ld + h * D + d] : 0.f;
            vs[kk][d] = in ? (float)v[((long)b * N + k0 + kk) * ld + h * D + d] : 0.f;
        }
        __syncthreads();
        const int kn = (N - k0) < 64 ? (N - k0) : 64;
        for (int kk = 0; kk < kn; ++kk) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) s = fmaf(qr[d], ks[kk][d], s);
            float mn = fmaxf(mx, s);
            float corr = __expf(mx - mn), pr = __expf(s - mn);
            l = l * corr + pr;
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = fmaf(acc[d], corr, pr * vs[kk][d]);
            mx = mn;
        }
    }
    if (act) {
        const float inv = 1.f / l;
#pragma unroll
        for (int d = 0; d < D; ++d) o[((long)b * N + qi) * ldo + h * D + d] = (T)(acc[d] * inv);
        lse[(long)bh * N + qi] = mx + __logf(l);
    }
}

// backward, query side: delta = sum(dO*O); dQ[q] = scale * sum_k p*(dO.V_k - delta) K_k
template <typename T, int D>
__global__ __launch_bounds__(64) void attn_bwd_q_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                        int ld, const T* __restrict__ o, const T* __restrict__ dO, int ldo,
                                                        const float* __restrict__ lse, T* __restrict__ dq, int lddq,
                                                        float* __restrict__ delta, int B, int N, int Hh, float scale) {
    __shared__ float ks[64][D + 1], vs[64][D + 1];
    const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
    const int qi = blockIdx.x * 64 + threadIdx.x;
    const bool act = qi < N;
    float qr[D], dor[D], acc[D];
    float dl = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        long idx = ((long)b * N + qi) * ld + h * D + d, ido = ((long)b * N + qi) * ldo + h * D + d;
        qr[d] = act ? (float)q[idx] * scale : 0.f;
        dor[d] = act ? (float)dO[ido] : 0.f;
        dl = fmaf(dor[d], act ? (float)o[ido] : 0.f, dl);
        acc[d] = 0.f;
    }
    const float L = act ? lse[(long)bh * N + qi] : 0.f;
    for (int k0 = 0; k0 < N; k0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * D; i += 64) {
            int kk = i / D, d = i % D;
            bool in = k0 + kk < N;
            ks[kk][d] = in ? (float)k[((long)b * N + k0 + kk) * ld + h * D + d] : 0.f;
            vs[kk][d] = in ? (float)v[((long)b * N + k0 + kk) * ld + h * D + d] : 0.f;
        }
        __syncthreads();
        const int kn = (N - k0) < 64 ? (N - k0) : 64;
        for (int kk = 0; kk < kn; ++kk) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) { s = fmaf(qr[d], ks[kk][d], s); dp = fmaf(dor[d], vs[kk][d], dp); }
            float ds = __expf(s - L) * (dp - dl);
#pragma unroll
            for (int d = 0; d < D; ++d) acc[d] = fmaf(ds, ks[kk][d], acc[d]);
        }
    }
    if (act) {
#pragma unroll
        for (int d = 0; d < D; ++d) dq[((long)b * N + qi) * lddq + h * D + d] = (T)(acc[d] * scale);
        delta[(long)bh * N + qi] = dl;
    }
}

// backward, key side: dV[k] = sum_q p*dO_q ; dK[k] = scale * sum_q p*(dO_q.V_k - delta_q) Q_q
template <typename T, int D>
__global__ __launch_bounds__(64) void attn_bwd_kv_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                         int ld, const T* __restrict__ dO, int ldo, const float* __restrict__ lse,
                                                         const float* __restrict__ delta, T* __restrict__ dk, T* __restrict__ dv,
                                                         int lddk, int B, int N, int Hh, float scale) {
    __shared__ float qs[64][D + 1], ds_[64][D + 1];
    __shared__ float ls[64], dls[64];
    const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
    const int ki = blockIdx.x * 64 + threadIdx.x;
    const bool act = ki < N;
    float kr[D], vr[D], ak[D], av[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        long idx = ((long)b * N + ki) * ld + h * D + d;
        kr[d] = act ? (float)k[idx] : 0.f;
        vr[d] = act ? (float)v[idx] : 0.f;
        ak[d] = av[d] = 0.f;
    }
    for (int q0 = 0; q0 < N; q0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * D; i += 64) {
            int qq = i / D, d = i % D;
            bool in = q0 + qq < N;
            qs[qq][d] = in ? (float)q[((long)b * N + q0 + qq) * ld + h * D + d] * scale : 0.f;
            ds_[qq][d] = in ? (float)dO[((long)b * N + q0 + qq) * ldo + h * D + d] : 0.f;
        }
        if (q0 + threadIdx.x < N) { ls[threadIdx.x] = lse[(long)bh * N + q0 + threadIdx.x]; dls[threadIdx.x] = delta[(long)bh * N + q0 + threadIdx.x]; }
        __syncthreads();
        const int qn = (N - q0) < 64 ? (N - q0) : 64;
        for (int qq = 0; qq < qn; ++qq) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) { s = fmaf(qs[qq][d], kr[d], s); dp = fmaf(ds_[qq][d], vr[d], dp); }
            float p = __expf(s - ls[qq]);
            float dsv = p * (dp - dls[qq]);
#pragma unroll
            for (int d = 0; d < D; ++d) { av[d] = fmaf(p, ds_[qq][d], av[d]); ak[d] = fmaf(dsv, qs[qq][d], ak[d]); }
        }
    }
    if (act) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            long idx = ((long)b * N + ki) * lddk + h * D + d;
            dk[idx] = (T)ak[d];            // qs already carries `scale`
            dv[idx] = (T)av[d];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// UpsamplingBilinear2d(scale_factor=2) = align_corners=True (vit_seg_modeling.py:307)
template <typename T>
__global__ void bilinear2x_fwd_kernel(const T* __restrict__ x, int ldx, const float4* __restrict__ tx, T* __restrict__ y, int ldy,
                                      int N, int H, int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long total = (long)N * Ho * Wo * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long p = i / C;
        int wo = (int)(p % Wo);
        long r = p / Wo;
        int ho = (int)(r % Ho), n = (int)(r / Ho);
        float fy = ho * sy, fx = wo * sx;
        int y0 = (int)fy, x0 = (int)fx;
        int y1 = y0 + 1 < H ? y0 + 1 : H - 1, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
        float ly = fy - y0, lx = fx - x0;
        const T* b = x + (long)n * H * W * ldx + c;
        float v00 = (float)b[((long)y0 * W + x0) * ldx], v01 = (float)b[((long)y0 * W + x1) * ldx];
        float v10 = (float)b[((long)y1 * W + x0) * ldx], v11 = (float)b[((long)y1 * W + x1) * ldx];
        if (tx) { const float4 t = tx[c]; v00 = umi_tx(v00, t); v01 = umi_tx(v01, t); v10 = umi_tx(v10, t); v11 = umi_tx(v11, t); }
        float v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
        y[p * ldy + c] = (T)v;
    }
}

// gather form of the adjoint: each input pixel sums the output pixels that read it (deterministic)
template <typename T>
__global__ void bilinear2x_bwd_kernel(const T* __restrict__ dy, int lddy, T* __restrict__ dx, int lddx, int N, int H, int W, int C) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const long total = (long)N * H * W * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long p = i / C;
        int w = (int)(p % W);
        long r = p / W;
        int h = (int)(r % H), n = (int)(r / H);
        // output rows whose source interval [y0, y0+1] touches h:  fy in (h-1, h+1)
        int ho_lo = sy > 0.f ? (int)floorf((h - 1) / sy) : 0, ho_hi = sy > 0.f ? (int)ceilf((h + 1) / sy) : Ho - 1;
        int wo_lo = sx > 0.f ? (int)floorf((w - 1) / sx) : 0, wo_hi = sx > 0.f ? (int)ceilf((w + 1) / sx) : Wo - 1;
        ho_lo = ho_lo < 0 ? 0 : ho_lo; wo_lo = wo_lo < 0 ? 0 : wo_lo;
        ho_hi = ho_hi > Ho - 1 ? Ho - 1 : ho_hi; wo_hi = wo_hi > Wo - 1 ? Wo - 1 : wo_hi;
        float acc = 0.f;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            float fy = ho * sy;
            int y0 = (int)fy;
            int y1 = y0 + 1 < H ? y0 + 1 : H - 1;
            float ly = fy - y0;
            float wy = (y0 == h ? 1.f - ly : 0.f) + (y1 == h ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                float fx = wo * sx;
                int x0 = (int)fx;
                int x1 = x0 + 1 < W ? x0 + 1 : W - 1;
                float lx = fx - x0;
                float wx = (x0 == w ? 1.f - lx : 0.f) + (x1 == w ? lx : 0.f);
                if (wx == 0.f) continue;
                acc = fmaf(wy * wx, (float)dy[((long)((long)n * Ho + ho) * Wo + wo) * lddy + c], acc);
            }
        }
        dx[p * lddx + c] = (T)acc;
    }
}

}  // namespace

void umi_launch_reduce_rows2(const float* ws, int rows, int C, float* out0, float* out1, float scale, hipStream_t s);
// groupnorm_f16.hip
int umi_gn_splits(int N, long HW);
bool umi_gn_fwd_f16v(const void* x, int ldx, const float* gamma, const float* beta, const void* res, int ldr, void* y, int ldy,
                     float* mean, float* rstd, int relu, int N, long HW, int C, int G, float eps, float* ws, hipStream_t s);
bool umi_gn_bwd_f16v(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx, const float* mean,
                     const float* rstd, const float* gamma, int relu, void* dx, int lddx, void* dres, int lddr, int N, long HW,
                     int C, int G, float* part, float* ws, hipStream_t s);
void umi_gn_param_grads_launch(int n, const float* const* parts, const int* Cs, int N, float* const* dgammas, float* const* dbetas,
                               float scale, hipStream_t s);
// elementwise_tu_f16.hip
bool umi_ew_f16v(int mode, const void* x, int ldx, const void* g, int ldg, void* y, int ldy, long M, int C, long bcast_rows,
                 hipStream_t s);
bool umi_pool3s2_fwd_f16v(const void* x, int ldx, void* y, int ldy, void* idx, int N, int H, int W, int C, hipStream_t s);
bool umi_pool3s2_bwd_f16v(const void* dy, int lddy, const void* idx, void* dx, int lddx, int N, int H, int W, int C, hipStream_t s);
bool umi_dropout_f16v(const void* x, int ldx, void* y, int ldy, void* mask, int backward, float p, unsigned seed, long M, int C,
                      const void* tx, const unsigned* seed_dev, hipStream_t s);
bool umi_dropout_fused_f16v(const void* x, int ldx, void* y, int ldy, void* mask, int backward, float p, unsigned seed, long M,
                            int C, const unsigned* seed_dev, const void* aux, int ldaux, int gelu, hipStream_t s);
int umi_ln_bwd_rows_f16v();
bool umi_ln_bwd_f16v(const void* dy, int lddy, const void* x, int ldx, const float* gamma, const float* mean, const float* rstd,
                     void* dx, int lddx, float* part, long M, int C, hipStream_t s);
bool umi_bilinear2x_f16v(const void* x, int ldx, const void* tx, void* y, int ldy, int backward, int N, int H, int W, int C,
                         hipStream_t s);
// attention_mfma.hip
bool umi_attn_mfma_ok(int D, int ld, int ldo, int dtype, const void* a, const void* b, const void* c);
int umi_attn_fwd_mfma(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse, int B, int N, int Hh,
                      hipStream_t s);
int umi_attn_bwd_mfma(const void* q, const void* k, const void* v, int ld, const void* o, const void* dO, int ldo,
                      const float* lse, void* dq, void* dk, void* dv, int ldd, float* delta, int B, int N, int Hh,
                      hipStream_t s);

#define DT_SWITCH(dtype, CALL_F32, CALL_F16)             \
    if ((dtype) == UMI_F32) { CALL_F32; }                \
    else if ((dtype) == UMI_F16) { CALL_F16; }           \
    else return UMI_ERR_BADARG;

extern "C" int umi_wstd_fwd(const float* w, float* wstd, float* rstd, int Co, int K, float eps, umi_stream_t st) {
    if (!w || !wstd || !rstd || Co <= 0 || K <= 0) return UMI_ERR_BADARG;
    hipLaunchKernelGGL(wstd_fwd_kernel, dim3(Co), dim3(256), 0, (hipStream_t)st, w, wstd, rstd, K, eps);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
extern "C" int umi_wstd_bwd(const float* wstd, const float* rstd, const float* g, float* dw, int Co, int K, umi_stream_t st) {
    if (!wstd || !rstd || !g || !dw || Co <= 0 || K <= 0) return UMI_ERR_BADARG;
    hipLaunchKernelGGL(wstd_bwd_kernel, dim3(Co), dim3(256), 0, (hipStream_t)st, wstd, rstd, g, dw, K);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_wstd_fwd_multi(const void* descs, int n_desc, int total_rows, umi_stream_t st) {
    if (!descs || n_desc <= 0 || total_rows <= 0) return UMI_ERR_BADARG;
    hipLaunchKernelGGL(wstd_fwd_multi_kernel, dim3(total_rows), dim3(256), 0, (hipStream_t)st, (const WstdDesc*)descs, n_desc);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
extern "C" int umi_wstd_bwd_multi(const void* descs, int n_desc, int total_rows, const float* g_base, float* dw_base,
                                  umi_stream_t st) {
    if (!descs || !g_base || n_desc <= 0 || total_rows <= 0) return UMI_ERR_BADARG;      // dw_base may be NULL if every entry has .dw
    hipLaunchKernelGGL(wstd_bwd_multi_kernel, dim3(total_rows), dim3(256), 0, (hipStream_t)st, (const WstdDesc*)descs, n_desc,
                       g_base, dw_base);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" size_t umi_gn_fwd_ws_bytes(int N, long HW, int C) { return (size_t)N * umi_gn_splits(N, HW) * 2 * C * sizeof(float); }

extern "C" int umi_gn_fwd(const void* x, int ldx, const float* gamma, const float* beta, const void* res, int ldr, void* y,
                          int ldy, float* mean, float* rstd, int relu, int N, long HW, int C, int G, float eps, int dtype,
                          void* ws, size_t ws_bytes, umi_stream_t st) {
    if (!x || !y || !gamma || !beta || !mean || !rstd || N <= 0 || HW <= 0 || C <= 0 || G <= 0 || C % G) return UMI_ERR_BADARG;
    hipStream_t s = (hipStream_t)st;
    const int grid = grid_for((long)N * HW * C);
    if (dtype == UMI_F16 && ws && ws_bytes >= umi_gn_fwd_ws_bytes(N, HW, C) &&
        umi_gn_fwd_f16v(x, ldx, gamma, beta, res, ldr, y, ldy, mean, rstd, relu, N, HW, C, G, eps, (float*)ws, s)) {
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    DT_SWITCH(dtype,
        hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(N * G), dim3(256), 0, s, (const float*)x, ldx, HW, C, G, eps, mean, rstd);
        hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, mean, rstd, gamma, beta, (const float*)res, ldr, (float*)y, ldy, relu, N, HW, C, G),
        hipLaunchKernelGGL(gn_stats_kernel<half_t>, dim3(N * G), dim3(256), 0, s, (const half_t*)x, ldx, HW, C, G, eps, mean, rstd);
        hipLaunchKernelGGL(gn_apply_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, mean, rstd, gamma, beta, (const half_t*)res, ldr, (half_t*)y, ldy, relu, N, HW, C, G))
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" size_t umi_gn_bwd_ws_bytes(int N, long HW, int C, int G) {
    return ((size_t)N * G * 2 + (size_t)N * 2 * C + (size_t)N * umi_gn_splits(N, HW) * 2 * C) * sizeof(float);
}

extern "C" int umi_gn_bwd(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx, const float* mean,
                          const float* rstd, const float* gamma, int relu, void* dx, int lddx, void* dres, int lddr,
                          float* dgamma, float* dbeta, float out_scale, int N, long HW, int C, int G, int dtype, void* ws,
                          size_t ws_bytes, float* part_out, umi_stream_t st) {
    if (!dy || !y || !x || !dx || !ws || C % G || (!part_out && (!dgamma || !dbeta))) return UMI_ERR_BADARG;
    if (ws_bytes < umi_gn_bwd_ws_bytes(N, HW, C, G)) return UMI_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)st;
    float* gsum = (float*)ws;
    float* part = part_out ? part_out : gsum + (size_t)N * G * 2;
    float* stage1 = gsum + (size_t)N * G * 2 + (size_t)N * 2 * C;
    const int grid = grid_for((long)N * HW * C);
    if (dtype == UMI_F16 &&
        umi_gn_bwd_f16v(dy, lddy, y, ldy, x, ldx, mean, rstd, gamma, relu, dx, lddx, dres, lddr, N, HW, C, G, part, stage1, s)) {
        UMI_LAUNCH_CHECK();
        if (dgamma && dbeta) {           // (a caller that keeps the per-sample rows may sum them later: umi_gn_param_grads_group)
            umi_launch_reduce_rows2(part, N, C, dgamma, dbeta, out_scale, s);
            UMI_LAUNCH_CHECK();
        }
        return UMI_OK;
    }
    DT_SWITCH(dtype,
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, dim3(N * G), dim3(256), 0, s, (const float*)dy, lddy, (const float*)y, ldy, (const float*)x, ldx, mean, rstd, gamma, relu, HW, C, G, gsum, part);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, lddy, (const float*)y, ldy, (const float*)x, ldx, mean, rstd, gamma, gsum, relu, (float*)dx, lddx, (float*)dres, lddr, N, HW, C, G),
        hipLaunchKernelGGL(gn_bwd_reduce_kernel<half_t>, dim3(N * G), dim3(256), 0, s, (const half_t*)dy, lddy, (const half_t*)y, ldy, (const half_t*)x, ldx, mean, rstd, gamma, relu, HW, C, G, gsum, part);
        hipLaunchKernelGGL(gn_bwd_apply_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)dy, lddy, (const half_t*)y, ldy, (const half_t*)x, ldx, mean, rstd, gamma, gsum, relu, (half_t*)dx, lddx, (half_t*)dres, lddr, N, HW, C, G))
    UMI_LAUNCH_CHECK();
    if (dgamma && dbeta) {
        umi_launch_reduce_rows2(part, N, C, dgamma, dbeta, out_scale, s);
        UMI_LAUNCH_CHECK();
    }
    return UMI_OK;
}

// dgamma / dbeta of n GroupNorm layers from the per-sample rows umi_gn_bwd left in part_out, one launch per 16 layers
extern "C" int umi_gn_param_grads_group(int n, const float* const* parts, const int* Cs, int N, float* const* dgammas,
                                        float* const* dbetas, float out_scale, umi_stream_t st) {
    if (n <= 0 || !parts || !Cs || !dgammas || !dbetas || N <= 0) return UMI_ERR_BADARG;
    for (int i = 0; i < n; ++i)
        if (!parts[i] || !dgammas[i] || !dbetas[i] || Cs[i] <= 0) return UMI_ERR_BADARG;
    umi_gn_param_grads_launch(n, parts, Cs, N, dgammas, dbetas, out_scale, (hipStream_t)st);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_pool3s2_fwd(const void* x, int ldx, void* y, int ldy, void* idx, int N, int H, int W, int C, int dtype,
                               umi_stream_t st) {
    if (!x || !y || H < 3 || W < 3) return UMI_ERR_BADARG;
    if (dtype == UMI_F16 && umi_pool3s2_fwd_f16v(x, ldx, y, ldy, idx, N, H, W, C, (hipStream_t)st)) {
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    if (idx) return UMI_ERR_UNSUPPORTED;           // only the vectorised fp16 kernel records the winning taps
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    const int grid = grid_for((long)N * Ho * Wo * C);
    DT_SWITCH(dtype,
        hipLaunchKernelGGL(pool3s2_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const float*)x, ldx, (float*)y, ldy, N, H, W, C, Ho, Wo),
        hipLaunchKernelGGL(pool3s2_fwd_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const half_t*)x, ldx, (half_t*)y, ldy, N, H, W, C, Ho, Wo))
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
extern "C" int umi_pool3s2_bwd(const void* dy, int lddy, const void* x, int ldx, const void* idx, void* dx, int lddx, int N, int H,
                               int W, int C, int dtype, umi_stream_t st) {
    if (!dy || !x || !dx || H < 3 || W < 3) return UMI_ERR_BADARG;
    if (idx && dtype == UMI_F16 && umi_pool3s2_bwd_f16v(dy, lddy, idx, dx, lddx, N, H, W, C, (hipStream_t)st)) {
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    const int grid = grid_for((long)N * H * W * C);
    DT_SWITCH(dtype,
        hipLaunchKernelGGL(pool3s2_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const float*)dy, lddy, (const float*)x, ldx, (float*)dx, lddx, N, H, W, C, Ho, Wo),
        hipLaunchKernelGGL(pool3s2_bwd_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const half_t*)dy, lddy, (const half_t*)x, ldx, (half_t*)dx, lddx, N, H, W, C, Ho, Wo))
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_ln_fwd(const void* x, int ldx, const float* gamma, const float* beta, void* y, int ldy, float* mean,
                          float* rstd, long M, int C, float eps, int dtype, umi_stream_t st) {
    if (!x || !y || !gamma || !beta || !mean || !rstd || M <= 0 || C <= 0) return UMI_ERR_BADARG;
    const int grid = (int)((M + 3) / 4);
    DT_SWITCH(dtype,
        hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const float*)x, ldx, gamma, beta, (float*)y, ldy, mean, rstd, M, C, eps),
        hipLaunchKernelGGL(ln_fwd_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const half_t*)x, ldx, gamma, beta, (half_t*)y, ldy, mean, rstd, M, C, eps))
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
extern "C" size_t umi_ln_bwd_ws_bytes(long M, int C) {
    const int rpb = umi_ln_bwd_rows_f16v() < LN_RPB ? umi_ln_bwd_rows_f16v() : LN_RPB;      // the path with more partial rows
    return (size_t)((M + rpb - 1) / rpb) * 2 * C * sizeof(float);
}
extern "C" int umi_ln_bwd(const void* dy, int lddy, const void* x, int ldx, const float* gamma, const float* mean,
                          const float* rstd, void* dx, int lddx, float* dgamma, float* dbeta, float out_scale, long M, int C,
                          int dtype, void* ws, size_t ws_bytes, int* rows_out, umi_stream_t st) {
    // dgamma == dbeta == NULL: the partial rows [rows][2][C] stay in `ws` (the caller's own buffer then, not scratch) for a
    // later umi_gn_param_grads_group over many layers; *rows_out = their count
    if (!dy || !x || !dx || !ws || (!dgamma != !dbeta) || (!dgamma && !rows_out)) return UMI_ERR_BADARG;
    if (ws_bytes < umi_ln_bwd_ws_bytes(M, C)) return UMI_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)st;
    if (dtype == UMI_F16 && umi_ln_bwd_f16v(dy, lddy, x, ldx, gamma, mean, rstd, dx, lddx, (float*)ws, M, C, s)) {
        UMI_LAUNCH_CHECK();
        const int vrows = (int)((M + umi_ln_bwd_rows_f16v() - 1) / umi_ln_bwd_rows_f16v());
        if (rows_out) *rows_out = vrows;
        if (dgamma) {
            umi_launch_reduce_rows2((const float*)ws, vrows, C, dgamma, dbeta, out_scale, s);
            UMI_LAUNCH_CHECK();
        }
        return UMI_OK;
    }
    if ((size_t)8 * C * sizeof(float) > 64 * 1024) return UMI_ERR_UNSUPPORTED;
    const int rows = (int)((M + LN_RPB - 1) / LN_RPB);
    const size_t smem = (size_t)8 * C * sizeof(float);
    DT_SWITCH(dtype,
        hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(rows), dim3(256), smem, s, (const float*)dy, lddy, (const float*)x, ldx, gamma, mean, rstd, (float*)dx, lddx, (float*)ws, M, C),
        hipLaunchKernelGGL(ln_bwd_kernel<half_t>, dim3(rows), dim3(256), smem, s, (const half_t*)dy, lddy, (const half_t*)x, ldx, gamma, mean, rstd, (half_t*)dx, lddx, (float*)ws, M, C))
    UMI_LAUNCH_CHECK();
    if (rows_out) *rows_out = rows;
    if (dgamma) {
        umi_launch_reduce_rows2((const float*)ws, rows, C, dgamma, dbeta, out_scale, s);
        UMI_LAUNCH_CHECK();
    }
    return UMI_OK;
}

extern "C" int umi_elementwise(int mode, const void* x, int ldx, const void* g, int ldg, void* y, int ldy, long M, int C,
                               long bcast_rows, int dtype, umi_stream_t st) {
    if (!x || !y || M <= 0 || C <= 0 || mode < 0 || mode > 3 || (mode && !g)) return UMI_ERR_BADARG;
    if (dtype == UMI_F16 && umi_ew_f16v(mode, x, ldx, g, ldg, y, ldy, M, C, bcast_rows > 0 ? bcast_rows : 1, (hipStream_t)st)) {
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    const int grid = grid_for(M * C);
    DT_SWITCH(dtype,
        hipLaunchKernelGGL(ew_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)st, mode, (const float*)x, ldx, (const float*)g, ldg, (float*)y, ldy, M, C, bcast_rows > 0 ? bcast_rows : 1),
        hipLaunchKernelGGL(ew_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)st, mode, (const half_t*)x, ldx, (const half_t*)g, ldg, (half_t*)y, ldy, M, C, bcast_rows > 0 ? bcast_rows : 1))
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// umi_dropout fused with the GELU before it and / or the residual add after it (fp16, C % 8 == 0: UMI_ERR_UNSUPPORTED
// otherwise, run the separate kernels):  forward  y = dropout(gelu ? GELU(x) : x) + (aux ? aux : 0)
//                                         backward y = dropout'(x) * (gelu ? GELU'(aux) : 1), aux = the forward's x
extern "C" int umi_dropout_fused(const void* x, int ldx, void* y, int ldy, void* mask, int backward, float p, unsigned seed, long M,
                                 int C, int dtype, const unsigned* seed_dev, const void* aux, int ldaux, int gelu,
                                 umi_stream_t st) {
    if (!x || !y || !mask || M <= 0 || C <= 0 || p < 0.f || p >= 1.f) return UMI_ERR_BADARG;
    if (dtype != UMI_F16 ||
        !umi_dropout_fused_f16v(x, ldx, y, ldy, mask, backward, p, seed, M, C, seed_dev, aux, ldaux, gelu, (hipStream_t)st))
        return UMI_ERR_UNSUPPORTED;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_dropout(const void* x, int ldx, void* y, int ldy, void* mask, int backward, float p, unsigned seed, long M,
                           int C, int dtype, const void* tx, const unsigned* seed_dev, umi_stream_t st) {
    if (!x || !y || !mask || p < 0.f || p >= 1.f) return UMI_ERR_BADARG;
    if (dtype == UMI_F16 && umi_dropout_f16v(x, ldx, y, ldy, mask, backward, p, seed, M, C, tx, seed_dev, (hipStream_t)st)) {
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    const int grid = grid_for(M * C);
    DT_SWITCH(dtype,
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const float*)x, ldx, (float*)y, ldy, (unsigned char*)mask, backward, p, seed, M, C, (const float4*)tx, seed_dev),
        hipLaunchKernelGGL(dropout_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)st, (const half_t*)x, ldx, (half_t*)y, ldy, (unsigned char*)mask, backward, p, seed, M, C, (const float4*)tx, seed_dev))
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

template <typename T>
static int attn_fwd_t(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse, int B, int N, int Hh,
                      int D, hipStream_t s) {
    const float scale = 1.f / sqrtf((float)D);
    dim3 grid((N + 63) / 64, B * Hh), block(64);
    if (D == 64) hipLaunchKernelGGL((attn_fwd_kernel<T, 64>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld, (T*)o, ldo, lse, B, N, Hh, scale);
    else if (D == 32) hipLaunchKernelGGL((attn_fwd_kernel<T, 32>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld, (T*)o, ldo, lse, B, N, Hh, scale);
    else if (D == 16) hipLaunchKernelGGL((attn_fwd_kernel<T, 16>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld, (T*)o, ldo, lse, B, N, Hh, scale);
    else return UMI_ERR_UNSUPPORTED;
    return UMI_OK;
}
extern "C" int umi_attn_fwd(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse, int B, int N,
                            int heads, int D, int dtype, umi_stream_t st) {
    if (!q || !k || !v || !o || !lse || B <= 0 || N <= 0 || heads <= 0) return UMI_ERR_BADARG;
    if (umi_attn_mfma_ok(D, ld, ldo, dtype, q, k, v) && (((uintptr_t)o) & 15) == 0)
        return umi_attn_fwd_mfma(q, k, v, ld, o, ldo, lse, B, N, heads, (hipStream_t)st);
    int rc;
    if (dtype == UMI_F32) rc = attn_fwd_t<float>(q, k, v, ld, o, ldo, lse, B, N, heads, D, (hipStream_t)st);
    else if (dtype == UMI_F16) rc = attn_fwd_t<half_t>(q, k, v, ld, o, ldo, lse, B, N, heads, D, (hipStream_t)st);
    else return UMI_ERR_BADARG;
    if (rc) return rc;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

template <typename T>
static int attn_bwd_t(const void* q, const void* k, const void* v, int ld, const void* o, const void* dO, int ldo,
                      const float* lse, void* dq, void* dk, void* dv, int ldd, float* delta, int B, int N, int Hh, int D,
                      hipStream_t s) {
    const float scale = 1.f / sqrtf((float)D);
    dim3 grid((N + 63) / 64, B * Hh), block(64);
#define GO(DD)                                                                                                              \
    hipLaunchKernelGGL((attn_bwd_q_kernel<T, DD>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld, (const T*)o, \
                       (const T*)dO, ldo, lse, (T*)dq, ldd, delta, B, N, Hh, scale);                                         \
    hipLaunchKernelGGL((attn_bwd_kv_kernel<T, DD>), grid, block, 0, s, (const T*)q, (const T*)k, (const T*)v, ld, (const T*)dO, \
                       ldo, lse, delta, (T*)dk, (T*)dv, ldd, B, N, Hh, scale)
    if (D == 64) { GO(64); } else if (D == 32) { GO(32); } else if (D == 16) { GO(16); } else return UMI_ERR_UNSUPPORTED;
#undef GO
    return UMI_OK;
}
extern "C" int umi_attn_bwd(const void* q, const void* k, const void* v, int ld, const void* o, const void* dO, int ldo,
                            const float* lse, void* dq, void* dk, void* dv, int ldd, float* delta, int B, int N, int heads,
                            int D, int dtype, umi_stream_t st) {
    if (!q || !k || !v || !o || !dO || !lse || !dq || !dk || !dv || !delta) return UMI_ERR_BADARG;
    if (umi_attn_mfma_ok(D, ld, ldo, dtype, q, k, v) && ldd % 8 == 0 &&
        ((((uintptr_t)o) | ((uintptr_t)dO) | ((uintptr_t)dq) | ((uintptr_t)dk) | ((uintptr_t)dv)) & 15) == 0)
        return umi_attn_bwd_mfma(q, k, v, ld, o, dO, ldo, lse, dq, dk, dv, ldd, delta, B, N, heads, (hipStream_t)st);
    int rc;
    if (dtype == UMI_F32) rc = attn_bwd_t<float>(q, k, v, ld, o, dO, ldo, lse, dq, dk, dv, ldd, delta, B, N, heads, D, (hipStream_t)st);
    else if (dtype == UMI_F16) rc = attn_bwd_t<half_t>(q, k, v, ld, o, dO, ldo, lse, dq, dk, dv, ldd, delta, B, N, heads, D, (hipStream_t)st);
    else return UMI_ERR_BADARG;
    if (rc) return rc;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_bilinear2x(const void* x, int ldx, const void* tx, void* y, int ldy, int backward, int N, int H, int W, int C,
                              int dtype, umi_stream_t st) {
    // forward: x [N,H,W,C] -> y [N,2H,2W,C];  backward: x = dy [N,2H,2W,C] -> y = dx [N,H,W,C]
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return UMI_ERR_BADARG;
    hipStream_t s = (hipStream_t)st;
    if (dtype == UMI_F16 && umi_bilinear2x_f16v(x, ldx, tx, y, ldy, backward, N, H, W, C, s)) {
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    if (!backward) {
        const int grid = grid_for((long)N * 4 * H * W * C);
        DT_SWITCH(dtype,
            hipLaunchKernelGGL(bilinear2x_fwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, (const float4*)tx, (float*)y, ldy, N, H, W, C),
            hipLaunchKernelGGL(bilinear2x_fwd_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)tx, (half_t*)y, ldy, N, H, W, C))
    } else {
        const int grid = grid_for((long)N * H * W * C);
        DT_SWITCH(dtype,
            hipLaunchKernelGGL(bilinear2x_bwd_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, N, H, W, C),
            hipLaunchKernelGGL(bilinear2x_bwd_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, (half_t*)y, ldy, N, H, W, C))
    }
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
