// HBM-bound ends of the network (fp16 storage), where matrix cores cannot help:
//   * stem : the first 3x3 conv, Ci <= 4 (reference Model.py:111 `inc` with n_channels 1 or 3): forward with the
//            BatchNorm-statistics epilogue, and its weight gradient.  Arithmetic intensity ~9 FLOP/B.
//   * head : OutConv 1x1 with n_classes <= 8 outputs (reference Model.py:86-92): forward (fp32 logits, bias,
//            BN+ReLU of the producer applied on load), data gradient (K = n_classes) and weight gradient.
// One thread = one pixel x 8 channels (16-B accesses, 8 lanes per 128-B line); per-channel reductions go
// through registers -> LDS -> one deterministic partial row per workgroup.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

void umi_launch_wgrad_reduce(const float* part, int splits, int RS, int Ci, int Co, float* dW, long s_co, long s_ci,
                             long s_t, float scale, hipStream_t st);

namespace {

constexpr int STEM_PPB = 1024;       // pixels per workgroup (stem forward)
constexpr int WG_PPB = 4096;         // pixels per workgroup (weight-gradient kernels)

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------------------
template <int CI>
__global__ __launch_bounds__(256) void stem3x3_fwd_kernel(const half_t* __restrict__ x, int ldx,
                                                          const float4* __restrict__ tx, const half_t* __restrict__ wp,
                                                          half_t* __restrict__ y, int ldy, float* __restrict__ part,
                                                          int N, int H, int W, int Co) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [9*CI][Co] weights, then reduction scratch
    const int tid = threadIdx.x;
    const int G = Co >> 3, PL = 256 / G;
    const int cg = tid % G, pl = tid / G;
    float* wsm = sm;
    for (int i = tid; i < 9 * CI * Co; i += 256) wsm[i] = (float)wp[i];
    __syncthreads();
    const long P = (long)N * H * W;
    const long p0 = (long)blockIdx.x * STEM_PPB;
    float s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
    // every thread walks a contiguous run of pixels: the (image, row, column) cursor advances without divisions and the
    // 3x3 input window slides (3*CI new values per pixel instead of 9*CI).  (Measured at the bench shape, 0.21 ms as is: dealing the
    // pixels round-robin to the lanes, as the weight-gradient kernel below does for its loads, 0.31 ms; runs of 4 pixels per lane
    // with adjacent runs across lanes 0.29 ms; the 72 weights of a one-channel stem in registers instead of LDS: no change.)
    const int RUN = STEM_PPB / PL;
    long p = p0 + (long)pl * RUN;
    long pend = p + RUN;
    if (pend > P) pend = P;
    if (pend > p0 + STEM_PPB) pend = p0 + STEM_PPB;
    int n = 0, yy = 0, xx = 0;
    if (p < pend) {
        n = (int)(p / ((long)H * W));
        const int r = (int)(p - (long)n * H * W);
        yy = r / W;
        xx = r - yy * W;
    }
    float win[3][3][CI];
    bool fresh = true;
    for (; p < pend; ++p) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int hi = yy + dy - 1;
            const bool rowok = hi >= 0 && hi < H;
            const half_t* xr = x + ((long)((long)n * H + (rowok ? hi : 0)) * W) * ldx;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if (!fresh && dx < 2) {
#pragma unroll
                    for (int ci = 0; ci < CI; ++ci) win[dy][dx][ci] = win[dy][dx + 1][ci];
                    continue;
                }
                const int wi = xx + dx - 1;
                const bool ok = rowok && wi >= 0 && wi < W;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) {
                    float v = 0.f;
                    if (ok) {
                        v = (float)xr[(long)wi * ldx + ci];
                        if (tx) v = umi_tx(v, tx[ci]);
                    }
                    win[dy][dx][ci] = v;
                }
            }
        }
        fresh = false;
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ci = 0; ci < CI; ++ci) {
                const float v = win[tap / 3][tap % 3][ci];
                const float4* wrow = reinterpret_cast<const float4*>(wsm + (tap * CI + ci) * Co + cg * 8);
                float4 w0 = wrow[0], w1 = wrow[1];
                acc[0] = fmaf(v, w0.x, acc[0]); acc[1] = fmaf(v, w0.y, acc[1]);
                acc[2] = fmaf(v, w0.z, acc[2]); acc[3] = fmaf(v, w0.w, acc[3]);
                acc[4] = fmaf(v, w1.x, acc[4]); acc[5] = fmaf(v, w1.y, acc[5]);
                acc[6] = fmaf(v, w1.z, acc[6]); acc[7] = fmaf(v, w1.w, acc[7]);
            }
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o[j] = (half_t)acc[j];
            float vr = (float)o[j];
            s[j] += vr;
            q[j] = fmaf(vr, vr, q[j]);
        }
        *reinterpret_cast<half8*>(y + p * ldy + cg * 8) = o;
        if (++xx == W) {
            xx = 0;
            fresh = true;
            if (++yy == H) { yy = 0; ++n; }
        }
    }
    if (part) {
        __syncthreads();
        float* red = sm;                                   // reuse: [2][256][9]
#pragma unroll
        for (int j = 0; j < 8; ++j) { red[(0 * 256 + tid) * 9 + j] = s[j]; red[(1 * 256 + tid) * 9 + j] = q[j]; }
        __syncthreads();
        for (int i = tid; i < 2 * Co; i += 256) {
            int which = i / Co, c = i - which * Co;
            float a = 0.f;
            for (int k = 0; k < PL; ++k) a += red[(which * 256 + k * G + (c >> 3)) * 9 + (c & 7)];
            part[((long)blockIdx.x * 2 + which) * Co + c] = a;
        }
    }
}

// BNA: `dy` is the gradient of the ACTIVATED output and the BatchNorm(+ReLU) backward's stage 3 -- dz = gamma*rstd*(relu'(z)*dA - c1 -
// xhat*c2), umi_bn_dz (common.h), rounded to fp16 exactly as bn_bwd_apply_v8 stores it -- is formed on the fly from (dA, y).  The stem's
// input needs no gradient, so nothing else reads dz: the apply pass over the layer's 537 MB tensors (read dA, read y, write dz) and
// this kernel's read of dz become one read of dA and y.
struct StemBna { const half_t* y; int ldy; const float4* tx; const float* rstd; const float* sum_dz; const float* sum_dzx; long M; };

// dW[co][ci][tap] partials: grid = (pixel blocks, Ci); thread = (pixel lane, 8 output channels)
template <bool BNA>
__global__ __launch_bounds__(256) void stem3x3_wgrad_kernel(const half_t* __restrict__ x, int ldx,
                                                            const float4* __restrict__ tx,
                                                            const half_t* __restrict__ dy, int lddy,
                                                            float* __restrict__ part, int N, int H, int W, int Ci,
                                                            int Co, StemBna ba) {
    __shared__ float red[256][9];
    const int tid = threadIdx.x, ci = blockIdx.y;
    const int G = Co >> 3, PL = 256 / G;
    const int cg = tid % G, pl = tid / G;
    const long P = (long)N * H * W;
    const long p0 = (long)blockIdx.x * WG_PPB;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    float4 tc = tx ? tx[ci] : make_float4(0.f, 1.f, 0.f, -INFINITY);
    float4 tb[BNA ? 8 : 1];
    float rsb[BNA ? 8 : 1], c1[BNA ? 8 : 1], c2[BNA ? 8 : 1];
    if (BNA) {
        const float invM = 1.f / (float)ba.M;            // on the device, as the standalone kernels compute it
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            tb[j] = ba.tx[c];
            rsb[j] = ba.rstd[c];
            c1[j] = ba.sum_dz[c] * invM;
            c2[j] = ba.sum_dzx[c] * invM;
        }
    }
    // Pixels are dealt to the PL pixel lanes round-robin: per trip a workgroup reads PL ADJACENT pixels of dy (PL x 128 B contiguous
    // at Co = 64).  (Round 1 gave every lane its own contiguous run -- a sliding 3x3 window, three input loads per pixel instead
    // of nine -- which made 32 K concurrent 128-byte streams of the 537 MB gradient tensor: 1.5 TB/s.  The nine input values
    // per pixel come from a one-channel tensor that lives in the caches.)
    long p = p0 + pl;
    long pend = p0 + WG_PPB;
    if (pend > P) pend = P;
    int n = 0, yy = 0, xx = 0;
    if (p < pend) {
        n = (int)(p / ((long)H * W));
        const int r = (int)(p - (long)n * H * W);
        yy = r / W;
        xx = r - yy * W;
    }
#pragma unroll 4
    for (; p < pend; p += PL) {
        const half8 g = *reinterpret_cast<const half8*>(dy + p * lddy + cg * 8);
        float win[9];
#pragma unroll
        for (int dy_ = 0; dy_ < 3; ++dy_) {
            const int hi = yy + dy_ - 1;
            const bool rowok = hi >= 0 && hi < H;
            const half_t* xr = x + ((long)((long)n * H + (rowok ? hi : 0)) * W) * ldx + ci;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int wi = xx + dx - 1;
                const bool ok = rowok && wi >= 0 && wi < W;
                float v = (float)xr[(long)(ok ? wi : 0) * ldx];
                if (tx) v = umi_tx(v, tc);
                win[dy_ * 3 + dx] = ok ? v : 0.f;
            }
        }
        float gf[8];
        if (BNA) {
            const half8 yv = *reinterpret_cast<const half8*>(ba.y + p * ba.ldy + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) gf[j] = (float)umi_bn_dz<_Float16>((float)yv[j], (float)g[j], tb[j], rsb[j], c1[j], c2[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) gf[j] = (float)g[j];
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[tap][j] = fmaf(win[tap], gf[j], acc[tap][j]);
        xx += PL;
        while (xx >= W) {
            xx -= W;
            if (++yy == H) { yy = 0; ++n; }
        }
    }
    for (int tap = 0; tap < 9; ++tap) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid][j] = acc[tap][j];
        __syncthreads();
        if (tid < Co) {
            float a = 0.f;
            for (int k = 0; k < PL; ++k) a += red[k * G + (tid >> 3)][tid & 7];
            part[(((long)blockIdx.x * 9 + tap) * Ci + ci) * Co + tid] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// head forward: y[p][k] = bias[k] + sum_c tx(x[p][c]) * w[c][k]   (NHWC, ldy = ncls or more)
// TO = float: the segmentation logits.  TO = half_t: a narrow pointwise conv inside the network (the attention gate's 1-channel
// psi branch, reference Model.py:283-287); STATS then also emits the BatchNorm partial sums part[block][2][NC] of the ROUNDED
// outputs, like every other conv epilogue of the library.
template <int NC, typename TO, bool STATS>
__global__ __launch_bounds__(256) void head1x1_fwd_kernel(const half_t* __restrict__ x, int ldx,
                                                          const float4* __restrict__ tx, const half_t* __restrict__ wp,
                                                          const float* __restrict__ bias, TO* __restrict__ y, int ldy,
                                                          float* __restrict__ part, long P, int C) {
    const int G = C >> 3;                                   // lanes per pixel (power of two <= 64)
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gt % G);
    const long stride_p = ((long)gridDim.x * 256) / G;
    float4 t[8];
    float w[8][NC];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        t[j] = tx ? tx[cg * 8 + j] : make_float4(0.f, 1.f, 0.f, -INFINITY);
#pragma unroll
        for (int k = 0; k < NC; ++k) w[j][k] = (float)wp[(cg * 8 + j) * NC + k];
    }
    float ssum[NC], ssq[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) ssum[k] = ssq[k] = 0.f;
    const long Pr = ((P + stride_p - 1) / stride_p) * stride_p;      // keep whole pixel groups in the shuffle
    for (long p = gt / G; p < Pr; p += stride_p) {
        float acc[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) acc[k] = 0.f;
        if (p < P) {
            half8 v = *reinterpret_cast<const half8*>(x + p * ldx + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = umi_tx((float)v[j], t[j]);
#pragma unroll
                for (int k = 0; k < NC; ++k) acc[k] = fmaf(a, w[j][k], acc[k]);
            }
        }
        for (int o = G >> 1; o > 0; o >>= 1)
#pragma unroll
            for (int k = 0; k < NC; ++k) acc[k] += __shfl_xor(acc[k], o);
        if (cg == 0 && p < P) {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                const TO o = (TO)(acc[k] + (bias ? bias[k] : 0.f));
                y[p * ldy + k] = o;
                if (STATS) {
                    const float vr = (float)o;
                    ssum[k] += vr;
                    ssq[k] = fmaf(vr, vr, ssq[k]);
                }
            }
        }
    }
    if (STATS) {
        __shared__ float red[4][2][NC];
#pragma unroll
        for (int k = 0; k < NC; ++k)
            for (int o = 32; o > 0; o >>= 1) {
                ssum[k] += __shfl_xor(ssum[k], o);
                ssq[k] += __shfl_xor(ssq[k], o);
            }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                red[threadIdx.x >> 6][0][k] = ssum[k];
                red[threadIdx.x >> 6][1][k] = ssq[k];
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * NC) {
            const int which = threadIdx.x / NC, k = threadIdx.x - which * NC;
            part[((long)blockIdx.x * 2 + which) * NC + k] =
                (red[0][which][k] + red[1][which][k]) + (red[2][which][k] + red[3][which][k]);
        }
    }
}

// head data gradient: da[p][c] = sum_k dl[p][k] * w[k][c]    (K = NC tiny; wp = [NC][C])
template <int NC>
__global__ __launch_bounds__(256) void smallk1x1_fwd_kernel(const half_t* __restrict__ x, int ldx,
                                                            const half_t* __restrict__ wp, half_t* __restrict__ y,
                                                            int ldy, long P, int C) {
    const int G = C >> 3;
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gt % G);
    const long stride_p = ((long)gridDim.x * 256) / G;
    float w[NC][8];
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) w[k][j] = (float)wp[k * C + cg * 8 + j];
    for (long p = gt / G; p < P; p += stride_p) {
        float d[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) d[k] = (float)x[p * ldx + k];
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) a = fmaf(d[k], w[k][j], a);
            o[j] = (half_t)a;
        }
        *reinterpret_cast<half8*>(y + p * ldy + cg * 8) = o;
    }
}

// The same data gradient + stage 1 of the BatchNorm(+ReLU) backward of the layer whose activated output `da` is the gradient of
// (ybn / txbn / rstd = that layer's raw output, transform rows and 1/std): part[block][2][C] <- sums of dz and dz * xhat over the
// block's pixels, dz = stored (rounded) da * [tx(y) > lo] -- what bn_bwd_reduce1_v8 computes in a pass of its own over da and y.
// wpart != nullptr: also the head's WEIGHT gradient partials wpart[block][C][NC] = sum_p tx(ybn[p][c]) * x[p][k] -- the head's input is the
// activated ybn, which this kernel reads anyway (head1x1_wgrad_kernel's pass over the 537 MB tensor goes away).
template <int NC>
__global__ __launch_bounds__(256) void smallk1x1_bnred_kernel(const half_t* __restrict__ x, int ldx,
                                                              const half_t* __restrict__ wp, half_t* __restrict__ y,
                                                              int ldy, const half_t* __restrict__ ybn, int ldybn,
                                                              const float4* __restrict__ txbn, const float* __restrict__ rstd,
                                                              float* __restrict__ part, long P, int C, float* __restrict__ wpart) {
    __shared__ float red[2][256][9];
    const int G = C >> 3;
    const long gt = (long)blockIdx.x * 256 + threadIdx.x;
    const int cg = (int)(gt % G);                            // (256 % G == 0: the same for every trip of this thread)
    const long stride_p = ((long)gridDim.x * 256) / G;
    float w[NC][8], wacc[NC][8];
    float4 t[8];
    float rs[8], s[8], q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        t[j] = txbn[cg * 8 + j];
        rs[j] = rstd[cg * 8 + j];
        s[j] = q[j] = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) wacc[k][j] = 0.f;
#pragma unroll
        for (int k = 0; k < NC; ++k) w[k][j] = (float)wp[k * C + cg * 8 + j];
    }
    for (long p = gt / G; p < P; p += stride_p) {
        const half8 yv = *reinterpret_cast<const half8*>(ybn + p * ldybn + cg * 8);
        float d[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) d[k] = (float)x[p * ldx + k];
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < NC; ++k) a = fmaf(d[k], w[k][j], a);
            o[j] = (half_t)a;
            const float yy = (float)yv[j];
            const float dz = umi_tx_pre(yy, t[j]) > t[j].w ? (float)o[j] : 0.f;
            s[j] += dz;
            q[j] = fmaf(dz, (yy - t[j].x) * rs[j], q[j]);
            if (wpart) {
                const float av = umi_tx(yy, t[j]);
#pragma unroll
                for (int k = 0; k < NC; ++k) wacc[k][j] = fmaf(av, d[k], wacc[k][j]);
            }
        }
        *reinterpret_cast<half8*>(y + p * ldy + cg * 8) = o;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][threadIdx.x][j] = s[j]; red[1][threadIdx.x][j] = q[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        float a = 0.f;
        for (int k = 0; k < 256 / G; ++k) a += red[which][k * G + (c >> 3)][c & 7];
        part[((long)blockIdx.x * 2 + which) * C + c] = a;
    }
    if (wpart) {
        for (int k = 0; k < NC; ++k) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) red[0][threadIdx.x][j] = wacc[k][j];
            __syncthreads();
            if ((int)threadIdx.x < C) {
                float a = 0.f;
                for (int qq = 0; qq < 256 / G; ++qq) a += red[0][qq * G + (threadIdx.x >> 3)][threadIdx.x & 7];
                wpart[((long)blockIdx.x * C + threadIdx.x) * NC + k] = a;
            }
        }
    }
}

// head weight gradient partials: part[blk][0][c][k] = sum_p tx(x[p][c]) * dl[p][k]   (layout [z][tap=0][ci][co])
template <int NC>
__global__ __launch_bounds__(256) void head1x1_wgrad_kernel(const half_t* __restrict__ x, int ldx,
                                                            const float4* __restrict__ tx,
                                                            const half_t* __restrict__ dl, int lddl,
                                                            float* __restrict__ part, long P, int C, int ppb) {
    __shared__ float red[256][9];
    const int tid = threadIdx.x;
    const int G = C >> 3, PL = 256 / G;
    const int cg = tid % G, pl = tid / G;
    const long p0 = (long)blockIdx.x * ppb;
    float4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = tx ? tx[cg * 8 + j] : make_float4(0.f, 1.f, 0.f, -INFINITY);
    float acc[NC][8];
#pragma unroll
    for (int k = 0; k < NC; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
    for (long p = p0 + pl; p < p0 + ppb && p < P; p += PL) {
        half8 v = *reinterpret_cast<const half8*>(x + p * ldx + cg * 8);
        float a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = umi_tx((float)v[j], t[j]);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            float d = (float)dl[p * lddl + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(a[j], d, acc[k][j]);
        }
    }
    for (int k = 0; k < NC; ++k) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) red[tid][j] = acc[k][j];
        __syncthreads();
        if (tid < C) {
            float s = 0.f;
            for (int q = 0; q < PL; ++q) s += red[q * G + (tid >> 3)][tid & 7];
            part[((long)blockIdx.x * C + tid) * NC + k] = s;
        }
    }
}

// wpart[rows][C][NC] -> dW[k * s_co + c * s_ci]: one workgroup per element, fp64 accumulation (thousands of rows x 128 elements: the
// generic split-K reduction, 4 workgroups of 8 split lanes there, took 365 us for it)
__global__ __launch_bounds__(256) void head_wpart_reduce_kernel(const float* __restrict__ wpart, int rows, int C, int NC,
                                                                float* __restrict__ dW, long s_co, long s_ci, float scale) {
    __shared__ double sh[256];
    const int e = blockIdx.x, tid = threadIdx.x;
    double a = 0.0;
    for (int r = tid; r < rows; r += 256) a += (double)wpart[(long)r * C * NC + e];
    sh[tid] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    if (tid == 0) dW[(e % NC) * s_co + (e / NC) * s_ci] = (float)(sh[0] * (double)scale);
}

int grid_for(long items) {
    long g = (items + 255) / 256;
    if (g > 8192) g = 8192;
    return g < 1 ? 1 : (int)g;
}

bool groups_ok(int C) { return C % 8 == 0 && C / 8 <= 64 && (256 % (C / 8)) == 0; }

}  // namespace

// ---- dispatch helpers used by api.hip -------------------------------------------------------------------------
bool umi_stem_fwd_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldy, int in_dtype, int out_dtype, int flags,
                     const float* bias) {
    if (flags & (UMI_CONV_UPSAMPLE2 | UMI_CONV_FORCE_GENERIC)) return false;
    return in_dtype == UMI_F16 && out_dtype == UMI_F16 && !bias && R == 3 && S == 3 && stride == 1 && pad == 1 &&
           Ci >= 1 && Ci <= 4 && groups_ok(Co) && ldy % 8 == 0 && 9 * Ci * Co * 4 <= 48 * 1024;
}
int umi_stem_stat_rows(int N, int H, int W) { return (int)(((long)N * H * W + STEM_PPB - 1) / STEM_PPB); }

int umi_stem_fwd(const void* x, int ldx, const void* tx, const void* wp, void* y, int ldy, float* part, int N, int H, int W,
                 int Ci, int Co, hipStream_t s) {
    if (!al16(y)) return UMI_ERR_BADARG;
    const int rows = umi_stem_stat_rows(N, H, W);
    size_t smem = (size_t)9 * Ci * Co * 4;
    if (smem < 2 * 256 * 9 * 4) smem = 2 * 256 * 9 * 4;
#define GO(CI) hipLaunchKernelGGL(stem3x3_fwd_kernel<CI>, dim3(rows), dim3(256), smem, s, (const half_t*)x, ldx, (const float4*)tx, (const half_t*)wp, (half_t*)y, ldy, part, N, H, W, Co)
    switch (Ci) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; default: GO(4); }
#undef GO
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

bool umi_stem_wgrad_ok(int Ci, int Co, int R, int S, int stride, int pad, int lddy, int dtype, int flags, const void* txb) {
    if (flags & UMI_CONV_FORCE_GENERIC) return false;
    return dtype == UMI_F16 && !txb && R == 3 && S == 3 && stride == 1 && pad == 1 && Ci >= 1 && Ci <= 4 && groups_ok(Co) &&
           Co <= 256 && lddy % 8 == 0;
}
size_t umi_stem_wgrad_ws_bytes(int N, int H, int W, int Ci, int Co) {
    long blocks = ((long)N * H * W + WG_PPB - 1) / WG_PPB;
    return (size_t)blocks * 9 * Ci * Co * sizeof(float);
}
int umi_stem_wgrad(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co, long s_ci,
                   long s_t, float out_scale, int N, int H, int W, int Ci, int Co, void* ws, size_t ws_bytes, hipStream_t s) {
    if (ws_bytes < umi_stem_wgrad_ws_bytes(N, H, W, Ci, Co)) return UMI_ERR_WORKSPACE;
    if (!al16(dy)) return UMI_ERR_BADARG;
    int blocks = (int)(((long)N * H * W + WG_PPB - 1) / WG_PPB);
    hipLaunchKernelGGL(stem3x3_wgrad_kernel<false>, dim3(blocks, Ci), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)txa,
                       (const half_t*)dy, lddy, (float*)ws, N, H, W, Ci, Co, StemBna{});
    UMI_LAUNCH_CHECK();
    umi_launch_wgrad_reduce((const float*)ws, blocks, 9, Ci, Co, dW, s_co, s_ci, s_t, out_scale, s);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
// the same with stage 3 of the following BatchNorm's backward formed on the fly (da = gradient of the activated output)
int umi_stem_wgrad_bnapply(const void* x, int ldx, const void* txa, const void* da, int ldda, const void* y, int ldy,
                           const void* tx_bn, const float* rstd, const float* sum_dz, const float* sum_dzx, float* dW, long s_co,
                           long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci, int Co, void* ws, size_t ws_bytes,
                           hipStream_t s) {
    if (ws_bytes < umi_stem_wgrad_ws_bytes(N, H, W, Ci, Co)) return UMI_ERR_WORKSPACE;
    if (!al16(da) || !al16(y)) return UMI_ERR_BADARG;
    int blocks = (int)(((long)N * H * W + WG_PPB - 1) / WG_PPB);
    const StemBna ba{(const half_t*)y, ldy, (const float4*)tx_bn, rstd, sum_dz, sum_dzx, (long)N * H * W};
    hipLaunchKernelGGL(stem3x3_wgrad_kernel<true>, dim3(blocks, Ci), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)txa,
                       (const half_t*)da, ldda, (float*)ws, N, H, W, Ci, Co, ba);
    UMI_LAUNCH_CHECK();
    umi_launch_wgrad_reduce((const float*)ws, blocks, 9, Ci, Co, dW, s_co, s_ci, s_t, out_scale, s);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// head forward: fp16 in, 1x1, Co <= 8 (weights in generic [1][Ci][Co] fp16 packing); fp32 out (logits) or fp16 out (+ statistics)
bool umi_head_fwd_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldx, int in_dtype, int out_dtype, int flags) {
    if (flags & (UMI_CONV_UPSAMPLE2 | UMI_CONV_FORCE_GENERIC)) return false;
    return in_dtype == UMI_F16 && (out_dtype == UMI_F32 || out_dtype == UMI_F16) && R == 1 && S == 1 && stride == 1 &&
           pad == 0 && Co >= 1 && Co <= 8 && groups_ok(Ci) && ldx % 8 == 0;
}
int umi_head_stat_rows(long P, int Ci) { return grid_for(P * (Ci / 8)); }
int umi_head_fwd(const void* x, int ldx, const void* tx, const void* wp, const float* bias, void* y, int ldy, float* part,
                 long P, int Ci, int Co, int out_dtype, hipStream_t s) {
    if (!al16(x)) return UMI_ERR_BADARG;
    if (part && out_dtype != UMI_F16) return UMI_ERR_UNSUPPORTED;
    int grid = grid_for(P * (Ci / 8));
#define GO_(NC, TO, ST) hipLaunchKernelGGL((head1x1_fwd_kernel<NC, TO, ST>), dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)tx, (const half_t*)wp, bias, (TO*)y, ldy, part, P, Ci)
#define GO(NC)                                                          \
    do {                                                                \
        if (out_dtype == UMI_F32) GO_(NC, float, false);                \
        else if (part) GO_(NC, half_t, true);                           \
        else GO_(NC, half_t, false);                                    \
    } while (0)
    switch (Co) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
                  case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break; default: GO(8); }
#undef GO
#undef GO_
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// tiny-K pointwise conv (the head's data gradient): fp16 -> fp16, Ci <= 8, no transform / bias
bool umi_smallk_fwd_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldy, int in_dtype, int out_dtype, int flags,
                       const void* tx, const float* bias) {
    if (flags & (UMI_CONV_UPSAMPLE2 | UMI_CONV_FORCE_GENERIC)) return false;
    return in_dtype == UMI_F16 && out_dtype == UMI_F16 && !tx && !bias && R == 1 && S == 1 && stride == 1 && pad == 0 &&
           Ci >= 1 && Ci <= 8 && groups_ok(Co) && ldy % 8 == 0;
}
int umi_smallk_fwd(const void* x, int ldx, const void* wp, void* y, int ldy, long P, int Ci, int Co, hipStream_t s) {
    if (!al16(y)) return UMI_ERR_BADARG;
    int grid = grid_for(P * (Co / 8));
#define GO(NC) hipLaunchKernelGGL(smallk1x1_fwd_kernel<NC>, dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, (const half_t*)wp, (half_t*)y, ldy, P, Co)
    switch (Ci) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
                  case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break; default: GO(8); }
#undef GO
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

int umi_smallk_bnred_rows(long P, int Co) { return grid_for(P * (Co / 8)); }
// dW != NULL: also the head's weight gradient dW[k * s_co + c * s_ci] = out_scale * sum_p tx(ybn[p][c]) * x[p][k]; ws >= rows * Co * Ci floats
int umi_smallk_fwd_bnred(const void* x, int ldx, const void* wp, void* y, int ldy, const void* ybn, int ldybn, const void* txbn,
                         const float* rstd, float* part, long P, int Ci, int Co, hipStream_t s, float* dW, long s_co, long s_ci,
                         float out_scale, void* ws, size_t ws_bytes) {
    if (!al16(y) || !al16(ybn)) return UMI_ERR_BADARG;
    int grid = grid_for(P * (Co / 8));
    float* wpart = nullptr;
    if (dW) {
        if (!ws || ws_bytes < (size_t)grid * Co * Ci * sizeof(float) || Co > 256) return UMI_ERR_WORKSPACE;
        wpart = (float*)ws;
    }
#define GO(NC) hipLaunchKernelGGL(smallk1x1_bnred_kernel<NC>, dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, (const half_t*)wp, (half_t*)y, ldy, (const half_t*)ybn, ldybn, (const float4*)txbn, rstd, part, P, Co, wpart)
    switch (Ci) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
                  case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break; default: GO(8); }
#undef GO
    UMI_LAUNCH_CHECK();
    if (dW) {      // (kernel roles: the head conv's input channels = this kernel's Co, its output channels = Ci)
        hipLaunchKernelGGL(head_wpart_reduce_kernel, dim3(Co * Ci), dim3(256), 0, s, (const float*)wpart, grid, Co, Ci, dW, s_co, s_ci,
                           out_scale);
        UMI_LAUNCH_CHECK();
    }
    return UMI_OK;
}

bool umi_head_wgrad_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldx, int dtype, int flags, const void* txb) {
    if (flags & UMI_CONV_FORCE_GENERIC) return false;
    return dtype == UMI_F16 && !txb && R == 1 && S == 1 && stride == 1 && pad == 0 && Co >= 1 && Co <= 8 && groups_ok(Ci) &&
           Ci <= 256 && ldx % 8 == 0;
}
// pixels per workgroup of the head weight gradient: ~2048 workgroups where the tensor allows it (a fixed 4096 left the 64x64
// maps of the attention gates with 16 workgroups: 249 us for 34 MB), whole multiples of 256 pixels
static int head_wgrad_ppb(long P) {
    long ppb = (P + 2047) / 2048;
    ppb = ((ppb + 255) / 256) * 256;
    if (ppb > WG_PPB) ppb = WG_PPB;
    return (int)ppb;
}
size_t umi_head_wgrad_ws_bytes(long P, int Ci, int Co) {
    const int ppb = head_wgrad_ppb(P);
    return (size_t)((P + ppb - 1) / ppb) * Ci * Co * sizeof(float);
}
int umi_head_wgrad(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co, long s_ci,
                   long s_t, float out_scale, long P, int Ci, int Co, void* ws, size_t ws_bytes, hipStream_t s) {
    if (ws_bytes < umi_head_wgrad_ws_bytes(P, Ci, Co)) return UMI_ERR_WORKSPACE;
    if (!al16(x)) return UMI_ERR_BADARG;
    const int ppb = head_wgrad_ppb(P);
    int blocks = (int)((P + ppb - 1) / ppb);
#define GO(NC) hipLaunchKernelGGL(head1x1_wgrad_kernel<NC>, dim3(blocks), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)txa, (const half_t*)dy, lddy, (float*)ws, P, Ci, ppb)
    switch (Co) { case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break; case 4: GO(4); break;
                  case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break; default: GO(8); }
#undef GO
    UMI_LAUNCH_CHECK();
    umi_launch_wgrad_reduce((const float*)ws, blocks, 1, Ci, Co, dW, s_co, s_ci, s_t, out_scale, s);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
