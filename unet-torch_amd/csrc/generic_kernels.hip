// Generic (any shape / stride / channel count) HIP kernels of libunetmi: LDS-tiled fp32-accumulate
// VALU kernels templated on the storage type.  They are the fp32 parity path and the
// fallback for shapes the MFMA kernels do not take; written for gfx950 (64-wide waves,
// 256-thread workgroups = 4 waves, float4 LDS reads).
#include "common.h"

// elementwise_f16.hip: 16-B vectorised fp16 fast paths (return false when the shape does not qualify)
int umi_bn_bwd_rpb_f16v(long M);
int umi_colsum_rows_f16v(long M, int C);
bool umi_colsum_f16v(const void* x, int ldx, float* ws, long M, int C, hipStream_t s);
bool umi_bn_bwd_reduce1_f16v(const void* da, int ldda, const void* y, int ldy, const void* tx, const float* rstd, float* ws,
                             long M, int C, hipStream_t s);
bool umi_bn_bwd_apply_f16v(void* da, int ldda, const void* y, int ldy, const void* tx, const float* rstd,
                           const float* sum_dz, const float* sum_dzx, long M, int C, hipStream_t s);
bool umi_pool2_fwd_f16v(const void* x, int ldx, const void* tx, void* y, int ldy, int N, int H, int W, int C,
                        hipStream_t s);
bool umi_pool2_bwd_f16v(const void* dp, int lddp, const void* x, int ldx, const void* tx, void* da, int ldda,
                        int accumulate, int N, int H, int W, int C, hipStream_t s);

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
template <typename T, bool K8>
__global__ void pack_kn_kernel(const float* __restrict__ src, T* __restrict__ dst, int Tn, int K, int N,
                               long st, long sk, long sn, int flip_t, int Kpad, int Npad) {
    long total = (long)Tn * Kpad * Npad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int n, k, t;
        if (K8) {       // dst[t][k/8][n][k%8]
            int k8 = (int)(i & 7);
            long r = i >> 3;
            n = (int)(r % Npad); r /= Npad;
            int kb = (int)(r % (Kpad / 8));
            t = (int)(r / (Kpad / 8));
            k = kb * 8 + k8;
        } else {        // dst[t][k][n]
            n = (int)(i % Npad);
            long r = i / Npad;
            k = (int)(r % Kpad);
            t = (int)(r / Kpad);
        }
        float v = 0.f;
        if (k < K && n < N) {
            int ts = flip_t ? (Tn - 1 - t) : t;
            v = src[ts * st + k * sk + n * sn];
        }
        dst[i] = (T)v;
    }
}

static int pack_impl(const float* src, void* dst, int T, int K, int N, long st, long sk, long sn, int flip_t,
                     int Kpad, int Npad, int dtype, bool k8, hipStream_t s) {
    if (T <= 0 || K <= 0 || N <= 0 || Kpad < K || Npad < N) return UMI_ERR_BADARG;
    if (k8 && (Kpad % 8)) return UMI_ERR_BADARG;
    long total = (long)T * Kpad * Npad;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (dtype == UMI_F32) {
        if (k8) hipLaunchKernelGGL((pack_kn_kernel<float, true>), dim3(grid), dim3(256), 0, s, src, (float*)dst, T, K, N, st, sk, sn, flip_t, Kpad, Npad);
        else    hipLaunchKernelGGL((pack_kn_kernel<float, false>), dim3(grid), dim3(256), 0, s, src, (float*)dst, T, K, N, st, sk, sn, flip_t, Kpad, Npad);
    } else if (dtype == UMI_F16) {
        if (k8) hipLaunchKernelGGL((pack_kn_kernel<half_t, true>), dim3(grid), dim3(256), 0, s, src, (half_t*)dst, T, K, N, st, sk, sn, flip_t, Kpad, Npad);
        else    hipLaunchKernelGGL((pack_kn_kernel<half_t, false>), dim3(grid), dim3(256), 0, s, src, (half_t*)dst, T, K, N, st, sk, sn, flip_t, Kpad, Npad);
    } else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_pack_kn(const float* src, void* dst, int T, int K, int N, long st, long sk, long sn,
                           int flip_t, int Kpad, int Npad, int dtype, umi_stream_t stream) {
    return pack_impl(src, dst, T, K, N, st, sk, sn, flip_t, Kpad, Npad, dtype, false, (hipStream_t)stream);
}
extern "C" int umi_pack_kn8(const float* src, void* dst, int T, int K, int N, long st, long sk, long sn,
                            int flip_t, int Kpad, int Npad, int dtype, umi_stream_t stream) {
    return pack_impl(src, dst, T, K, N, st, sk, sn, flip_t, Kpad, Npad, dtype, true, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------
// generic convolution forward (implicit GEMM: 64 output pixels x 64 output channels per block)
// ------------------------------------------------------------------------------------------
constexpr int GBP = 64, GBC = 64, GBK = 16;

template <typename TI, typename TO, bool UPS, bool DGS = false>
__global__ __launch_bounds__(256) void conv_generic_kernel(
    const TI* __restrict__ x, int ldx, const float4* __restrict__ tx, const TI* __restrict__ wp,
    const float* __restrict__ bias, TO* __restrict__ y, int ldy, float* __restrict__ part,
    int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo,
    int off_h, int off_w, int out_H, int out_W) {
    __shared__ __attribute__((aligned(16))) float As[GBK][GBP + 4];
    __shared__ __attribute__((aligned(16))) float Bs[GBK][GBC + 4];
    __shared__ float red[2][16][GBC];

    const int tid = threadIdx.x;
    const int txi = tid & 15, tyi = tid >> 4;
    const long P = (long)N * Ho * Wo;
    const long p0 = (long)blockIdx.x * GBP;
    const int c0 = blockIdx.y * GBC;

    // A-tile loader: this thread stages 4 consecutive ci of one pixel
    const int lp = tid >> 2, lk = (tid & 3) * 4;
    const long pg = p0 + lp;
    const bool pv = pg < P;
    int n = 0, ho = 0, wo = 0;
    if (pv) {
        n = (int)(pg / ((long)Ho * Wo));
        int rem = (int)(pg - (long)n * Ho * Wo);
        ho = rem / Wo;
        wo = rem - ho * Wo;
    }
    // B-tile loader: 4 consecutive co of one k
    const int bk = tid >> 4, bc = (tid & 15) * 4;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    const int tap_lo = UPS ? blockIdx.z : 0;
    const int tap_hi = UPS ? blockIdx.z + 1 : R * S;
    for (int tap = tap_lo; tap < tap_hi; ++tap) {
        int hi, wi;
        bool ok = true;
        if (UPS) { hi = ho; wi = wo; }
        else if (DGS) {
            // data gradient of a strided conv: "output" pixel (ho,wo) is an INPUT-image pixel; it receives
            // dy[(ho+pad-r)/stride, (wo+pad-s)/stride] through tap (r,s) when the division is exact
            int r = tap / S, s = tap - r * S;
            int a = ho + pad - r, b = wo + pad - s;
            ok = a >= 0 && b >= 0 && (a % stride) == 0 && (b % stride) == 0;
            hi = a / stride;
            wi = b / stride;
        } else {
            int r = tap / S, s = tap - r * S;
            hi = ho * stride - pad + r;
            wi = wo * stride - pad + s;
        }
        const bool inb = pv && ok && hi >= 0 && hi < H && wi >= 0 && wi < W;
        const TI* xp = x + ((long)((long)n * H + hi) * W + wi) * ldx;
        for (int k0 = 0; k0 < Ci; k0 += GBK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int ci = k0 + lk + j;
                float v = 0.f;
                if (inb && ci < Ci) {
                    v = (float)xp[ci];
                    if (tx) v = umi_tx(v, tx[ci]);
                }
                As[lk + j][lp] = v;
            }
            {
                int ci = k0 + bk;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int co = c0 + bc + j;
                    float v = 0.f;
                    if (ci < Ci && co < Co) v = (float)wp[((long)tap * Ci + ci) * Co + co];
                    Bs[bk][bc + j] = v;
                }
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < GBK; ++kk) {
                float4 a = *(const float4*)&As[kk][tyi * 4];
                float4 b = *(const float4*)&Bs[kk][txi * 4];
                float av[4] = {a.x, a.y, a.z, a.w};
                float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
            }
            __syncthreads();
        }
    }

    // epilogue: this thread owns pixels p0 + tyi*4 + i, channels c0 + txi*4 + j
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        long pq = p0 + tyi * 4 + i;
        if (pq >= P) continue;
        int n2 = (int)(pq / ((long)Ho * Wo));
        int rem = (int)(pq - (long)n2 * Ho * Wo);
        int h2 = rem / Wo, w2 = rem - h2 * Wo;
        if (UPS) {
            h2 = 2 * h2 + (blockIdx.z >> 1) + off_h;
            w2 = 2 * w2 + (blockIdx.z & 1) + off_w;
            if (h2 < 0 || h2 >= out_H || w2 < 0 || w2 >= out_W) continue;
        }
        TO* yp = y + ((long)((long)n2 * out_H + h2) * out_W + w2) * ldy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int co = c0 + txi * 4 + j;
            if (co < Co) {
                float v = acc[i][j] + (bias ? bias[co] : 0.f);
                TO o = (TO)v;
                yp[co] = o;
                float vr = (float)o;
                ssum[j] += vr;
                ssq[j] = fmaf(vr, vr, ssq[j]);
            }
        }
    }
    if (part) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[0][tyi][txi * 4 + j] = ssum[j];
            red[1][tyi][txi * 4 + j] = ssq[j];
        }
        __syncthreads();
        if (tid < 2 * GBC) {
            int which = tid >> 6, c = tid & 63;
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[which][r][c];
            if (c0 + c < Co) part[((long)blockIdx.x * 2 + which) * Co + c0 + c] = s;
        }
    }
}

int umi_conv_fwd_generic(const void* x, int ldx, const void* tx, const void* wp, const float* bias, void* y, int ldy,
                         float* stat_part, int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad,
                         int Ho, int Wo, int off_h, int off_w, int out_H, int out_W, int in_dtype, int out_dtype,
                         int flags, hipStream_t s) {
    const bool ups = flags & UMI_CONV_UPSAMPLE2;
    const bool dgs = flags & UMI_CONV_DGRAD_STRIDED;
    if (ups && (R != 2 || S != 2 || stat_part)) return UMI_ERR_BADARG;
    long P = (long)N * Ho * Wo;
    dim3 grid(umi_cdiv(P, GBP), umi_cdiv(Co, GBC), ups ? 4 : 1), block(256);
    if (dgs) {
        if (ups || stat_part) return UMI_ERR_BADARG;
#define LAUNCHD(TI, TO)                                                                                          \
    hipLaunchKernelGGL((conv_generic_kernel<TI, TO, false, true>), grid, block, 0, s, (const TI*)x, ldx,          \
                       (const float4*)tx, (const TI*)wp, bias, (TO*)y, ldy, stat_part, N, H, W, Ci, Co, R, S,     \
                       stride, pad, Ho, Wo, off_h, off_w, out_H, out_W)
        if (in_dtype == UMI_F32 && out_dtype == UMI_F32) LAUNCHD(float, float);
        else if (in_dtype == UMI_F16 && out_dtype == UMI_F16) LAUNCHD(half_t, half_t);
        else return UMI_ERR_UNSUPPORTED;
#undef LAUNCHD
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
#define LAUNCH(TI, TO, U)                                                                                        \
    hipLaunchKernelGGL((conv_generic_kernel<TI, TO, U>), grid, block, 0, s, (const TI*)x, ldx, (const float4*)tx, \
                       (const TI*)wp, bias, (TO*)y, ldy, stat_part, N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo,   \
                       off_h, off_w, out_H, out_W)
    if (in_dtype == UMI_F32 && out_dtype == UMI_F32) { if (ups) LAUNCH(float, float, true); else LAUNCH(float, float, false); }
    else if (in_dtype == UMI_F16 && out_dtype == UMI_F16) { if (ups) LAUNCH(half_t, half_t, true); else LAUNCH(half_t, half_t, false); }
    else if (in_dtype == UMI_F16 && out_dtype == UMI_F32) { if (ups) LAUNCH(half_t, float, true); else LAUNCH(half_t, float, false); }
    else return UMI_ERR_UNSUPPORTED;
#undef LAUNCH
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// ------------------------------------------------------------------------------------------
// Partial-row reductions part[rows][2][C] -> per-channel totals, two forms:
//   * few rows: one workgroup per channel (column-strided reads; the rows are few, the launch is latency-bound anyway);
//   * many rows (the 512 x 512 / 256 x 256 layers write 4,096-8,192 partial rows = 4 MB: the one-workgroup-per-channel form
//     spent 28 us on it, 64-128 workgroups each pulling a whole line per 4-byte element): a 2-D grid of (32-channel block,
//     64-row slice) workgroups reads whole 128-byte runs, every slice leaves one double per column in a scratch row, and the
//     LAST workgroup of a channel block to finish (a counter per block; agent-scope stores / loads of the scratch rows) adds the slice
//     rows in slice order and finalizes.  The order of every addition is fixed by the indices, never by arrival, so the
//     totals are run-to-run identical (the counter only picks WHO does the final pass).
// The scratch rows and counters are static device memory (nothing is allocated at launch time: safe under graph capture),
// dealt round-robin over RED_REGIONS regions: launches on one stream serialize anyway; up to RED_REGIONS launches may run
// concurrently on different streams.
// ------------------------------------------------------------------------------------------
constexpr int RED_RPS = 128;             // rows per slice (two batches of 16 per wave)
constexpr int RED_MAX_SLICES = 4096;     // (channel block, slice) pairs of one launch: rows x C <= 16 M
constexpr int RED_MAX_CB = 128;          // <= 4,096 channels (the ViT's 3,072-wide fc1 bias gradient)
constexpr int RED_REGIONS = 4;
constexpr int RED_MIN_ROWS = 512;        // below this the one-workgroup-per-channel form is as fast
__device__ double g_red_scratch[RED_REGIONS][RED_MAX_SLICES * 64];       // [region][(cb * nslices + slice) * 64 + lane]: 8 MB
__device__ unsigned g_red_count[RED_REGIONS][RED_MAX_CB];

static bool red2d_ok(int rows, int C) {
    const int nslices = (rows + RED_RPS - 1) / RED_RPS, ncb = (C + 31) / 32;
    return rows >= RED_MIN_ROWS && ncb <= RED_MAX_CB && (long)ncb * nslices <= RED_MAX_SLICES;
}
static int red2d_region() {
    static unsigned turn = 0;
    return (int)(turn++ % RED_REGIONS);
}

// Column sums of a 32-channel block: lanes 0..31 = column 0 (sums) of channels c0 + lane, lanes 32..63 = column 1 of the same
// channels.  Returns true in wave 0 of the workgroup that holds the block's totals (`tot`, valid where c < C).
__device__ __forceinline__ bool red2d_block(const float* __restrict__ part, int rows, int C, bool two, int region, double& tot) {
    __shared__ double sh[4][64];
    __shared__ unsigned is_last;
    const int cb = blockIdx.x, slice = blockIdx.y, nslices = gridDim.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = cb * 32 + (lane & 31), which = lane >> 5;
    const bool live = c < C && (two || which == 0);
    const int csel = c < C ? c : C - 1, wsel = two ? which : 0;
    double a = 0.0;
#pragma unroll
    for (int b = 0; b < RED_RPS / 64; ++b) {               // batches of 16 rows per wave, all 16 loads in flight
        const int r0 = slice * RED_RPS + b * 64 + wave;
        float v[16];
        // (unconditional loads from clamped addresses, all 16 issued before the first is touched, then a select: hipcc turns a
        //  predicated load into a branch with a wait of its own, and sinks a plain one back under the predicate)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = r0 + 4 * i, rc = r < rows ? r : rows - 1;
            v[i] = part[((long)rc * 2 + wsel) * C + csel];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(v[i]));
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = (live && r0 + 4 * i < rows) ? v[i] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += (double)v[i];
    }
    sh[wave][lane] = a;
    __syncthreads();
    a = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
    if (nslices == 1) { tot = a; return wave == 0; }
    double* scr = g_red_scratch[region] + ((long)cb * nslices) * 64;
    if (wave == 0) {
        // The slice row leaves as agent-scope (write-through) stores and is complete -- vmcnt(0) -- before this wave's count:
        // a full release fence here would write back the XCD's whole L2 once per workgroup (measured: 22 us for the
        // 8,192-row case, more than the column-strided kernel it replaces on the smaller ones).  The finishing workgroup
        // reads the rows with agent-scope loads, which do not hit in a stale line of its own L2.
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(scr + (long)slice * 64 + lane),
                           __builtin_bit_cast(unsigned long long, a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0)
            is_last = __hip_atomic_fetch_add(&g_red_count[region][cb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
                              (unsigned)nslices - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!is_last) return false;
    // the finishing workgroup: wave w adds slices w, w + 4, ... in that order (plain loads, 8 in flight), then the four
    // wave sums are added in wave order -- fixed by the indices, whoever finishes last
    double t = 0.0;
    for (int s0 = wave; s0 < nslices; s0 += 32) {
        double u[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
            u[i] = s0 + 4 * i < nslices
                       ? __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(scr + (long)(s0 + 4 * i) * 64 + lane),
                                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                       : 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += u[i];
    }
    __syncthreads();                                       // (sh is read above by every wave)
    sh[wave][lane] = t;
    __syncthreads();
    if (wave != 0) return false;
    tot = ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane];
    if (lane == 0) __hip_atomic_store(&g_red_count[region][cb], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch that draws this region
    return true;
}

// the per-channel arithmetic of the BatchNorm statistics finalize (both forms)
__device__ __forceinline__ void bn_finalize_channel(int c, double sum, double sumsq, double count, const float* gamma,
                                                    const float* beta, float eps, float momentum, float* rmean, float* rvar,
                                                    float4* tx_out, float* rstd_out) {
    double mean = sum / count;
    double var = sumsq / count - mean * mean;
    if (var < 0.0) var = 0.0;
    double rstd = 1.0 / sqrt(var + (double)eps);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const double scale = (double)g * rstd;
    tx_out[c] = make_float4((float)mean, (float)scale, (float)((double)b - mean * scale), 0.f);
    if (rstd_out) rstd_out[c] = (float)rstd;
    if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
    if (rvar) {
        double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

// ------------------------------------------------------------------------------------------
// BatchNorm statistics finalize: fixed-order double reduction (one block per channel, or the 2-D form above).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int rows, int C, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* __restrict__ rmean,
                                                          float* __restrict__ rvar, float4* __restrict__ tx_out,
                                                          float* __restrict__ rstd_out) {
    __shared__ double sh[2][256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    // a thread's rows are a line apart each: the launch is bound by load round trips, so 8 rows (16 loads) are in flight per
    // trip (clamped address + select: no branch around a load); the additions keep their row order (same sums as a plain loop)
    for (int r0 = tid; r0 < rows; r0 += 256 * 8) {
        float vs[8], vq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = r0 + 256 * i, rc = r < rows ? r : rows - 1;
            vs[i] = part[((long)rc * 2 + 0) * C + c];
            vq[i] = part[((long)rc * 2 + 1) * C + c];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(vs[i]), "+v"(vq[i]));      // (all 16 loads issued; none sunk under a predicate)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const bool ok = r0 + 256 * i < rows; vs[i] = ok ? vs[i] : 0.f; vq[i] = ok ? vq[i] : 0.f; }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s += (double)vs[i]; q += (double)vq[i]; }
    }
    sh[0][tid] = s;
    sh[1][tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) bn_finalize_channel(c, sh[0][0], sh[1][0], count, gamma, beta, eps, momentum, rmean, rvar, tx_out, rstd_out);
}

__global__ __launch_bounds__(256) void bn_finalize2d_kernel(const float* __restrict__ part, int rows, int C, double count,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, float momentum, float* __restrict__ rmean,
                                                            float* __restrict__ rvar, float4* __restrict__ tx_out,
                                                            float* __restrict__ rstd_out, int region) {
    double tot = 0.0;
    if (!red2d_block(part, rows, C, true, region, tot)) return;
    // wave 0 of the finishing workgroup: lane l < 32 holds channel c's sum, lane l + 32 its sum of squares
    const int lane = threadIdx.x & 63;
    const double q = __shfl(tot, (lane & 31) + 32, 64);
    const int c = blockIdx.x * 32 + lane;
    if (lane < 32 && c < C) bn_finalize_channel(c, tot, q, count, gamma, beta, eps, momentum, rmean, rvar, tx_out, rstd_out);
}

extern "C" int umi_bn_finalize(const float* stat_part, int rows, int C, double count, const float* gamma,
                               const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                               void* tx_out, float* rstd_out, umi_stream_t stream) {
    if (!stat_part || !tx_out || rows <= 0 || C <= 0 || count <= 0) return UMI_ERR_BADARG;
    if (red2d_ok(rows, C))
        hipLaunchKernelGGL(bn_finalize2d_kernel, dim3((C + 31) / 32, (rows + RED_RPS - 1) / RED_RPS), dim3(256), 0,
                           (hipStream_t)stream, stat_part, rows, C, count, gamma, beta, eps, momentum, running_mean,
                           running_var, (float4*)tx_out, rstd_out, red2d_region());
    else
        hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, stat_part, rows, C, count, gamma,
                           beta, eps, momentum, running_mean, running_var, (float4*)tx_out, rstd_out);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// ------------------------------------------------------------------------------------------
// MaxPool2d(2) forward / backward on the transformed tensor (thread = one output pixel x 4 channels)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void pool2_fwd_kernel(const T* __restrict__ x, int ldx, const float4* __restrict__ tx, T* __restrict__ y,
                                 int ldy, int N, int H, int W, int C) {
    const int Ho = H / 2, Wo = W / 2, C4 = (C + 3) / 4;
    long total = (long)N * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cg = (int)(i % C4);
        long p = i / C4;
        int wo = (int)(p % Wo);
        long r = p / Wo;
        int ho = (int)(r % Ho);
        int n = (int)(r / Ho);
        for (int j = 0; j < 4; ++j) {
            int c = cg * 4 + j;
            if (c >= C) break;
            float m = -INFINITY;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int h = 2 * ho + (d >> 1), w = 2 * wo + (d & 1);
                float v = (float)x[((long)((long)n * H + h) * W + w) * ldx + c];
                if (tx) v = umi_tx(v, tx[c]);
                m = (v > m) ? v : m;
            }
            y[((long)((long)n * Ho + ho) * Wo + wo) * ldy + c] = (T)m;
        }
    }
}

template <typename T>
__global__ void pool2_bwd_kernel(const T* __restrict__ dp, int lddp, const T* __restrict__ x, int ldx,
                                 const float4* __restrict__ tx, T* __restrict__ da, int ldda, int accumulate, int N,
                                 int H, int W, int C) {
    // one thread per (input pixel window, 4 channels): covers the full HxW input incl. odd tails
    const int Hw = (H + 1) / 2, Ww = (W + 1) / 2, Ho = H / 2, Wo = W / 2, C4 = (C + 3) / 4;
    long total = (long)N * Hw * Ww * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cg = (int)(i % C4);
        long p = i / C4;
        int wq = (int)(p % Ww);
        long r = p / Ww;
        int hq = (int)(r % Hw);
        int n = (int)(r / Hw);
        const bool win = hq < Ho && wq < Wo;
        for (int j = 0; j < 4; ++j) {
            int c = cg * 4 + j;
            if (c >= C) break;
            int best = -1;
            float g = 0.f;
            if (win) {
                float m = -INFINITY;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    int h = 2 * hq + (d >> 1), w = 2 * wq + (d & 1);
                    float v = (float)x[((long)((long)n * H + h) * W + w) * ldx + c];
                    if (tx) v = umi_tx(v, tx[c]);
                    if (v > m || best < 0) { m = v; best = d; }      // first max wins ties (PyTorch rule)
                }
                g = (float)dp[((long)((long)n * Ho + hq) * Wo + wq) * lddp + c];
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int h = 2 * hq + (d >> 1), w = 2 * wq + (d & 1);
                if (h >= H || w >= W) continue;
                T* q = da + ((long)((long)n * H + h) * W + w) * ldda + c;
                float v = (d == best) ? g : 0.f;
                if (accumulate) v += (float)(*q);
                *q = (T)v;
            }
        }
    }
}

extern "C" int umi_pool2_fwd(const void* x, int ldx, const void* tx, void* y, int ldy, int N, int H, int W, int C,
                             int dtype, umi_stream_t stream) {
    if (N <= 0 || H < 2 || W < 2 || C <= 0) return UMI_ERR_BADARG;
    if (dtype == UMI_F16 && umi_pool2_fwd_f16v(x, ldx, tx, y, ldy, N, H, W, C, (hipStream_t)stream)) {
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    long total = (long)N * (H / 2) * (W / 2) * ((C + 3) / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    if (dtype == UMI_F32) hipLaunchKernelGGL(pool2_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, ldx, (const float4*)tx, (float*)y, ldy, N, H, W, C);
    else if (dtype == UMI_F16) hipLaunchKernelGGL(pool2_fwd_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx, (const float4*)tx, (half_t*)y, ldy, N, H, W, C);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_pool2_bwd(const void* dpool, int lddp, const void* x, int ldx, const void* tx, void* da, int ldda,
                             int accumulate, int N, int H, int W, int C, int dtype, umi_stream_t stream) {
    if (N <= 0 || H < 2 || W < 2 || C <= 0) return UMI_ERR_BADARG;
    if (dtype == UMI_F16 &&
        umi_pool2_bwd_f16v(dpool, lddp, x, ldx, tx, da, ldda, accumulate, N, H, W, C, (hipStream_t)stream)) {
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    long total = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * ((C + 3) / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    if (dtype == UMI_F32) hipLaunchKernelGGL(pool2_bwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)dpool, lddp, (const float*)x, ldx, (const float4*)tx, (float*)da, ldda, accumulate, N, H, W, C);
    else if (dtype == UMI_F16) hipLaunchKernelGGL(pool2_bwd_kernel<half_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const half_t*)dpool, lddp, (const half_t*)x, ldx, (const float4*)tx, (half_t*)da, ldda, accumulate, N, H, W, C);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// ------------------------------------------------------------------------------------------
// BatchNorm + ReLU backward: two-stage deterministic reduction, then elementwise apply.
// Stage 1: block b reduces pixel rows [b*RPB, (b+1)*RPB) -> ws[b][2][C]; stage 2: one block / channel.
// ------------------------------------------------------------------------------------------
constexpr int BNB_RPB = 256;     // pixel rows per stage-1 block

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce1_kernel(const T* __restrict__ da, int ldda, const T* __restrict__ y,
                                                             int ldy, const float4* __restrict__ tx,
                                                             const float* __restrict__ rstd, float* __restrict__ ws,
                                                             long M, int C) {
    // thread layout: 256 threads = (256/CT) pixel lanes x CT channel lanes, CT = min(C rounded up, 64)
    __shared__ float red[2][256];
    const int tid = threadIdx.x;
    const int CT = C >= 64 ? 64 : (C >= 32 ? 32 : (C >= 16 ? 16 : (C >= 8 ? 8 : (C >= 4 ? 4 : (C >= 2 ? 2 : 1)))));
    const int PL = 256 / CT;
    const int cl = tid % CT, pl = tid / CT;
    const long r0 = (long)blockIdx.x * BNB_RPB;
    long r1 = r0 + BNB_RPB;
    if (r1 > M) r1 = M;
    for (int cb = 0; cb < C; cb += CT) {
        int c = cb + cl;
        float s = 0.f, q = 0.f;
        if (c < C) {
            const float4 t = tx[c];
            const float rs = rstd[c];
            for (long r = r0 + pl; r < r1; r += PL) {
                float yv = (float)y[r * ldy + c];
                float g = (float)da[r * ldda + c];
                float z = umi_tx_pre(yv, t);
                float dz = z > t.w ? g : 0.f;
                s += dz;
                q = fmaf(dz, (yv - t.x) * rs, q);
            }
        }
        red[0][tid] = s;
        red[1][tid] = q;
        __syncthreads();
        if (tid < CT) {
            float a = 0.f, b = 0.f;
            for (int k = 0; k < PL; ++k) { a += red[0][k * CT + tid]; b += red[1][k * CT + tid]; }
            if (cb + tid < C) {
                ws[((long)blockIdx.x * 2 + 0) * C + cb + tid] = a;
                ws[((long)blockIdx.x * 2 + 1) * C + cb + tid] = b;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void reduce_rows2_kernel(const float* __restrict__ ws, int rows, int C,
                                                           float* __restrict__ out0, float* __restrict__ out1,
                                                           float scale) {
    __shared__ double sh[2][256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s = 0.0, q = 0.0;
    const int w1 = out1 ? 1 : 0;                            // (without a second output the second column is not read: same address twice)
    for (int r0 = tid; r0 < rows; r0 += 256 * 8) {          // 8 rows in flight per trip, additions in row order (see bn_finalize_kernel)
        float vs[8], vq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = r0 + 256 * i, rc = r < rows ? r : rows - 1;
            vs[i] = ws[((long)rc * 2 + 0) * C + c];
            vq[i] = ws[((long)rc * 2 + w1) * C + c];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(vs[i]), "+v"(vq[i]));
#pragma unroll
        for (int i = 0; i < 8; ++i) { const bool ok = r0 + 256 * i < rows; vs[i] = ok ? vs[i] : 0.f; vq[i] = (ok && out1) ? vq[i] : 0.f; }
#pragma unroll
        for (int i = 0; i < 8; ++i) { s += (double)vs[i]; q += (double)vq[i]; }
    }
    sh[0][tid] = s;
    sh[1][tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { sh[0][tid] += sh[0][tid + o]; sh[1][tid] += sh[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        out0[c] = (float)(sh[0][0] * (double)scale);
        if (out1) out1[c] = (float)(sh[1][0] * (double)scale);
    }
}

__global__ __launch_bounds__(256) void reduce_rows2_2d_kernel(const float* __restrict__ ws, int rows, int C,
                                                              float* __restrict__ out0, float* __restrict__ out1,
                                                              float scale, int region) {
    double tot = 0.0;
    if (!red2d_block(ws, rows, C, out1 != nullptr, region, tot)) return;
    const int lane = threadIdx.x & 63, c = blockIdx.x * 32 + (lane & 31);
    if (c < C) {
        if (lane < 32) out0[c] = (float)(tot * (double)scale);
        else if (out1) out1[c] = (float)(tot * (double)scale);
    }
}

// few rows, very many columns (the position-embedding gradient: 24 batch rows x 150,528 columns): one THREAD per column, the
// lanes of a wave on adjacent columns -- a workgroup per column there is 150 K workgroups of 256 threads for 24 additions each
// (149 us against ~6 us)
__global__ __launch_bounds__(256) void reduce_rows2_wide_kernel(const float* __restrict__ ws, int rows, int C,
                                                                float* __restrict__ out0, float* __restrict__ out1, float scale) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int r = 0; r < rows; ++r) {
        s += (double)ws[((long)r * 2 + 0) * C + c];
        if (out1) q += (double)ws[((long)r * 2 + 1) * C + c];
    }
    out0[c] = (float)(s * (double)scale);
    if (out1) out1[c] = (float)(q * (double)scale);
}

void umi_launch_reduce_rows2(const float* ws, int rows, int C, float* out0, float* out1, float scale, hipStream_t s) {
    if (rows <= 64 && C >= 4096)
        hipLaunchKernelGGL(reduce_rows2_wide_kernel, dim3((C + 255) / 256), dim3(256), 0, s, ws, rows, C, out0, out1, scale);
    else if (red2d_ok(rows, C))
        hipLaunchKernelGGL(reduce_rows2_2d_kernel, dim3((C + 31) / 32, (rows + RED_RPS - 1) / RED_RPS), dim3(256), 0, s, ws,
                           rows, C, out0, out1, scale, red2d_region());
    else
        hipLaunchKernelGGL(reduce_rows2_kernel, dim3(C), dim3(256), 0, s, ws, rows, C, out0, out1, scale);
}

template <typename T>
__global__ void bn_bwd_apply_kernel(T* __restrict__ da, int ldda, const T* __restrict__ y, int ldy,
                                    const float4* __restrict__ tx, const float* __restrict__ rstd,
                                    const float* __restrict__ sum_dz, const float* __restrict__ sum_dzx, long M, int C) {
    const float invM = 1.f / (float)M;
    long total = M * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C);
        long r = i / C;
        // the same expression, term by term, as the vectorised kernel and the fused weight-gradient kernel (common.h)
        da[r * ldda + c] = umi_bn_dz<T>((float)y[r * ldy + c], (float)da[r * ldda + c], tx[c], rstd[c], sum_dz[c] * invM,
                                        sum_dzx[c] * invM);
    }
}

extern "C" size_t umi_bn_bwd_ws_bytes(long M, int C) {
    const int a = umi_cdiv(M, BNB_RPB), b = umi_cdiv(M, umi_bn_bwd_rpb_f16v(M));       // whichever path needs more rows
    return (size_t)(a > b ? a : b) * 2 * (size_t)C * sizeof(float);
}

extern "C" int umi_bn_bwd_reduce(const void* da, int ldda, const void* y, int ldy, const void* tx, const float* rstd,
                                 float* sum_dz, float* sum_dzx, long M, int C, int dtype, void* ws, size_t ws_bytes,
                                 umi_stream_t stream) {
    if (!da || !y || !tx || !rstd || M <= 0 || C <= 0) return UMI_ERR_BADARG;
    if (ws_bytes < umi_bn_bwd_ws_bytes(M, C)) return UMI_ERR_WORKSPACE;
    int rows = umi_cdiv(M, BNB_RPB);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UMI_F16 && umi_bn_bwd_reduce1_f16v(da, ldda, y, ldy, tx, rstd, (float*)ws, M, C, s)) {
        UMI_LAUNCH_CHECK();
        rows = umi_cdiv(M, umi_bn_bwd_rpb_f16v(M));
        umi_launch_reduce_rows2((const float*)ws, rows, C, sum_dz, sum_dzx, 1.f, s);
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    if (dtype == UMI_F32) hipLaunchKernelGGL(bn_bwd_reduce1_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)da, ldda, (const float*)y, ldy, (const float4*)tx, rstd, (float*)ws, M, C);
    else if (dtype == UMI_F16) hipLaunchKernelGGL(bn_bwd_reduce1_kernel<half_t>, dim3(rows), dim3(256), 0, s, (const half_t*)da, ldda, (const half_t*)y, ldy, (const float4*)tx, rstd, (float*)ws, M, C);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    umi_launch_reduce_rows2((const float*)ws, rows, C, sum_dz, sum_dzx, 1.f, s);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_bn_bwd_apply(void* da, int ldda, const void* y, int ldy, const void* tx, const float* rstd,
                                const float* sum_dz, const float* sum_dzx, long M, int C, int dtype,
                                umi_stream_t stream) {
    if (!da || !y || !tx || !rstd || M <= 0 || C <= 0) return UMI_ERR_BADARG;
    long total = M * C;
    int grid = (int)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UMI_F16 && umi_bn_bwd_apply_f16v(da, ldda, y, ldy, tx, rstd, sum_dz, sum_dzx, M, C, s)) {
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    if (dtype == UMI_F32) hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)da, ldda, (const float*)y, ldy, (const float4*)tx, rstd, sum_dz, sum_dzx, M, C);
    else if (dtype == UMI_F16) hipLaunchKernelGGL(bn_bwd_apply_kernel<half_t>, dim3(grid), dim3(256), 0, s, (half_t*)da, ldda, (const half_t*)y, ldy, (const float4*)tx, rstd, sum_dz, sum_dzx, M, C);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// ------------------------------------------------------------------------------------------
// per-channel column sum (bias gradients)
// ------------------------------------------------------------------------------------------
// 2-D decomposition: block (column tile of 64, row split); thread = (column, 1 of 4 row lanes); partial rows
// ws[split][2][C] are reduced in fixed order by reduce_rows2_kernel.
template <typename T>
__global__ __launch_bounds__(256) void colsum1_kernel(const T* __restrict__ x, int ldx, float* __restrict__ ws, long M,
                                                      int C, long rows_per_split, float* __restrict__ out, float scale) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long r0 = (long)blockIdx.y * rows_per_split;
    long r1 = r0 + rows_per_split;
    if (r1 > M) r1 = M;
    float s = 0.f;
    if (c < C)
        for (long r = r0 + rl; r < r1; r += 4) s += (float)x[r * ldx + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        const float t = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
        if (gridDim.y == 1) out[c] = t * scale;                       // single split: no second stage
        else ws[((long)blockIdx.y * 2 + 0) * C + c] = t;
    }
}

__global__ void fold_cols_kernel(const float* __restrict__ tmp8, int C, float* __restrict__ out, float scale) {
    const int c = threadIdx.x;
    if (c < C) {
        float s = 0.f;
        for (int k = c; k < 8; k += C) s += tmp8[k];
        out[c] = s * scale;
    }
}

static void colsum_plan(long M, int C, int* splits, long* rps) {
    long ctiles = (C + 63) / 64;
    long want = (1024 + ctiles - 1) / ctiles;
    long maxs = (M + 31) / 32;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    *rps = (M + want - 1) / want;
    *splits = (int)((M + *rps - 1) / *rps);
}

extern "C" size_t umi_colsum_ws_bytes(long M, int C) {
    int splits; long rps;
    colsum_plan(M, C, &splits, &rps);
    const int vrows = umi_colsum_rows_f16v(M, C);
    size_t b = (size_t)(splits > vrows ? splits : vrows) * 2 * (size_t)C * sizeof(float);
    if (C < 8 && 8 % C == 0 && (M * C) % 8 == 0) {           // the small-C path reads the tensor as rows of 8
        const size_t b8 = ((size_t)umi_colsum_rows_f16v(M * C / 8, 8) * 2 * 8 + 8) * sizeof(float);
        if (b8 > b) b = b8;
    }
    return b;
}

// umi_colsum for n tensors of one shape in two launches per 16 (fp16, C % 8 == 0; UMI_ERR_UNSUPPORTED otherwise): `ws` needs
// min(n, 16) * umi_colsum_ws_bytes(M, C) bytes
bool umi_colsum_group_f16v(int n, const void* const* xs, int ldx, float* const* outs, float scale, float* ws, long M, int C,
                           hipStream_t s);
extern "C" int umi_colsum_group(int n, const void* const* xs, int ldx, float* const* outs, float out_scale, long M, int C,
                                int dtype, void* ws, size_t ws_bytes, umi_stream_t stream) {
    if (n <= 0 || !xs || !outs || M <= 0 || C <= 0 || !ws) return UMI_ERR_BADARG;
    for (int i = 0; i < n; ++i)
        if (!xs[i] || !outs[i]) return UMI_ERR_BADARG;
    if (ws_bytes < (size_t)(n < 16 ? n : 16) * umi_colsum_ws_bytes(M, C)) return UMI_ERR_WORKSPACE;
    if (dtype != UMI_F16 || !umi_colsum_group_f16v(n, xs, ldx, outs, out_scale, (float*)ws, M, C, (hipStream_t)stream))
        return UMI_ERR_UNSUPPORTED;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_colsum(const void* x, int ldx, float* out, float out_scale, long M, int C, int dtype, void* ws,
                          size_t ws_bytes, umi_stream_t stream) {
    if (!x || !out || M <= 0 || C <= 0) return UMI_ERR_BADARG;
    if (ws_bytes < umi_colsum_ws_bytes(M, C)) return UMI_ERR_WORKSPACE;
    int splits; long rps;
    colsum_plan(M, C, &splits, &rps);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UMI_F16 && umi_colsum_f16v(x, ldx, (float*)ws, M, C, s)) {
        UMI_LAUNCH_CHECK();
        umi_launch_reduce_rows2((const float*)ws, umi_colsum_rows_f16v(M, C), C, out, (float*)nullptr, out_scale, s);
        UMI_LAUNCH_CHECK();
        return UMI_OK;
    }
    // a dense tensor with fewer than 8 channels (the n_classes-wide logits gradient): read it as rows of 8 values, sum the 8
    // interleaved columns with the vectorised kernel, fold column c + C*k into channel c
    if (dtype == UMI_F16 && C < 8 && 8 % C == 0 && ldx == C && (M * C) % 8 == 0 &&
        ws_bytes >= ((size_t)umi_colsum_rows_f16v(M * C / 8, 8) * 2 * 8 + 8) * sizeof(float)) {
        const long M8 = M * C / 8;
        float* tmp = (float*)ws + (size_t)umi_colsum_rows_f16v(M8, 8) * 2 * 8;
        if (umi_colsum_f16v(x, 8, (float*)ws, M8, 8, s)) {
            UMI_LAUNCH_CHECK();
            umi_launch_reduce_rows2((const float*)ws, umi_colsum_rows_f16v(M8, 8), 8, tmp, (float*)nullptr, 1.f, s);
            hipLaunchKernelGGL(fold_cols_kernel, dim3(1), dim3(64), 0, s, (const float*)tmp, C, out, out_scale);
            UMI_LAUNCH_CHECK();
            return UMI_OK;
        }
    }
    dim3 grid((C + 63) / 64, splits);
    if (dtype == UMI_F32) hipLaunchKernelGGL(colsum1_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ldx, (float*)ws, M, C, rps, out, out_scale);
    else if (dtype == UMI_F16) hipLaunchKernelGGL(colsum1_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)x, ldx, (float*)ws, M, C, rps, out, out_scale);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    if (splits > 1) {
        umi_launch_reduce_rows2((const float*)ws, splits, C, out, (float*)nullptr, out_scale, s);
        UMI_LAUNCH_CHECK();
    }
    return UMI_OK;
}

// ------------------------------------------------------------------------------------------
// generic weight gradient: per tap a [64 ci] x [64 co] tile, K = a chunk of output pixels
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void wgrad_generic_kernel(
    const T* __restrict__ x, int ldx, const float4* __restrict__ txa, const T* __restrict__ dy, int lddy,
    const float4* __restrict__ txb, float* __restrict__ part, int N, int H, int W, int Ci, int Co, int R, int S,
    int stride, int pad, int Ho, int Wo, long chunk, int tiles_co) {
    __shared__ __attribute__((aligned(16))) float As[GBK][64 + 4];
    __shared__ __attribute__((aligned(16))) float Bs[GBK][64 + 4];
    const int tid = threadIdx.x;
    const int txi = tid & 15, tyi = tid >> 4;
    const int tile_ci = blockIdx.x / tiles_co, tile_co = blockIdx.x % tiles_co;
    const int ci0 = tile_ci * 64, co0 = tile_co * 64;
    const int tap = blockIdx.y;
    const int r = tap / S, s = tap - r * S;
    const long P = (long)N * Ho * Wo;
    const long k_begin = (long)blockIdx.z * chunk;
    long k_end = k_begin + chunk;
    if (k_end > P) k_end = P;

    const int lk = tid >> 4;          // pixel within the K tile
    const int lc = (tid & 15) * 4;    // 4 consecutive channels
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (long k0 = k_begin; k0 < k_end; k0 += GBK) {
        long pg = k0 + lk;
        bool pv = pg < k_end;
        int n = 0, ho = 0, wo = 0;
        if (pv) {
            n = (int)(pg / ((long)Ho * Wo));
            int rem = (int)(pg - (long)n * Ho * Wo);
            ho = rem / Wo;
            wo = rem - ho * Wo;
        }
        int hi = ho * stride - pad + r, wi = wo * stride - pad + s;
        bool inb = pv && hi >= 0 && hi < H && wi >= 0 && wi < W;
        const T* xp = x + ((long)((long)n * H + hi) * W + wi) * ldx;
        const T* dp = dy + pg * lddy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int ci = ci0 + lc + j;
            float v = 0.f;
            if (inb && ci < Ci) {
                v = (float)xp[ci];
                if (txa) v = umi_tx(v, txa[ci]);
            }
            As[lk][lc + j] = v;
            int co = co0 + lc + j;
            float g = 0.f;
            if (pv && co < Co) {
                g = (float)dp[co];
                if (txb) g = umi_tx(g, txb[co]);
            }
            Bs[lk][lc + j] = g;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GBK; ++kk) {
            float4 a = *(const float4*)&As[kk][tyi * 4];
            float4 b = *(const float4*)&Bs[kk][txi * 4];
            float av[4] = {a.x, a.y, a.z, a.w};
            float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    // part[z][tap][ci][co]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ci = ci0 + tyi * 4 + i;
        if (ci >= Ci) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int co = co0 + txi * 4 + j;
            if (co < Co) part[(((long)blockIdx.z * R * S + tap) * Ci + ci) * Co + co] = acc[i][j];
        }
    }
}

// Fixed-order split-K reduction: block = 32 outputs x 8 split lanes; lane l sums splits l, l+8, ... then the 8 lane
// sums are added in lane order (deterministic for a given number of splits).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int splits, int RS, int Ci,
                                                           int Co, float* __restrict__ dW, long s_co, long s_ci, long s_t,
                                                           float scale) {
    __shared__ float red[8][33];
    const long total = (long)RS * Ci * Co;
    const int ox = threadIdx.x & 31, sl = threadIdx.x >> 5;
    for (long base = (long)blockIdx.x * 32; base < total; base += (long)gridDim.x * 32) {
        long i = base + ox;
        float s = 0.f;
        if (i < total)
            for (int z = sl; z < splits; z += 8) s += part[(long)z * total + i];
        red[sl][ox] = s;
        __syncthreads();
        if (sl == 0 && i < total) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) a += red[k][ox];
            int co = (int)(i % Co);
            long rr = i / Co;
            int ci = (int)(rr % Ci);
            int t = (int)(rr / Ci);
            dW[co * s_co + ci * s_ci + t * s_t] = a * scale;
        }
        __syncthreads();
    }
}

// Same reduction with 16-B slab loads: block = 32 float4 outputs x 8 split lanes (lane l sums splits l, l+8, ... with two
// independent chains, then the 8 lane sums are added in lane order); the stores scatter into the parameter's own layout.
__global__ __launch_bounds__(256) void wgrad_reduce_v4_kernel(const float* __restrict__ part, int splits, int RS, int Ci,
                                                              int Co, float* __restrict__ dW, long s_co, long s_ci, long s_t,
                                                              float scale) {
    __shared__ float4 red[8][33];
    const long total4 = (long)RS * Ci * Co / 4;
    const int Co4 = Co >> 2;
    const int ox = threadIdx.x & 31, sl = threadIdx.x >> 5;
    for (long base = (long)blockIdx.x * 32; base < total4; base += (long)gridDim.x * 32) {
        const long i = base + ox;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (i < total4) {
            const float4* p = reinterpret_cast<const float4*>(part) + i;
            int z = sl;
            for (; z + 8 < splits; z += 16) {
                float4 u = p[(long)z * total4], v = p[(long)(z + 8) * total4];
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
            }
            if (z < splits) { float4 u = p[(long)z * total4]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
        }
        red[sl][ox] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        __syncthreads();
        if (sl == 0 && i < total4) {
            float4 t4 = red[0][ox];
#pragma unroll
            for (int k = 1; k < 8; ++k) { float4 u = red[k][ox]; t4.x += u.x; t4.y += u.y; t4.z += u.z; t4.w += u.w; }
            const int co = (int)(i % Co4) * 4;
            const long rr = i / Co4;
            const int ci = (int)(rr % Ci), t = (int)(rr / Ci);
            float* o = dW + ci * s_ci + t * s_t;
            o[(co + 0) * s_co] = t4.x * scale;
            o[(co + 1) * s_co] = t4.y * scale;
            o[(co + 2) * s_co] = t4.z * scale;
            o[(co + 3) * s_co] = t4.w * scale;
        }
        __syncthreads();
    }
}

// Deferred form: umi_conv_wgrad_deferred arms a slot; the one reduction a weight-gradient path would launch is recorded there
// instead, and umi_wgrad_reduce_group later runs the reductions of many layers in one launch per 16 (a split-K reduction of a
// small weight is a ~8-us launch for a few hundred KB; a U-Net step has 22 of them, a TransUNet step 62).
struct WgPending { const float* part; float* dW; long s_co, s_ci, s_t; float scale; int splits, RS, Ci, Co; };   // = umi_wgrad_pending
static thread_local WgPending* g_wgrad_defer = nullptr;
void umi_wgrad_defer_set(void* slot) { g_wgrad_defer = (WgPending*)slot; }

// Transposing form for the large weights (round 3).  The slabs are [tap][ci][co] (co fastest), the parameter is OIHW (or IOHW for
// ConvTranspose2d): written element by element the outputs of a workgroup are 4-byte stores a whole row apart -- 31 M of them per
// U-Net step, partial lines that leave L2 before their neighbours arrive (0.44 ms/step for 0.7 GB = 1.6 TB/s).  Here a workgroup
// owns a tile of 32 output channels x 8 input channels x all taps, reduces it row by row with the SAME per-element arithmetic
// (lane l sums splits l, l + 8, ... in two chains, the 8 lane sums added in lane order: bit-identical results), collects the
// values in LDS in the parameter's order and writes runs of 8 * RS (OIHW) or 32 * RS (IOHW) consecutive floats.
// Taken where it yields >= 512 tiles (small weights keep the element-wise form: their parallelism is in the split dimension).
constexpr int WGT_CO = 32, WGT_CI = 8;
__host__ __device__ inline int wg_tmode(const WgPending& d) {       // 0 = element-wise, 1 = OIHW tiles, 2 = IOHW tiles
    if (d.Co % WGT_CO || d.Ci % WGT_CI || d.RS > 9 || d.s_t != 1 || (((uintptr_t)d.part) & 15)) return 0;
    if ((long)(d.Co / WGT_CO) * (d.Ci / WGT_CI) < 512) return 0;
    if (d.s_ci == d.RS && d.s_co == (long)d.Ci * d.RS) return 1;
    if (d.s_co == d.RS && d.s_ci == (long)d.Co * d.RS) return 2;
    return 0;
}
__device__ void wg_reduce_tiles(const WgPending& d, int mode, int my_blk, int n_blk) {
    __shared__ float4 redt[8][33];
    __shared__ float tile[WGT_CO * WGT_CI * 9 + 64];
    const int ox = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int RS = d.RS, ncob = d.Co / WGT_CO, ncib = d.Ci / WGT_CI, rows = WGT_CI * RS;
    const long total4 = (long)RS * d.Ci * d.Co / 4;             // float4 per split slab
    for (int tl = my_blk; tl < ncob * ncib; tl += n_blk) {
        const int cob = tl % ncob, cib = tl / ncob;
        for (int r0 = 0; r0 < rows; r0 += 4) {                  // 4 (tap, ci) rows x 8 float4 of output channels per pass
            const int row = r0 + (ox >> 3), c4 = ox & 7;
            const int tap = row / WGT_CI, cil = row - tap * WGT_CI;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (row < rows) {
                const float4* p = reinterpret_cast<const float4*>(d.part) + (((long)tap * d.Ci + cib * WGT_CI + cil) * d.Co + cob * WGT_CO) / 4 + c4;
                int z = sl;
                for (; z + 8 < d.splits; z += 16) {
                    float4 u = p[(long)z * total4], v = p[(long)(z + 8) * total4];
                    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                    b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
                }
                if (z < d.splits) { float4 u = p[(long)z * total4]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
                a = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            }
            redt[sl][ox] = a;
            __syncthreads();
            if (sl == 0 && row < rows) {
                float4 t4 = redt[0][ox];
#pragma unroll
                for (int k = 1; k < 8; ++k) { float4 u = redt[k][ox]; t4.x += u.x; t4.y += u.y; t4.z += u.z; t4.w += u.w; }
                const float v[4] = {t4.x * d.scale, t4.y * d.scale, t4.z * d.scale, t4.w * d.scale};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int co = c4 * 4 + k;
                    // OIHW: [co][ci][tap]   IOHW: [ci][co][tap]   (+1 float of padding per outer row against bank conflicts)
                    const int idx = mode == 1 ? co * (WGT_CI * RS + 1) + cil * RS + tap : cil * (WGT_CO * RS + 1) + co * RS + tap;
                    tile[idx] = v[k];
                }
            }
            __syncthreads();
        }
        // the tile in the parameter's order: runs of consecutive floats
        const int inner = mode == 1 ? WGT_CI * RS : WGT_CO * RS, outer = mode == 1 ? WGT_CO : WGT_CI;
        for (int e = threadIdx.x; e < inner * outer; e += 256) {
            const int o = e / inner, r = e - o * inner;
            float* dst = mode == 1 ? d.dW + (long)(cob * WGT_CO + o) * d.s_co + (long)(cib * WGT_CI) * d.s_ci + r
                                   : d.dW + (long)(cib * WGT_CI + o) * d.s_ci + (long)(cob * WGT_CO) * d.s_co + r;
            *dst = tile[o * (inner + 1) + r];
        }
        __syncthreads();
    }
}

// up to 16 recorded reductions per launch: blockIdx.y = entry, the block loop and the arithmetic (lane l sums splits l, l+8, ...
// in two chains, lanes added in order) are wgrad_reduce_v4_kernel's / wgrad_reduce_kernel's, so the results are identical
struct WgTable { WgPending e[16]; int blk0[17]; };          // blk0: first workgroup of each entry (entry i owns blk0[i+1] - blk0[i])
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(WgTable t) {
    __shared__ float4 red[8][33];
    int ent = 0;
#pragma unroll
    for (int i = 1; i < 16; ++i) ent += (int)blockIdx.x >= t.blk0[i];
    const WgPending d = t.e[ent];
    const int my_blk = (int)blockIdx.x - t.blk0[ent], n_blk = t.blk0[ent + 1] - t.blk0[ent];
    if (const int tm = wg_tmode(d)) { wg_reduce_tiles(d, tm, my_blk, n_blk); return; }
    const int ox = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const bool vec = d.Co % 4 == 0 && (((uintptr_t)d.part) & 15) == 0;
    const int W4 = vec ? 4 : 1;
    const long total = (long)d.RS * d.Ci * d.Co / W4;
    const int CoW = d.Co / W4;
    for (long base = (long)my_blk * 32; base < total; base += (long)n_blk * 32) {
        const long i = base + ox;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (i < total) {
            if (vec) {
                const float4* p = reinterpret_cast<const float4*>(d.part) + i;
                int z = sl;
                for (; z + 8 < d.splits; z += 16) {
                    float4 u = p[(long)z * total], v = p[(long)(z + 8) * total];
                    a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                    b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
                }
                if (z < d.splits) { float4 u = p[(long)z * total]; a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w; }
                a = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            } else {
                for (int z = sl; z < d.splits; z += 8) a.x += d.part[(long)z * total + i];
            }
        }
        red[sl][ox] = a;
        __syncthreads();
        if (sl == 0 && i < total) {
            float4 t4 = red[0][ox];
            if (vec) {
#pragma unroll
                for (int k = 1; k < 8; ++k) { float4 u = red[k][ox]; t4.x += u.x; t4.y += u.y; t4.z += u.z; t4.w += u.w; }
            } else {
                float s0 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) s0 += red[k][ox].x;
                t4.x = s0;
            }
            const int co = (int)(i % CoW) * W4;
            const long rr = i / CoW;
            const int ci = (int)(rr % d.Ci), tp = (int)(rr / d.Ci);
            float* o = d.dW + ci * d.s_ci + tp * d.s_t;
            o[(co + 0) * d.s_co] = t4.x * d.scale;
            if (vec) {
                o[(co + 1) * d.s_co] = t4.y * d.scale;
                o[(co + 2) * d.s_co] = t4.z * d.scale;
                o[(co + 3) * d.s_co] = t4.w * d.scale;
            }
        }
        __syncthreads();
    }
}

extern "C" int umi_wgrad_reduce_group(int n, const void* items, umi_stream_t stream) {
    if (n <= 0 || !items) return UMI_ERR_BADARG;
    const WgPending* it = (const WgPending*)items;
    for (int g0 = 0; g0 < n; g0 += 16) {
        const int cnt = n - g0 < 16 ? n - g0 : 16;
        WgTable t;
        int total_blk = 0;
        for (int i = 0; i < 16; ++i) {
            t.e[i] = it[g0 + (i < cnt ? i : 0)];
            t.blk0[i] = total_blk;
            if (i >= cnt) continue;                              // (padding entries own no workgroups)
            if (!t.e[i].part || !t.e[i].dW || t.e[i].splits <= 0) return UMI_ERR_BADARG;
            const bool vec = t.e[i].Co % 4 == 0 && (((uintptr_t)t.e[i].part) & 15) == 0;
            long g = ((long)t.e[i].RS * t.e[i].Ci * t.e[i].Co / (vec ? 4 : 1) + 31) / 32;
            if (wg_tmode(t.e[i])) g = (long)(t.e[i].Co / WGT_CO) * (t.e[i].Ci / WGT_CI);
            total_blk += (int)(g > 4096 ? 4096 : g);
        }
        t.blk0[16] = total_blk;
        hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3((unsigned)total_blk), dim3(256), 0, (hipStream_t)stream, t);
        UMI_LAUNCH_CHECK();
    }
    return UMI_OK;
}

void umi_launch_wgrad_reduce(const float* part, int splits, int RS, int Ci, int Co, float* dW, long s_co, long s_ci,
                             long s_t, float scale, hipStream_t st) {
    if (g_wgrad_defer) {
        *g_wgrad_defer = WgPending{part, dW, s_co, s_ci, s_t, scale, splits, RS, Ci, Co};
        g_wgrad_defer = nullptr;
        return;
    }
    {
        const WgPending one{part, dW, s_co, s_ci, s_t, scale, splits, RS, Ci, Co};
        if (wg_tmode(one)) {                                    // large weight: the transposing tile form (a one-entry group launch)
            WgTable t;
            for (int i = 0; i < 16; ++i) { t.e[i] = one; t.blk0[i] = 0; }
            long g = (long)(Co / WGT_CO) * (Ci / WGT_CI);
            if (g > 4096) g = 4096;
            for (int i = 1; i <= 16; ++i) t.blk0[i] = (int)g;
            hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3((unsigned)g), dim3(256), 0, st, t);
            return;
        }
    }
    if (Co % 4 == 0 && (((uintptr_t)part) & 15) == 0) {
        long g4 = ((long)RS * Ci * Co / 4 + 31) / 32;
        if (g4 > 16384) g4 = 16384;
        hipLaunchKernelGGL(wgrad_reduce_v4_kernel, dim3((unsigned)g4), dim3(256), 0, st, part, splits, RS, Ci, Co, dW, s_co, s_ci, s_t, scale);
        return;
    }
    long total = (long)RS * Ci * Co;
    long g = (total + 31) / 32;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, part, splits, RS, Ci, Co, dW, s_co, s_ci, s_t, scale);
}


static void wgrad_generic_plan(long P, int Ci, int Co, int RS, int* splits, long* chunk) {
    long tiles = (long)umi_cdiv(Ci, 64) * umi_cdiv(Co, 64) * RS;
    long want = (2048 + tiles - 1) / tiles;          // aim for >= 2048 workgroups
    long maxs = (P + 255) / 256;                     // at least 256 pixels per split
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    long ch = (P + want - 1) / want;
    ch = (ch + GBK - 1) / GBK * GBK;
    *chunk = ch;
    *splits = (int)((P + ch - 1) / ch);
}

size_t umi_conv_wgrad_generic_ws_bytes(int N, int Ho, int Wo, int Ci, int Co, int R, int S) {
    int splits; long chunk;
    wgrad_generic_plan((long)N * Ho * Wo, Ci, Co, R * S, &splits, &chunk);
    return (size_t)splits * R * S * Ci * Co * sizeof(float);
}

int umi_conv_wgrad_generic(const void* x, int ldx, const void* txa, const void* dy, int lddy, const void* txb, float* dW,
                           long s_co, long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci, int Co, int R,
                           int S, int stride, int pad, int Ho, int Wo, int dtype, void* ws, size_t ws_bytes,
                           hipStream_t st) {
    int splits; long chunk;
    const long P = (long)N * Ho * Wo;
    wgrad_generic_plan(P, Ci, Co, R * S, &splits, &chunk);
    if (ws_bytes < (size_t)splits * R * S * Ci * Co * sizeof(float)) return UMI_ERR_WORKSPACE;
    const int tiles_co = umi_cdiv(Co, 64);
    dim3 grid(umi_cdiv(Ci, 64) * tiles_co, R * S, splits), block(256);
    if (dtype == UMI_F32) hipLaunchKernelGGL(wgrad_generic_kernel<float>, grid, block, 0, st, (const float*)x, ldx, (const float4*)txa, (const float*)dy, lddy, (const float4*)txb, (float*)ws, N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, chunk, tiles_co);
    else if (dtype == UMI_F16) hipLaunchKernelGGL(wgrad_generic_kernel<half_t>, grid, block, 0, st, (const half_t*)x, ldx, (const float4*)txa, (const half_t*)dy, lddy, (const float4*)txb, (float*)ws, N, H, W, Ci, Co, R, S, stride, pad, Ho, Wo, chunk, tiles_co);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    umi_launch_wgrad_reduce((const float*)ws, splits, R * S, Ci, Co, dW, s_co, s_ci, s_t, out_scale, st);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// ------------------------------------------------------------------------------------------
// materialize an activation: NHWC storage + consumer transform -> NCHW fp32 (block outputs)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void materialize_nchw_kernel(const T* __restrict__ x, int ldx, const float4* __restrict__ tx,
                                        float* __restrict__ y, int N, int H, int W, int C) {
    long total = (long)N * C * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int w = (int)(i % W);
        long r = i / W;
        int h = (int)(r % H);
        r /= H;
        int c = (int)(r % C);
        int n = (int)(r / C);
        float v = (float)x[((long)((long)n * H + h) * W + w) * ldx + c];
        if (tx) v = umi_tx(v, tx[c]);
        y[i] = v;
    }
}

extern "C" int umi_materialize_nchw(const void* x, int ldx, const void* tx, float* y, int N, int H, int W, int C,
                                    int dtype, umi_stream_t stream) {
    if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return UMI_ERR_BADARG;
    long total = (long)N * C * H * W;
    int grid = (int)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UMI_F32) hipLaunchKernelGGL(materialize_nchw_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, (const float4*)tx, y, N, H, W, C);
    else if (dtype == UMI_F16) hipLaunchKernelGGL(materialize_nchw_kernel<half_t>, dim3(grid), dim3(256), 0, s, (const half_t*)x, ldx, (const float4*)tx, y, N, H, W, C);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
