// GroupNorm for fp16 NHWC tensors with channel-coalesced 16-B accesses (thread = pixel lane x 8 channels), two launches each
// way.  The statistics of a (sample, group) span all pixels of the sample, while coalesced access wants workgroups that stream
// whole pixel rows: so stage 1 (gn_rowsum_v8) writes per-channel partial sums of every (sample, row block), and the apply
// kernels finish the reduction themselves, in their prologue: every apply workgroup of a sample sums that sample's <= ~24
// partial rows (a few KB, L2-resident) into per-group statistics in LDS before it touches the tensor.  Redundant across the
// workgroups of a sample, but it replaces the finalize / part / gsum launches of the first version: on the R50 hybrid's 52
// layers a tiny launch costs ~5-8 us of an otherwise idle chip, inside a HIP graph too.  (Folding those stages into the
// LAST-ARRIVING stage-1 workgroup -- ticket counters -- was measured as well: no launch, but the tail runs on one workgroup
// per sample while the chip idles, 16.4 us against 6.4 + 4.9 for the forward statistics; and a device-scope release fence per
// workgroup writes back the whole L2: 21.1 -> 27.9 ms per TransUNet step.)
// The generic kernels in transformer_kernels.hip (one (sample, group) per workgroup, pixel-strided) remain for other shapes.
#include "common.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace {

// pixel rows per workgroup: ~512 workgroups per launch (two per CU in one round; more row blocks mean more partial rows for
// every apply workgroup to sum), at least 16 rows
static int gn_rows(int N, long HW) {
    long r = ((long)N * HW + 511) / 512;
    r = (r + 7) / 8 * 8;
    return (int)(r < 16 ? 16 : (r > 1024 ? 1024 : r));
}

// MODE 0: out = (sum (x - shift), sum (x - shift)^2) per channel.   MODE 1: out = (sum dz*xhat, sum dz) per channel, dz = dy*[y>0 | 1]
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
    // MODE 0 sums x - shift, shift = this channel's value at the sample's first pixel (the same for every row block of the
    // sample; gn_apply_fin_kernel re-reads it): a channel whose |mean| is large against its spread -- the per-channel groups of
    // the projection shortcuts' GroupNorm(C, C) -- would otherwise lose its variance in sum(x^2) - sum(x)^2 / n at fp32
    half8 sh8;
    if (MODE == 0) sh8 = *reinterpret_cast<const half8*>(x + (long)n * HW * ldx + cg * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = b[j] = 0.f;
        if (MODE == 0) mu[j] = (float)sh8[j];
        if (MODE == 1) {
            int g = (cg * 8 + j) / (C / G);
            mu[j] = mean[n * G + g];
            rs[j] = rstd[n * G + g];
        }
    }
#define UMI_GN_ACC(xv_, gv_, yv_)                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                               \
        if (MODE == 0) { float f = (float)xv_[j] - mu[j]; a[j] += f; b[j] = fmaf(f, f, b[j]); }                   \
        else {                                                                                                    \
            float dz = (float)gv_[j];                                                                             \
            if (relu && !((float)yv_[j] > 0.f)) dz = 0.f;                                                         \
            a[j] = fmaf(dz, ((float)xv_[j] - mu[j]) * rs[j], a[j]);                                               \
            b[j] += dz;                                                                                           \
        }                                                                                                         \
    }
    long r = r0 + pl;
    // four rows per trip: 4 (forward) / 8-12 (backward) independent 16-B loads in flight per thread -- a row block is only
    // ~10-20 rows per pixel lane, one load per trip left the loop at one memory round trip per row
    for (; r + 3 * PL < r1; r += 4 * PL) {
        half8 xv[4], gv[4], yv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long row = (long)n * HW + r + u * PL;
            xv[u] = *reinterpret_cast<const half8*>(x + row * ldx + cg * 8);
            if (MODE == 1) {
                gv[u] = *reinterpret_cast<const half8*>(dy + row * lddy + cg * 8);
                if (relu) yv[u] = *reinterpret_cast<const half8*>(y + row * ldy + cg * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) UMI_GN_ACC(xv[u], gv[u], yv[u])
    }
    for (; r < r1; r += PL) {
        const long row = (long)n * HW + r;
        half8 xv = *reinterpret_cast<const half8*>(x + row * ldx + cg * 8), gv, yv;
        if (MODE == 1) {
            gv = *reinterpret_cast<const half8*>(dy + row * lddy + cg * 8);
            if (relu) yv = *reinterpret_cast<const half8*>(y + row * ldy + cg * 8);
        }
        UMI_GN_ACC(xv, gv, yv)
    }
#undef UMI_GN_ACC
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

// ---- forward: y = [relu]((x - mean) * rstd * gamma + beta [+ res]), statistics finished in the prologue -------------------
// grid (N, S2); ws = stage-1 rows [N][S][2][C].  Workgroups with blockIdx.y == 0 also store mean / rstd for the backward.
__global__ __launch_bounds__(256) void gn_apply_fin_kernel(const half_t* __restrict__ x, int ldx, const float* __restrict__ ws,
                                                           int S, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const half_t* __restrict__ res, int ldr,
                                                           half_t* __restrict__ y, int ldy, int relu, long HW, int C, int G,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out, int ROWS) {
    __shared__ double chs[2][1024];                    // per-channel sums over the row blocks (C <= 1024: shape_ok)
    __shared__ float shf[1024];                        // per-channel shift of those sums (see gn_rowsum_v8)
    __shared__ float gm[1024], gr[1024];               // per-group mean / rstd (G <= C: the projection shortcuts' GroupNorm(C, C))
    const int tid = threadIdx.x, n = blockIdx.x, Cg = C / G;
    for (int c = tid; c < C; c += 256) {
        double sm = 0.0, q = 0.0;
#pragma unroll 8
        for (int s2 = 0; s2 < S; ++s2) {
            sm += (double)ws[(((long)n * S + s2) * 2 + 0) * C + c];
            q += (double)ws[(((long)n * S + s2) * 2 + 1) * C + c];
        }
        chs[0][c] = sm;
        chs[1][c] = q;
        shf[c] = (float)x[(long)n * HW * ldx + c];
    }
    __syncthreads();
    for (int g = tid; g < G; g += 256) {
        const double cnt = (double)HW * Cg;
        double sm = 0.0;
        for (int c = g * Cg; c < (g + 1) * Cg; ++c) sm += chs[0][c] + (double)HW * (double)shf[c];
        const double m = sm / cnt;
        double q = 0.0;                                 // sum_p (x - m)^2 = S2 - 2 d S1 + HW d^2 per channel, d = m - shift
        for (int c = g * Cg; c < (g + 1) * Cg; ++c) {
            const double d = m - (double)shf[c];
            q += chs[1][c] - 2.0 * d * chs[0][c] + (double)HW * d * d;
        }
        double var = q / cnt;
        if (var < 0.0) var = 0.0;
        const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)eps));
        gm[g] = mf;
        gr[g] = rf;
        if (blockIdx.y == 0) { mean_out[n * G + g] = mf; rstd_out[n * G + g] = rf; }
    }
    __syncthreads();
    const int G8 = C >> 3, PL = 256 / G8;
    const int cg = tid % G8, pl = tid / G8;
    float m[8], rr[8], gg[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        m[j] = gm[c / Cg];
        rr[j] = gr[c / Cg];
        gg[j] = gamma[c];
        bb[j] = beta[c];
    }
    const long r0 = (long)blockIdx.y * ROWS;
    long r1 = r0 + ROWS;
    if (r1 > HW) r1 = HW;
#define UMI_GN_APPLY(xv_, rv_, row_)                                                                               \
    do {                                                                                                          \
        half8 o;                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                 /* gn_apply8_kernel's expression */        \
            float v = ((float)xv_[j] - m[j]) * rr[j] * gg[j] + bb[j];                                             \
            if (res) v += (float)rv_[j];                                                                          \
            if (relu) v = fmaxf(v, 0.f);                                                                          \
            o[j] = (half_t)v;                                                                                     \
        }                                                                                                         \
        *reinterpret_cast<half8*>(y + (row_) * ldy + cg * 8) = o;                                                 \
    } while (0)
    long r = r0 + pl;
    for (; r + 3 * PL < r1; r += 4 * PL) {
        half8 xv[4], rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long row = (long)n * HW + r + u * PL;
            xv[u] = *reinterpret_cast<const half8*>(x + row * ldx + cg * 8);
            if (res) rv[u] = *reinterpret_cast<const half8*>(res + row * ldr + cg * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) UMI_GN_APPLY(xv[u], rv[u], (long)n * HW + r + u * PL);
    }
    for (; r < r1; r += PL) {
        const long row = (long)n * HW + r;
        half8 xv = *reinterpret_cast<const half8*>(x + row * ldx + cg * 8), rv;
        if (res) rv = *reinterpret_cast<const half8*>(res + row * ldr + cg * 8);
        UMI_GN_APPLY(xv, rv, row);
    }
#undef UMI_GN_APPLY
}

// ---- backward: dx = rstd * (dz*gamma - mean_g(dz*gamma) - xhat * mean_g(dz*gamma*xhat)), dres = dz -------------------------
// Prologue: this sample's stage-1 rows -> per-channel (sum dz*xhat, sum dz) -> per-group sums weighted by gamma.  Workgroups
// with blockIdx.y == 0 store the per-channel rows to part[n][2][C]: dgamma / dbeta are their sums over the samples.
__global__ __launch_bounds__(256) void gn_bwd_apply_fin_kernel(const half_t* __restrict__ dy, int lddy, const half_t* __restrict__ y,
                                                               int ldy, const half_t* __restrict__ x, int ldx,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ ws, int S,
                                                               int relu, half_t* __restrict__ dx, int lddx, half_t* __restrict__ dres,
                                                               int lddr, long HW, int C, int G, float* __restrict__ part, int ROWS) {
    __shared__ float chs[2][1024];
    __shared__ float gq[2][1024];
    const int tid = threadIdx.x, n = blockIdx.x, Cg = C / G;
    for (int i = tid; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        float sacc = 0.f;
#pragma unroll 8
        for (int s2 = 0; s2 < S; ++s2) sacc += ws[(((long)n * S + s2) * 2 + which) * C + c];       // (gn_bwd_part's order)
        chs[which][c] = sacc;
        if (blockIdx.y == 0) part[(long)n * 2 * C + i] = sacc;
    }
    __syncthreads();
    const float invm = 1.f / (float)(HW * Cg);
    for (int g = tid; g < G; g += 256) {
        float t1 = 0.f, t2 = 0.f;
        for (int c = g * Cg; c < (g + 1) * Cg; ++c) { t1 += chs[1][c] * gamma[c]; t2 += chs[0][c] * gamma[c]; }
        gq[0][g] = t1 * invm;
        gq[1][g] = t2 * invm;
    }
    __syncthreads();
    const int G8 = C >> 3, PL = 256 / G8;
    const int cg = tid % G8, pl = tid / G8;
    float m[8], rr[8], q1[8], q2[8], gg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j, g = c / Cg;
        m[j] = mean[n * G + g];
        rr[j] = rstd[n * G + g];
        q1[j] = gq[0][g];
        q2[j] = gq[1][g];
        gg[j] = gamma[c];
    }
    const long r0 = (long)blockIdx.y * ROWS;
    long r1 = r0 + ROWS;
    if (r1 > HW) r1 = HW;
#define UMI_GN_BAPPLY(gv_, xv_, yv_, row_)                                                                         \
    do {                                                                                                          \
        half8 o, dz8;                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                 /* gn_bwd_apply8_kernel's expression */    \
            float dz = (float)gv_[j];                                                                             \
            if (relu && !((float)yv_[j] > 0.f)) dz = 0.f;                                                         \
            const float xh = ((float)xv_[j] - m[j]) * rr[j];                                                      \
            o[j] = (half_t)(rr[j] * (dz * gg[j] - q1[j] - xh * q2[j]));                                           \
            dz8[j] = (half_t)dz;                                                                                  \
        }                                                                                                         \
        *reinterpret_cast<half8*>(dx + (row_) * lddx + cg * 8) = o;                                               \
        if (dres) *reinterpret_cast<half8*>(dres + (row_) * lddr + cg * 8) = dz8;                                 \
    } while (0)
    long r = r0 + pl;
    for (; r + PL < r1; r += 2 * PL) {
        half8 gv[2], xv[2], yv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long row = (long)n * HW + r + u * PL;
            gv[u] = *reinterpret_cast<const half8*>(dy + row * lddy + cg * 8);
            xv[u] = *reinterpret_cast<const half8*>(x + row * ldx + cg * 8);
            if (relu) yv[u] = *reinterpret_cast<const half8*>(y + row * ldy + cg * 8);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) UMI_GN_BAPPLY(gv[u], xv[u], yv[u], (long)n * HW + r + u * PL);
    }
    for (; r < r1; r += PL) {
        const long row = (long)n * HW + r;
        half8 gv = *reinterpret_cast<const half8*>(dy + row * lddy + cg * 8);
        half8 xv = *reinterpret_cast<const half8*>(x + row * ldx + cg * 8), yv;
        if (relu) yv = *reinterpret_cast<const half8*>(y + row * ldy + cg * 8);
        UMI_GN_BAPPLY(gv, xv, yv, row);
    }
#undef UMI_GN_BAPPLY
}

// dgamma / dbeta of up to 16 normalisation layers per launch: out[c] = scale * sum over the N rows of part[n][which][c]
// (fp64; rows split over 4 lanes in a fixed pattern).  GroupNorm: N = samples; LayerNorm: N = its kernel's partial rows.
// blockIdx.y = layer, 64 channels x 4 row lanes per workgroup.
struct GnPg { const float* part[16]; float* dgamma[16]; float* dbeta[16]; int C[16]; };
__global__ __launch_bounds__(256) void gn_param_grads_kernel(GnPg t, int N, float scale) {
    __shared__ double sh[4][64];
    const int l = blockIdx.y, C = t.C[l];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + cl;                      // index into the 2*C outputs (which = i / C)
    const int which = i < 2 * C ? i / C : 0, c = i < 2 * C ? i - which * C : 0;
    double a = 0.0;
    if (i < 2 * C) {
#pragma unroll 4
        for (int n = rl; n < N; n += 4) a += (double)t.part[l][((long)n * 2 + which) * C + c];
    }
    sh[rl][cl] = a;
    __syncthreads();
    if (rl == 0 && i < 2 * C)
        (which ? t.dbeta[l] : t.dgamma[l])[c] = (float)((sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]) * (double)scale);
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
bool shape_ok(int C, int G) {
    if (C % 8 || C % G || C > 1024) return false;           // (LDS tables of the apply kernels' prologues)
    const int G8 = C / 8;
    return G8 <= 256 && 256 % G8 == 0;
}

}  // namespace

int umi_gn_splits(int N, long HW) { const int r = gn_rows(N, HW); return (int)((HW + r - 1) / r); }

// forward, both launches; false when the shape does not qualify (the caller falls back to the generic kernels).
// ws: [N][S][2][C] floats.
bool umi_gn_fwd_f16v(const void* x, int ldx, const float* gamma, const float* beta, const void* res, int ldr, void* y, int ldy,
                     float* mean, float* rstd, int relu, int N, long HW, int C, int G, float eps, float* ws, hipStream_t s) {
    if (!shape_ok(C, G) || ldx % 8 || ldy % 8 || (res && ldr % 8) || !al16(x) || !al16(y) || (res && !al16(res))) return false;
    const int S = umi_gn_splits(N, HW), rows = gn_rows(N, HW);
    hipLaunchKernelGGL(gn_rowsum_v8<0>, dim3(N, S), dim3(256), 0, s, (const half_t*)x, ldx, (const half_t*)nullptr, 0,
                       (const half_t*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, 0, HW, C, G, S, ws, rows);
    hipLaunchKernelGGL(gn_apply_fin_kernel, dim3(N, S), dim3(256), 0, s, (const half_t*)x, ldx, (const float*)ws, S, eps, gamma, beta,
                       (const half_t*)res, ldr, (half_t*)y, ldy, relu, HW, C, G, mean, rstd, rows);
    return true;
}

// backward, both launches: dx (and dres), and the per-sample rows part[N][2][C] whose sums over N are dgamma / dbeta
bool umi_gn_bwd_f16v(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx, const float* mean,
                     const float* rstd, const float* gamma, int relu, void* dx, int lddx, void* dres, int lddr, int N, long HW,
                     int C, int G, float* part, float* ws, hipStream_t s) {
    if (!shape_ok(C, G) || ldx % 8 || lddy % 8 || ldy % 8 || lddx % 8 || (dres && lddr % 8) || !al16(x) || !al16(dy) || !al16(y) ||
        !al16(dx) || (dres && !al16(dres)))
        return false;
    const int S = umi_gn_splits(N, HW), rows = gn_rows(N, HW);
    hipLaunchKernelGGL(gn_rowsum_v8<1>, dim3(N, S), dim3(256), 0, s, (const half_t*)x, ldx, (const half_t*)dy, lddy,
                       (const half_t*)y, ldy, mean, rstd, relu, HW, C, G, S, ws, rows);
    hipLaunchKernelGGL(gn_bwd_apply_fin_kernel, dim3(N, S), dim3(256), 0, s, (const half_t*)dy, lddy, (const half_t*)y, ldy,
                       (const half_t*)x, ldx, mean, rstd, gamma, (const float*)ws, S, relu, (half_t*)dx, lddx, (half_t*)dres, lddr,
                       HW, C, G, part, rows);
    return true;
}

void umi_gn_param_grads_launch(int n, const float* const* parts, const int* Cs, int N, float* const* dgammas, float* const* dbetas,
                               float scale, hipStream_t s) {
    for (int g0 = 0; g0 < n; g0 += 16) {
        const int cnt = n - g0 < 16 ? n - g0 : 16;
        GnPg t;
        int maxC = 0;
        for (int i = 0; i < 16; ++i) {
            const int j = g0 + (i < cnt ? i : 0);
            t.part[i] = parts[j]; t.dgamma[i] = dgammas[j]; t.dbeta[i] = dbetas[j]; t.C[i] = Cs[j];
            if (Cs[j] > maxC) maxC = Cs[j];
        }
        hipLaunchKernelGGL(gn_param_grads_kernel, dim3((2 * maxC + 63) / 64, cnt), dim3(256), 0, s, t, N, scale);
    }
}
