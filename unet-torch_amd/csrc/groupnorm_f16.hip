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

// Backward (MODE 1): the later reduction stages folded into the first launch.  The workgroup that finishes LAST for a sample (a
// ticket counter per sample, zero on entry and left zero) combines that sample's partial rows in a fixed order, so the
// result does not depend on which workgroup it is; the one that completes the last sample also folds the per-sample rows
// into dgamma / dbeta: 2 launches per GroupNorm backward instead of 5 (27.5 vs 32 us of kernel time on the R50 hybrid's
// layers).  tickets == nullptr: stage 1 only (separate launches follow).  The forward statistics keep their separate 5-us
// finalize launch: measured, the same fold costs more there (16.4 us against 6.4 + 4.9) -- the tail runs on one workgroup
// per sample while the chip idles, which is what a tiny launch inside a HIP graph costs anyway.
struct GnFin {
    int* tickets;                                     // [N + 1]
    const float* gamma; float* part; float* gsum; float* dgamma; float* dbeta; float out_scale; int N;
};
// Hand-over of the partial rows between workgroups WITHOUT a device-scope release fence: on gfx950 that fence is a write-back
// of the whole L2 of the XCD (buffer_wbl2), executed here by ~1,000 workgroups per launch next to kernels that have just
// written tens of MB -- measured 21.1 -> 27.9 ms per TransUNet step.  Instead the rows are written and read with
// device-coherent (sc1) accesses, every thread waits for its stores to be acknowledged (vmcnt(0)) before the workgroup's
// barrier, and only then is the ticket taken: the same ordering the fence gives, for these addresses only.
__device__ inline float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// MODE 0: out = (sum x, sum x^2) per channel.   MODE 1: out = (sum dz*xhat, sum dz) per channel, dz = dy*[y>0 | 1]
template <int MODE>
__global__ __launch_bounds__(256) void gn_rowsum_v8(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ dy,
                                                    int lddy, const half_t* __restrict__ y, int ldy,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd, int relu,
                                                    long HW, int C, int G, int S, float* __restrict__ ws, int ROWS, GnFin fin) {
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
        if (fin.tickets) st_agent(ws + (((long)n * S + sp) * 2 + which) * C + c, s);
        else ws[(((long)n * S + sp) * 2 + which) * C + c] = s;
    }
    if (!fin.tickets) return;
    __shared__ int last;
    stores_done();                                     // this workgroup's partial rows have reached device coherence ...
    __syncthreads();
    if (tid == 0) last = atomicAdd(&fin.tickets[n], 1) == S - 1;      // ... before its ticket is taken
    __syncthreads();
    if (!last) return;
    // the one workgroup per sample that gets here may pay for an acquire (drops its stale cache lines): the rows are then read
    // with ordinary, pipelined loads
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float* wsr = ws;
    const int Cg = C / G;
    // The tail runs on ONE workgroup while the rest of the chip waits for it: what matters is the number of dependent memory
    // round trips, so every thread sums the splits of its own channels with 16 loads in flight, the per-channel results meet
    // in LDS (the stage-1 scratch, free by now) and the per-group step reads only LDS.
    float* chs = &red[0][0][0];                           // [2][C] floats (C <= 2048 fits the 18 KB)
    // gn_bwd_part + gn_bwd_gsum for sample n
    for (int i = tid; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        float sacc = 0.f;
#pragma unroll 16
        for (int s2 = 0; s2 < S; ++s2) sacc += wsr[(((long)n * S + s2) * 2 + which) * C + c];
        st_agent(fin.part + (long)n * 2 * C + i, sacc);
        chs[i] = sacc;
    }
    stores_done();
    __syncthreads();
    for (int g = tid; g < G; g += 256) {
        float t1 = 0.f, t2 = 0.f;
        for (int c = g * Cg; c < (g + 1) * Cg; ++c) {
            t1 += chs[C + c] * fin.gamma[c];
            t2 += chs[c] * fin.gamma[c];
        }
        fin.gsum[(n * G + g) * 2 + 0] = t1;
        fin.gsum[(n * G + g) * 2 + 1] = t2;
    }
    __syncthreads();
    if (tid == 0) {
        fin.tickets[n] = 0;
        last = atomicAdd(&fin.tickets[fin.N], 1) == fin.N - 1;
    }
    __syncthreads();
    if (!last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float* partr = fin.part;
    // dgamma / dbeta = out_scale * sum over samples of the per-sample rows (fixed order, fp64)
    for (int i = tid; i < 2 * C; i += 256) {
        const int which = i / C, c = i - which * C;
        double a2 = 0.0;
#pragma unroll 8
        for (int m = 0; m < fin.N; ++m) a2 += (double)partr[((long)m * 2 + which) * C + c];
        (which ? fin.dbeta : fin.dgamma)[c] = (float)(a2 * (double)fin.out_scale);
    }
    if (tid == 0) fin.tickets[fin.N] = 0;
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
    GnFin fin{};
    hipLaunchKernelGGL(gn_rowsum_v8<0>, dim3(N, S), dim3(256), 0, s, (const half_t*)x, ldx, (const half_t*)nullptr, 0,
                       (const half_t*)nullptr, 0, (const float*)nullptr, (const float*)nullptr, 0, HW, C, G, S, ws, gn_rows(N, HW), fin);
    hipLaunchKernelGGL(gn_stats_finalize, dim3((N * G + 3) / 4), dim3(256), 0, s, (const float*)ws, N, S, C, G, HW, eps,
                           mean, rstd);
    return true;
}

// fills gsum [N*G][2] and part [N][2][C] (the layouts gn_bwd_apply_kernel / reduce_rows2 expect); with tickets also
// dgamma / dbeta (the caller then skips its reduce_rows2 launch)
bool umi_gn_bwd_reduce_f16v(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx, const float* mean,
                            const float* rstd, const float* gamma, int relu, int N, long HW, int C, int G, float* gsum,
                            float* part, float* ws, int* tickets, float* dgamma, float* dbeta, float out_scale, hipStream_t s) {
    if (!shape_ok(C, G) || ldx % 8 || lddy % 8 || ldy % 8 || !al16(x) || !al16(dy) || !al16(y)) return false;
    const int S = umi_gn_splits(N, HW);
    GnFin fin{};
    fin.tickets = tickets; fin.gamma = gamma; fin.part = part; fin.gsum = gsum; fin.dgamma = dgamma; fin.dbeta = dbeta;
    fin.out_scale = out_scale; fin.N = N;
    hipLaunchKernelGGL(gn_rowsum_v8<1>, dim3(N, S), dim3(256), 0, s, (const half_t*)x, ldx, (const half_t*)dy, lddy,
                       (const half_t*)y, ldy, mean, rstd, relu, HW, C, G, S, ws, gn_rows(N, HW), fin);
    if (tickets) return true;
    const long np = (long)N * 2 * C;
    hipLaunchKernelGGL(gn_bwd_part, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, (const float*)ws, N, S, C, part);
    hipLaunchKernelGGL(gn_bwd_gsum, dim3((N * G + 255) / 256), dim3(256), 0, s, (const float*)part, gamma, N, C, G, gsum);
    return true;
}
