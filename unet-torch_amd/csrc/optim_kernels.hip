// Multi-tensor parameter update and weight re-pack: one launch per training step each instead of one per tensor.
//
//   umi_optim_sgd_multi / umi_optim_adam_multi   torch.optim.SGD / Adam arithmetic (reference train.py:341-347: SGD lr 0.01
//       momentum 0.9 wd 1e-4, Adam lr 5e-4 wd 1e-4), fp32, same operation order and roundings as torch's foreach kernels.
//   umi_pack_kn_multi                            umi_pack_kn / umi_pack_kn8 of every convolution weight of a model.
//
// All three walk a descriptor table that lives in device memory.  A workgroup owns one fixed-size block of ONE tensor
// (`blk0` = first block of a descriptor, found by binary search), so the grid is balanced however unequal the tensors
// are (a U-Net has 3x3x1024x1024 weights next to 64-element BatchNorm vectors).  HBM-bound streaming work: 16-byte
// accesses whenever the four pointers of a tensor allow it.
#include "common.h"
#include <stdint.h>

namespace {

struct OptDesc {           // mirrors umi_optim_desc (include/unetmi.h)
    float* p;
    const float* g;
    float* s0;
    float* s1;
    long n;
    int blk0;
    int pad_;
};

struct PackDesc {          // mirrors umi_pack_desc
    const float* src;
    void* dst;
    long st, sk, sn;
    int T, K, N, flip_t, Kpad, Npad, k8, blk0;
    int ldn, pad_;         // ldn: row length of the destination when this entry fills a column slice of a wider matrix (0 = Npad)
};

constexpr int OPT_BLOCK = 4096;      // elements per workgroup (256 threads x 4 x float4)
constexpr int PACK_BLOCK = 2048;     // packed elements per workgroup

template <typename D>
__device__ inline int find_desc(const D* d, int n, int blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (d[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Device-resident hyper-parameters of one param group (mirrors umi_optim_hyper, include/unetmi.h): what a captured HIP graph
// must NOT freeze.  The doubles are the state; the floats are what the update kernel reads, derived by hyper_pre_kernel
// with the roundings torch applies to its Python doubles.
struct Hyper {
    double lr, base_lr, iter, max_iter, power, adam_t, beta1, beta2;
    float lr_f, step_size_f, bc2_sqrt_f, pad_;
    double pad2_[2];
};
static_assert(sizeof(Hyper) == 96, "umi_optim_hyper is 96 bytes");

struct SgdArgs { float lr, momentum, omd, wd; int nesterov, first; const Hyper* hp; };   // omd = 1 - dampening

__device__ inline void sgd_one(float& p, float g, float& m, const SgdArgs a) {
    if (a.wd != 0.f) g = fmaf(a.wd, p, g);                            // grad.add(param, alpha=wd)
    if (a.momentum != 0.f) {
        if (a.first) m = g;                                            // buf = clone(grad)
        else {
            float mm = __fmul_rn(m, a.momentum);                       // buf.mul_(momentum)
            m = a.omd == 1.f ? __fadd_rn(mm, g) : fmaf(a.omd, g, mm);   // .add_(grad, alpha=1-dampening)
        }
        g = a.nesterov ? fmaf(a.momentum, m, g) : m;
    }
    p = fmaf(-a.lr, g, p);                                             // param.add_(grad, alpha=-lr)
}

struct AdamArgs { float step_size, omb1, beta2, omb2, bc2_sqrt, eps, wd; const Hyper* hp; };   // omb = 1 - beta, rounded from double

__device__ inline void adam_one(float& p, float g, float& m, float& v, const AdamArgs a) {
    if (a.wd != 0.f) g = fmaf(a.wd, p, g);                            // grad.add(param, alpha=wd)  (L2, not AdamW)
    m = fmaf(a.omb1, g - m, m);                                        // exp_avg.lerp_(grad, 1-beta1)
    v = fmaf(a.omb2 * g, g, __fmul_rn(v, a.beta2));                   // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = __fdiv_rn(__fsqrt_rn(v), a.bc2_sqrt) + a.eps;
    p = fmaf(-a.step_size, __fdiv_rn(m, denom), p);                    // param.addcdiv_(exp_avg, denom, value=-step_size)
}

template <bool ADAM, typename A>
__global__ __launch_bounds__(256) void optim_multi_kernel(const OptDesc* __restrict__ descs, int n_desc, A a) {
    if (a.hp) {                               // learning rate / bias corrections from device memory (graph-safe)
        if constexpr (ADAM) { a.step_size = a.hp->step_size_f; a.bc2_sqrt = a.hp->bc2_sqrt_f; }
        else a.lr = a.hp->lr_f;
    }
    const int blk = blockIdx.x;
    const OptDesc d = descs[find_desc(descs, n_desc, blk)];
    const long base = (long)(blk - d.blk0) * OPT_BLOCK;
    const long left = d.n - base;
    const int cnt = left < OPT_BLOCK ? (int)left : OPT_BLOCK;
    float* p = d.p + base;
    const float* g = d.g + base;
    float* s0 = d.s0 ? d.s0 + base : nullptr;
    float* s1 = d.s1 ? d.s1 + base : nullptr;
    const unsigned long al = (unsigned long)p | (unsigned long)g | (unsigned long)s0 | (unsigned long)s1;
    if ((al & 15) == 0) {
        for (int i = threadIdx.x * 4; i + 4 <= cnt; i += 1024) {
            float4 pv = *reinterpret_cast<float4*>(p + i);
            const float4 gv = *reinterpret_cast<const float4*>(g + i);
            float4 mv = s0 ? *reinterpret_cast<float4*>(s0 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 vv = (ADAM && s1) ? *reinterpret_cast<float4*>(s1 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (ADAM) {
                adam_one(pv.x, gv.x, mv.x, vv.x, a); adam_one(pv.y, gv.y, mv.y, vv.y, a);
                adam_one(pv.z, gv.z, mv.z, vv.z, a); adam_one(pv.w, gv.w, mv.w, vv.w, a);
                *reinterpret_cast<float4*>(s1 + i) = vv;
            } else {
                sgd_one(pv.x, gv.x, mv.x, a); sgd_one(pv.y, gv.y, mv.y, a);
                sgd_one(pv.z, gv.z, mv.z, a); sgd_one(pv.w, gv.w, mv.w, a);
            }
            *reinterpret_cast<float4*>(p + i) = pv;
            if (s0) *reinterpret_cast<float4*>(s0 + i) = mv;
        }
        const int tail0 = cnt & ~3;
        const int i = tail0 + threadIdx.x;
        if (i < cnt) {
            float pv = p[i], mv = s0 ? s0[i] : 0.f;
            if constexpr (ADAM) { float vv = s1[i]; adam_one(pv, g[i], mv, vv, a); s1[i] = vv; }
            else sgd_one(pv, g[i], mv, a);
            p[i] = pv;
            if (s0) s0[i] = mv;
        }
    } else {
        for (int i = threadIdx.x; i < cnt; i += 256) {
            float pv = p[i], mv = s0 ? s0[i] : 0.f;
            if constexpr (ADAM) { float vv = s1[i]; adam_one(pv, g[i], mv, vv, a); s1[i] = vv; }
            else sgd_one(pv, g[i], mv, a);
            p[i] = pv;
            if (s0) s0[i] = mv;
        }
    }
}

// (a variant with one thread per 8 outputs of the k8 layout -- coalesced reads for the data-gradient layouts -- measured
// slower: 227 vs 173 us per launch on a TransUNet's 105 M parameters; the launch moves 1.26 GB, 0.31 ms at 4 TB/s)
// k8 layouts of matrices with T <= 9 taps go through an LDS tile of 32 n x 32 k x T (round 3): the source is read in ITS order
// (runs of 32 T consecutive floats along whichever of k / n is its inner index: OIHW conv weights, [out][in] linears and their
// transposes alike), the destination is written in 512-byte runs (32 n x 8 k per tap and k-group).  One element per thread in
// destination order -- the path below -- reads 4 bytes per 36-byte stride at best and a different line per lane at worst
// (1.26 GB in 0.50 ms on the TransUNet's 105 M parameters, 0.18 ms on the U-Net's 31 M).
constexpr int PT = 32;
template <typename T, int U>
__device__ __forceinline__ void pack_stage(const PackDesc& d, T* tl, int k0, int n0, bool k_inner) {
    const int TT = d.T;
    for (int e0 = threadIdx.x; e0 < PT * PT * TT; e0 += 256 * U) {
        float v[U];
        int li[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 256 * u;
            const int o_ = e / (PT * TT), r = e - o_ * (PT * TT);
            const int i_ = r / TT, t = r - i_ * TT;
            const int kl = k_inner ? i_ : o_, nl = k_inner ? o_ : i_;
            const int k = k0 + kl, n = n0 + nl;
            ok[u] = e < PT * PT * TT && k < d.K && n < d.N;
            li[u] = e < PT * PT * TT ? (t * PT + kl) * (PT + 2) + nl : -1;
            // unconditional loads from clamped (in-range) addresses, selected afterwards: `ok ? src[..] : 0` compiles to a
            // branch per load with a wait of its own, i.e. ONE load in flight per thread instead of U
            const int tc = t < TT ? t : TT - 1, kc = k < d.K ? k : d.K - 1, nc = n < d.N ? n : d.N - 1;
            v[u] = d.src[(long)(d.flip_t ? TT - 1 - tc : tc) * d.st + (long)kc * d.sk + (long)nc * d.sn];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) asm volatile("" : "+v"(v[u]));       // (keeps hipcc from sinking the loads under `ok`)
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ok[u] ? v[u] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (li[u] >= 0) tl[li[u]] = (T)v[u];
    }
}

template <typename T>
__device__ void pack_tiles_k8(const PackDesc& d, int my_blk, int n_blk) {
    __shared__ T tl[9 * PT * (PT + 2)];                       // [t][k][n], row pad 2
    const int nkt = (d.Kpad + PT - 1) / PT, nnt = (d.Npad + PT - 1) / PT;
    const int kb8 = d.Kpad >> 3, ldn = d.ldn ? d.ldn : d.Npad, TT = d.T;
    const bool k_inner = d.sk <= d.sn;                        // the source's inner index among (k, n)
    T* dst = (T*)d.dst;
    for (int tile = my_blk; tile < nkt * nnt; tile += n_blk) {
        const int k0 = (tile % nkt) * PT, n0 = (tile / nkt) * PT;
        // U loads in flight per thread (a one-load loop runs at the memory latency: 36 trips per tile at T = 9): 12 for the
        // 3 x 3 convs' tiles (three trips), 4 otherwise (a T = 1 tile is one trip of four)
        if (TT == 9) pack_stage<T, 12>(d, tl, k0, n0, k_inner);
        else pack_stage<T, 4>(d, tl, k0, n0, k_inner);
        __syncthreads();
        // pieces of 8 k for one (t, k-group, n): 16 bytes (fp16) each, 32 n adjacent
        for (int e = threadIdx.x; e < TT * (PT / 8) * PT; e += 256) {
            const int nl = e % PT, r = e / PT, kg = r % (PT / 8), t = r / (PT / 8);
            const int k = k0 + kg * 8, n = n0 + nl;
            if (k >= d.Kpad || n >= d.Npad) continue;
            T pk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[j] = tl[(t * PT + kg * 8 + j) * (PT + 2) + nl];
            T* o = dst + (((long)t * kb8 + (k >> 3)) * ldn + n) * 8;
            typedef T vec8 __attribute__((ext_vector_type(8)));
            vec8 pv;
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = pk[j];
            *reinterpret_cast<vec8*>(o) = pv;
        }
        __syncthreads();
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const PackDesc* __restrict__ descs, int n_desc) {
    const int blk = blockIdx.x;
    const PackDesc d = descs[find_desc(descs, n_desc, blk)];
    const long total = (long)d.T * d.Kpad * d.Npad;
    if (d.k8 && d.T <= 9 && sizeof(T) == 2) {                 // (uniform per workgroup)
        pack_tiles_k8<T>(d, blk - d.blk0, (int)((total + PACK_BLOCK - 1) / PACK_BLOCK));
        return;
    }
    const long base = (long)(blk - d.blk0) * PACK_BLOCK;
    T* dst = (T*)d.dst;
    const int kb8 = d.Kpad >> 3;
    const int ldn = d.ldn ? d.ldn : d.Npad;
#pragma unroll 2
    for (int j = 0; j < PACK_BLOCK / 256; ++j) {
        const long i = base + j * 256 + threadIdx.x;
        if (i >= total) break;
        int n, k, t;
        long o;                          // destination index (== i unless the entry is a column slice: ldn > Npad)
        if (d.k8) {                      // dst[t][k/8][n][k%8]
            const int k8 = (int)(i & 7);
            long r = i >> 3;
            n = (int)(r % d.Npad); r /= d.Npad;
            const int kb = (int)(r % kb8);
            t = (int)(r / kb8);
            k = kb * 8 + k8;
            o = (((long)t * kb8 + kb) * ldn + n) * 8 + k8;
        } else {                         // dst[t][k][n]
            n = (int)(i % d.Npad);
            const long r = i / d.Npad;
            k = (int)(r % d.Kpad);
            t = (int)(r / d.Kpad);
            o = ((long)t * d.Kpad + k) * ldn + n;
        }
        float v = 0.f;
        if (k < d.K && n < d.N) {
            const int ts = d.flip_t ? (d.T - 1 - t) : t;
            v = d.src[ts * d.st + k * d.sk + n * d.sn];
        }
        dst[o] = (T)v;
    }
}

}  // namespace

// Descriptor-table upload as a KERNEL: `src` is pinned host memory (device-mapped), `dst` the device copy the update /
// pack kernels read.  Inside a HIP-graph capture this is an ordinary kernel node that re-reads the host buffer on every
// replay; a captured host-to-device memcpy of a torch pinned buffer did not replay reliably on ROCm 7.2 (the replayed
// optimizer step saw a stale table).
__global__ __launch_bounds__(256) void table_upload_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
}
extern "C" int umi_table_upload(const void* host_pinned, void* dev, size_t nbytes, umi_stream_t stream) {
    if (!host_pinned || !dev || nbytes == 0 || (nbytes & 15) || (((uintptr_t)host_pinned | (uintptr_t)dev) & 15)) return UMI_ERR_BADARG;
    const int n16 = (int)(nbytes / 16);
    int grid = (n16 + 255) / 256;
    if (grid > 64) grid = 64;
    hipLaunchKernelGGL(table_upload_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)host_pinned, (uint4*)dev, n16);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_optim_block_elems(void) { return OPT_BLOCK; }
extern "C" int umi_pack_block_elems(void) { return PACK_BLOCK; }

// Hyper-parameters arrive as doubles and are rounded to fp32 exactly where torch rounds them (1 - dampening and 1 - beta
// are formed in double first: 1 - 0.999 is 0.001f there, not 1.f - 0.999f).
extern "C" int umi_optim_sgd_multi(const void* descs, int n_desc, int total_blocks, double lr, double momentum,
                                   double dampening, double weight_decay, int nesterov, int first_step,
                                   umi_stream_t stream) {
    if (!descs || n_desc <= 0 || total_blocks <= 0) return UMI_ERR_BADARG;
    SgdArgs a{(float)lr, (float)momentum, (float)(1.0 - dampening), (float)weight_decay, nesterov, first_step, nullptr};
    hipLaunchKernelGGL((optim_multi_kernel<false, SgdArgs>), dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const OptDesc*)descs, n_desc, a);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_optim_adam_multi(const void* descs, int n_desc, int total_blocks, double step_size, double beta1,
                                    double beta2, double bc2_sqrt, double eps, double weight_decay, umi_stream_t stream) {
    if (!descs || n_desc <= 0 || total_blocks <= 0) return UMI_ERR_BADARG;
    AdamArgs a{(float)step_size, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)bc2_sqrt, (float)eps,
               (float)weight_decay, nullptr};
    hipLaunchKernelGGL((optim_multi_kernel<true, AdamArgs>), dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const OptDesc*)descs, n_desc, a);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// ---- device-resident hyper-parameters (graph-safe learning rate / Adam step count) ------------------------------------
__global__ void hyper_pre_kernel(Hyper* h, int adam) {
    if (adam) {                                   // torch.optim.Adam: step += 1; bias corrections in double, then rounded
        const double t = h->adam_t + 1.0;
        h->adam_t = t;
        const double bc1 = 1.0 - pow(h->beta1, t), bc2 = 1.0 - pow(h->beta2, t);
        h->step_size_f = (float)(h->lr / bc1);
        h->bc2_sqrt_f = (float)sqrt(bc2);
    }
    h->lr_f = (float)h->lr;
}
// reference Trainer.py:722-726: lr = base_lr * (1 - iter_num / max_iterations) ** 0.9 with the PRE-increment iter_num, written
// after the optimizer step; then iter_num += 1
__global__ void hyper_poly_kernel(Hyper* h) {
    h->lr = h->base_lr * pow(1.0 - h->iter / h->max_iter, h->power);
    h->iter += 1.0;
}

extern "C" size_t umi_optim_hyper_bytes(void) { return sizeof(Hyper); }

extern "C" int umi_optim_hyper_pre(void* hyper, int adam, umi_stream_t stream) {
    if (!hyper || ((uintptr_t)hyper & 7)) return UMI_ERR_BADARG;
    hipLaunchKernelGGL(hyper_pre_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (Hyper*)hyper, adam);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_optim_hyper_poly(void* hyper, umi_stream_t stream) {
    if (!hyper || ((uintptr_t)hyper & 7)) return UMI_ERR_BADARG;
    hipLaunchKernelGGL(hyper_poly_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (Hyper*)hyper);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_optim_sgd_multi_dev(const void* descs, int n_desc, int total_blocks, const void* hyper, double momentum,
                                       double dampening, double weight_decay, int nesterov, int first_step,
                                       umi_stream_t stream) {
    if (!descs || !hyper || n_desc <= 0 || total_blocks <= 0) return UMI_ERR_BADARG;
    SgdArgs a{0.f, (float)momentum, (float)(1.0 - dampening), (float)weight_decay, nesterov, first_step, (const Hyper*)hyper};
    hipLaunchKernelGGL((optim_multi_kernel<false, SgdArgs>), dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const OptDesc*)descs, n_desc, a);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_optim_adam_multi_dev(const void* descs, int n_desc, int total_blocks, const void* hyper, double beta1,
                                        double beta2, double eps, double weight_decay, umi_stream_t stream) {
    if (!descs || !hyper || n_desc <= 0 || total_blocks <= 0) return UMI_ERR_BADARG;
    AdamArgs a{0.f, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), 1.f, (float)eps, (float)weight_decay,
               (const Hyper*)hyper};
    hipLaunchKernelGGL((optim_multi_kernel<true, AdamArgs>), dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const OptDesc*)descs, n_desc, a);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

extern "C" int umi_pack_kn_multi(const void* descs, int n_desc, int total_blocks, int dtype, umi_stream_t stream) {
    if (!descs || n_desc <= 0 || total_blocks <= 0) return UMI_ERR_BADARG;
    if (dtype == UMI_F16)
        hipLaunchKernelGGL((pack_multi_kernel<half_t>), dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                           (const PackDesc*)descs, n_desc);
    else if (dtype == UMI_F32)
        hipLaunchKernelGGL((pack_multi_kernel<float>), dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                           (const PackDesc*)descs, n_desc);
    else return UMI_ERR_BADARG;
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
