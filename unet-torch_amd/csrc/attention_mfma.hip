// Multi-head softmax attention on the gfx950 matrix cores (fp16 storage, fp32 softmax), head dim 64
// (reference TransUnet/vit_seg_modeling.py:73-91: softmax(Q K^T / sqrt(64)) V, 12 heads, 196 or 1024 tokens).
//
// Flash-style: scores are never materialised.  All products use v_mfma_f32_32x32x16_f16 with the "swapped" orientation
// S^T[key][query] = K . Q^T so that a lane owns ONE query column: its softmax row statistics are plain register
// reductions plus one exchange with lane^32, and the fp32 accumulator tile (rows = keys in registers) is re-used
// directly as the B operand of the next product (O^T = V^T . P^T, dQ^T = K^T . dS^T) without touching LDS.
// The A operand of those products is a transposed matrix (V^T, K^T, dO^T, Q^T): it is read with ds_read_b64_tr_b16 from
// [row][32-channel] 64-byte LDS rows, in the row order the accumulator-as-operand trick requires.
//
//   forward      : block = 128 queries (4 waves x 32) of one (batch, head); K/V streamed in 256-key chunks.
//   backward dQ  : same decomposition; also writes delta = rowsum(dO * O).
//   backward dKV : block = 128 keys (4 waves x 32) of one (batch, head); Q/dO streamed in 32-query tiles.
// Deterministic (no atomics): dQ and dK/dV come from two kernels that each recompute P from the saved log-sum-exp.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int D = 64;
constexpr int RROW = 144;          // LDS row bytes of a row-major [row][64 halfs] tile (+16 B pad: conflict-free b128)
constexpr int CROW = 64;           // LDS row bytes of one 32-channel chunk (transposing reads)

__device__ __forceinline__ float lane_xchg32(float v) { return __shfl_xor(v, 32); }

// fragment of a TRANSPOSED operand (rows = channels d, k = the 16 rows of `k-step` in accumulator-operand order):
// element j of lane half h is tile row 8*(j>>2) + 4*h + (j&3)  -> two transposing reads at rows 4h.. and 8+4h..
__device__ __forceinline__ half8 tr_frag_acc_order(const unsigned char* p) {
    typedef __attribute__((address_space(3))) short4v* lds_ptr;
    short4v r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    short4v r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 8 * CROW));
    half4 h0 = __builtin_bit_cast(half4, r0), h1 = __builtin_bit_cast(half4, r1);
    return __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// accumulator tile (rows in registers) -> the two B-operand fragments (k-steps 0 and 1) of the next product
__device__ __forceinline__ void acc_to_frags(const floatx16& a, half8& f0, half8& f1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { f0[j] = (half_t)a[j]; f1[j] = (half_t)a[8 + j]; }
}

// stage `rows` rows of a [*, H*64] token tensor (one head slice) into LDS: row-major padded copy and/or chunked copy
template <bool ROWMAJOR, bool CHUNKED>
__device__ __forceinline__ void stage_rows(const half_t* __restrict__ src, int ld, long row0, int rows, long row_limit,
                                           unsigned char* rm, unsigned char* ch, int ch_rows, int tid) {
    for (int i = tid; i < rows * 8; i += 256) {
        int r = i >> 3, piece = i & 7;
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (half_t)0.f;
        if (row0 + r < row_limit) v = *reinterpret_cast<const half8*>(src + (row0 + r) * ld + piece * 8);
        if (ROWMAJOR) *reinterpret_cast<half8*>(rm + r * RROW + piece * 16) = v;
        if (CHUNKED) *reinterpret_cast<half8*>(ch + (piece >> 2) * ch_rows * CROW + r * CROW + (piece & 3) * 16) = v;
    }
}

// keys per staged chunk: forward 224 (the 196-token sequence in one pass, 59.5 KB), dQ 128 (52 KB with both K layouts + V)
template <bool BWD> struct ChunkKeys { static constexpr int value = BWD ? 128 : 224; };

// ------------------------------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256, 2) void attn_q_side_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k,
                                                             const half_t* __restrict__ v, int ld,
                                                             half_t* __restrict__ o /*fwd: out; bwd: forward output (read)*/,
                                                             const half_t* __restrict__ dO, int ldo,
                                                             float* __restrict__ lse, half_t* __restrict__ dq, int lddq,
                                                             float* __restrict__ delta, int B, int N, int Hh, float scale) {
    constexpr int KC = ChunkKeys<BWD>::value;
    // LDS: K row-major [KC][RROW] (+ chunked copy for dQ), V: fwd chunked [2][KC][CROW]; bwd row-major [KC][RROW]
    __shared__ __attribute__((aligned(16))) unsigned char smem[KC * RROW + 2 * KC * CROW + (BWD ? KC * RROW : 0)];
    unsigned char* k_rm = smem;
    unsigned char* c_ch = smem + KC * RROW;                       // fwd: V chunked ; bwd: K chunked
    unsigned char* v_rm = smem + KC * RROW + 2 * KC * CROW;       // bwd only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
    const long tok0 = (long)b * N;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int lq = lane & 31, lh = lane >> 5;
    const int qi = q0 + lq;
    const bool qv = qi < N;
    const half_t* qh = q + h * D;
    const half_t* kh = k + h * D;
    const half_t* vh = v + h * D;

    // Q^T (and dO^T) B-operand fragments: lane = query column, k = 8 consecutive d of chunk dc
    half8 qf[4], dof[4];
    float dl = 0.f;
#pragma unroll
    for (int dc = 0; dc < 4; ++dc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { qf[dc][j] = (half_t)0.f; dof[dc][j] = (half_t)0.f; }
        if (qv) {
            qf[dc] = *reinterpret_cast<const half8*>(qh + (tok0 + qi) * ld + dc * 16 + lh * 8);
            if (BWD) {
                dof[dc] = *reinterpret_cast<const half8*>(dO + (tok0 + qi) * ldo + h * D + dc * 16 + lh * 8);
                half8 ov = *reinterpret_cast<const half8*>(o + (tok0 + qi) * ldo + h * D + dc * 16 + lh * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) dl = fmaf((float)dof[dc][j], (float)ov[j], dl);
            }
        }
    }
    float L = 0.f;
    if (BWD) {
        dl += lane_xchg32(dl);                                     // the two lane halves hold different d ranges of one query
        L = qv ? lse[(long)bh * N + qi] : 0.f;
    }

    floatx16 acc[2];                                               // fwd: O^T [d tile][.]; bwd: dQ^T
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float mx = -INFINITY, lsum = 0.f;

    const int g = lane >> 4, li = lane & 15, lqq = li >> 2, lp = li & 3;
    const int tr_lane = (4 * (g >> 1) + lqq) * CROW + (16 * (g & 1) + 4 * lp) * 2;   // + chunk*rows*CROW + (16*s + key0)*CROW
    const int a_lane = lq * RROW + lh * 16;                                           // + key0*RROW + dc*32

    for (int kc0 = 0; kc0 < N; kc0 += KC) {
        __syncthreads();
        if (BWD) {
            stage_rows<true, true>(kh, ld, tok0 + kc0, KC, tok0 + N, k_rm, c_ch, KC, tid);
            stage_rows<true, false>(vh, ld, tok0 + kc0, KC, tok0 + N, v_rm, nullptr, KC, tid);
        } else {
            stage_rows<true, false>(kh, ld, tok0 + kc0, KC, tok0 + N, k_rm, nullptr, KC, tid);
            stage_rows<false, true>(vh, ld, tok0 + kc0, KC, tok0 + N, nullptr, c_ch, KC, tid);
        }
        __syncthreads();
        const int kend = (N - kc0) < KC ? (N - kc0) : KC;
        for (int kt = 0; kt < kend; kt += 32) {
            // S^T tile [32 keys][32 queries]
            floatx16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int dc = 0; dc < 4; ++dc) {
                half8 af = *reinterpret_cast<const half8*>(k_rm + kt * RROW + a_lane + dc * 32);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, qf[dc], s, 0, 0, 0);
            }
            floatx16 p;
            if (!BWD) {
                float tmax = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int key = kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    s[r] = key < kend ? s[r] * scale : -INFINITY;
                    tmax = fmaxf(tmax, s[r]);
                }
                tmax = fmaxf(tmax, lane_xchg32(tmax));
                const float mn = fmaxf(mx, tmax);
                const float corr = __expf(mx - mn);
                float ps = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { p[r] = __expf(s[r] - mn); ps += p[r]; }
                ps += lane_xchg32(ps);
                lsum = lsum * corr + ps;
                mx = mn;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][r] *= corr;
            } else {
                // dP^T tile, then dS^T = P * (dP - delta)
                floatx16 dp;
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
                for (int dc = 0; dc < 4; ++dc) {
                    half8 af = *reinterpret_cast<const half8*>(v_rm + kt * RROW + a_lane + dc * 32);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, dof[dc], dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int key = kt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float pr = key < kend ? __expf(s[r] * scale - L) : 0.f;
                    p[r] = pr * (dp[r] - dl);
                }
            }
            half8 f0, f1;
            acc_to_frags(p, f0, f1);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                half8 t0 = tr_frag_acc_order(c_ch + mt * KC * CROW + (kt + 0) * CROW + tr_lane);
                half8 t1 = tr_frag_acc_order(c_ch + mt * KC * CROW + (kt + 16) * CROW + tr_lane);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t0, f0, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(t1, f1, acc[mt], 0, 0, 0);
            }
        }
    }
    if (qv) {
        const float fin = BWD ? scale : 1.f / lsum;
        half_t* dst = BWD ? dq + (tok0 + qi) * lddq + h * D : o + (tok0 + qi) * ldo + h * D;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                half4 hv;
#pragma unroll
                for (int j = 0; j < 4; ++j) hv[j] = (half_t)(acc[mt][gq * 4 + j] * fin);
                *reinterpret_cast<half4*>(dst + mt * 32 + gq * 8 + lh * 4) = hv;
            }
        if (lh == 0) {
            if (BWD) delta[(long)bh * N + qi] = dl;
            else lse[(long)bh * N + qi] = mx + __logf(lsum);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// dK / dV: wave = 32 keys (lane = key column of S[q][key]); Q and dO tiles of 32 queries are staged row-major (A operand of
// S = Q.K^T and dP = dO.V^T) and chunked (transposed A operand of dV^T = dO^T.P, dK^T = Q^T.dS).
__global__ __launch_bounds__(256, 2) void attn_kv_side_kernel(const half_t* __restrict__ q, const half_t* __restrict__ k,
                                                              const half_t* __restrict__ v, int ld,
                                                              const half_t* __restrict__ dO, int ldo,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              half_t* __restrict__ dk, half_t* __restrict__ dv, int lddk, int B,
                                                              int N, int Hh, float scale) {
    constexpr int QT = 32;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (QT * RROW + 2 * QT * CROW)];
    __shared__ float lse_s[QT], del_s[QT];
    unsigned char* q_rm = smem;
    unsigned char* q_ch = q_rm + QT * RROW;
    unsigned char* o_rm = q_ch + 2 * QT * CROW;
    unsigned char* o_ch = o_rm + QT * RROW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.y, b = bh / Hh, h = bh % Hh;
    const long tok0 = (long)b * N;
    const int k0 = blockIdx.x * 128 + wave * 32;
    const int lk = lane & 31, lh = lane >> 5;
    const int ki = k0 + lk;
    const bool kv = ki < N;
    half8 kf[4], vf[4];                                            // K^T / V^T B-operand fragments (lane = key column)
#pragma unroll
    for (int dc = 0; dc < 4; ++dc) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { kf[dc][j] = (half_t)0.f; vf[dc][j] = (half_t)0.f; }
        if (kv) {
            kf[dc] = *reinterpret_cast<const half8*>(k + (tok0 + ki) * ld + h * D + dc * 16 + lh * 8);
            vf[dc] = *reinterpret_cast<const half8*>(v + (tok0 + ki) * ld + h * D + dc * 16 + lh * 8);
        }
    }
    floatx16 dkT[2], dvT[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkT[a][r] = 0.f; dvT[a][r] = 0.f; }

    const int g = lane >> 4, li = lane & 15, lqq = li >> 2, lp = li & 3;
    const int tr_lane = (4 * (g >> 1) + lqq) * CROW + (16 * (g & 1) + 4 * lp) * 2;
    const int a_lane = lk * RROW + lh * 16;                        // A operand rows = queries: lane&31 = query row

    for (int q0 = 0; q0 < N; q0 += QT) {
        __syncthreads();
        stage_rows<true, true>(q + h * D, ld, tok0 + q0, QT, tok0 + N, q_rm, q_ch, QT, tid);
        stage_rows<true, true>(dO + h * D, ldo, tok0 + q0, QT, tok0 + N, o_rm, o_ch, QT, tid);
        if (tid < QT) {
            bool in = q0 + tid < N;
            lse_s[tid] = in ? lse[(long)bh * N + q0 + tid] : 0.f;
            del_s[tid] = in ? delta[(long)bh * N + q0 + tid] : 0.f;
        }
        __syncthreads();
        floatx16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
        for (int dc = 0; dc < 4; ++dc) {
            half8 aq = *reinterpret_cast<const half8*>(q_rm + a_lane + dc * 32);
            half8 ao = *reinterpret_cast<const half8*>(o_rm + a_lane + dc * 32);
            s = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq, kf[dc], s, 0, 0, 0);        // S[q][key]
            dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(ao, vf[dc], dp, 0, 0, 0);      // dP[q][key]
        }
        floatx16 p, ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int qq = (r & 3) + 8 * (r >> 2) + 4 * lh;                                   // query row of this register
            float pr = (q0 + qq < N && kv) ? __expf(s[r] * scale - lse_s[qq]) : 0.f;
            p[r] = pr;
            ds[r] = pr * (dp[r] - del_s[qq]) * scale;
        }
        half8 p0, p1, d0, d1;
        acc_to_frags(p, p0, p1);
        acc_to_frags(ds, d0, d1);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            half8 to0 = tr_frag_acc_order(o_ch + mt * QT * CROW + 0 * CROW + tr_lane);
            half8 to1 = tr_frag_acc_order(o_ch + mt * QT * CROW + 16 * CROW + tr_lane);
            dvT[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(to0, p0, dvT[mt], 0, 0, 0);
            dvT[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(to1, p1, dvT[mt], 0, 0, 0);
            half8 tq0 = tr_frag_acc_order(q_ch + mt * QT * CROW + 0 * CROW + tr_lane);
            half8 tq1 = tr_frag_acc_order(q_ch + mt * QT * CROW + 16 * CROW + tr_lane);
            dkT[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tq0, d0, dkT[mt], 0, 0, 0);
            dkT[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tq1, d1, dkT[mt], 0, 0, 0);
        }
    }
    if (kv) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                half4 hk, hv;
#pragma unroll
                for (int j = 0; j < 4; ++j) { hk[j] = (half_t)dkT[mt][gq * 4 + j]; hv[j] = (half_t)dvT[mt][gq * 4 + j]; }
                long off = (tok0 + ki) * lddk + h * D + mt * 32 + gq * 8 + lh * 4;
                *reinterpret_cast<half4*>(dk + off) = hk;
                *reinterpret_cast<half4*>(dv + off) = hv;
            }
    }
}

}  // namespace

bool umi_attn_mfma_ok(int D_, int ld, int ldo, int dtype, const void* a, const void* b_, const void* c) {
    return dtype == UMI_F16 && D_ == 64 && ld % 8 == 0 && ldo % 8 == 0 &&
           ((((uintptr_t)a) | ((uintptr_t)b_) | ((uintptr_t)c)) & 15) == 0;
}

int umi_attn_fwd_mfma(const void* q, const void* k, const void* v, int ld, void* o, int ldo, float* lse, int B, int N, int Hh,
                      hipStream_t s) {
    dim3 grid((N + 127) / 128, B * Hh), block(256);
    hipLaunchKernelGGL(attn_q_side_kernel<false>, grid, block, 0, s, (const half_t*)q, (const half_t*)k, (const half_t*)v, ld,
                       (half_t*)o, (const half_t*)nullptr, ldo, lse, (half_t*)nullptr, 0, (float*)nullptr, B, N, Hh, 0.125f);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

int umi_attn_bwd_mfma(const void* q, const void* k, const void* v, int ld, const void* o, const void* dO, int ldo,
                      const float* lse, void* dq, void* dk, void* dv, int ldd, float* delta, int B, int N, int Hh,
                      hipStream_t s) {
    dim3 grid((N + 127) / 128, B * Hh), block(256);
    hipLaunchKernelGGL(attn_q_side_kernel<true>, grid, block, 0, s, (const half_t*)q, (const half_t*)k, (const half_t*)v, ld,
                       (half_t*)o, (const half_t*)dO, ldo, (float*)lse, (half_t*)dq, ldd, delta, B, N, Hh, 0.125f);
    UMI_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_kv_side_kernel, grid, block, 0, s, (const half_t*)q, (const half_t*)k, (const half_t*)v, ld,
                       (const half_t*)dO, ldo, lse, (const float*)delta, (half_t*)dk, (half_t*)dv, ldd, B, N, Hh, 0.125f);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
