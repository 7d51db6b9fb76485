// conv3x3 (stride 1, pad 1) weight gradient on the gfx950 matrix cores, fp16 storage, fp32 accumulate.
//
//   dW[tap][ci][co] = sum over pixels  A[pixel + tap][ci] * dY[pixel][co],   A = tx(x) (BN+ReLU on load)
//
// is a GEMM whose contraction index is the PIXEL, while both tensors are NHWC (channel contiguous).  The
// MFMA operands therefore need 8 consecutive pixels of one channel per lane -- a transpose of what is in
// memory.  gfx950's ds_read_b64_tr_b16 does that transpose inside the LDS read: the tiles are staged
// exactly as they sit in HBM ([pixel][32 channels], 64-byte rows -> the 4 pixel rows of one 32-lane half
// hit 4 disjoint quarter bank-rows, conflict free) and each fragment is two transposing reads.
//
// Workgroup = 256 threads = 4 waves (2 ci-halves x 2 co-halves), 2 workgroups per CU.
// Block tile = 64 ci x 64 co x all 9 taps; every wave holds 9 accumulator tiles (32 ci x 32 co per tap,
// 144 registers).  K loop = pixel tiles of 4 rows x 32 pixels: the (4+2) x 34 input halo (64 ci) and the
// 4 x 32 dY tile (64 co) are staged through registers (issue-early / write-late, transform applied on
// the way), then per 16-pixel k-step and tap column the wave reuses 6 input-row fragments for the
// 3 taps x 4 rows.  Split-K over pixel tiles; partial slabs [split][tap][ci][co] fp32 are reduced in
// fixed order (deterministic) by wgrad_reduce_kernel into the parameter's own layout.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

void umi_launch_wgrad_reduce(const float* part, int splits, int RS, int Ci, int Co, float* dW, long s_co, long s_ci,
                             long s_t, float scale, hipStream_t st);

#ifdef UMI_STAMP
// diagnostic build only (tools/exp_stamp_wgrad.py): per-wave cycle sums of the tile-loop segments
__device__ unsigned long long umi_stamp_buf_w[2048 * 8];
#define UMI_TW(var)                                                                       \
    unsigned long long var;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0)
extern "C" int umi_debug_read_stamps_w(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(umi_stamp_buf_w), sizeof(umi_stamp_buf_w));
}
#endif

namespace {

constexpr int TR = 4;                    // image rows per pixel tile
constexpr int HPIX = (TR + 2) * 34;      // halo pixels
constexpr int PROW = 64;                 // LDS bytes per pixel row of one 32-channel chunk
constexpr int A_CHUNK = HPIX * PROW + 64; // 13056 + 64: the two chunks' rows fall on opposite halves of the 128-B store bank window
constexpr int A_BYTES = 2 * A_CHUNK;     // 64 input channels
constexpr int B_CHUNK = TR * 32 * PROW + 64;
constexpr int B_BYTES = 2 * B_CHUNK;
constexpr int SMEM = A_BYTES + B_BYTES;  // 42496
constexpr int KPA = (HPIX * 8 + 255) / 256;   // 7 16-B pieces per thread for the halo
constexpr int KPB = TR;                       // 4 pieces per thread for dY (piece k = image row k)

__device__ __forceinline__ half8 tr_frag(const unsigned char* p) {
    // two transposing reads: pixels +0..3 and +4..7 of this lane's channel
    typedef __attribute__((address_space(3))) short4v* lds_ptr;
    short4v r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    short4v r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * PROW));
    half4 h0 = __builtin_bit_cast(half4, r0), h1 = __builtin_bit_cast(half4, r1);
    return __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <bool HAS_TX>
__global__ __launch_bounds__(256, 2) void wgrad3x3_mfma_kernel(
    const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx, const half_t* __restrict__ dy, int lddy,
    float* __restrict__ part, int N, int H, int W, int Ci, int Co, int tiles_x, int tiles_y, int tiles_total,
    int tiles_per_split, int n_co_t, int fast_ci) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    __shared__ float4 txs[64];               // this block's 64 transform rows, read from LDS every tile (a global load per
                                             // tile costs ~1.6k cycles of the ~7k-cycle iteration, measured with stamps)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wci = wave >> 1, wco = wave & 1;
    // which channel-tile index runs fastest over consecutive workgroup ids (= over the 8 XCDs): with ci fastest every XCD
    // streams its own input-channel tiles once and all of dy, with co fastest the reverse; the host picks the cheaper one
    const int n_ci_t = (int)gridDim.x / n_co_t;
    const int ci0 = (fast_ci ? (int)blockIdx.x % n_ci_t : (int)blockIdx.x / n_co_t) * 64;
    const int co0 = (fast_ci ? (int)blockIdx.x / n_ci_t : (int)blockIdx.x % n_co_t) * 64;
#ifdef UMI_STAMP
    UMI_TW(t_start);
    unsigned long long seg[5] = {0, 0, 0, 0, 0};
#endif
    if (HAS_TX) {
        // stored transposed ([j][sub]) so that the 8 channel groups read 8 adjacent float4 (conflict-free)
        if (tid < 64) txs[(tid & 7) * 8 + (tid >> 3)] = ci0 + tid < Ci ? tx[ci0 + tid] : make_float4(0.f, 1.f, 0.f, 0.f);
        __syncthreads();
    }
    const int t_begin = blockIdx.y * tiles_per_split;
    int t_end = t_begin + tiles_per_split;
    if (t_end > tiles_total) t_end = tiles_total;

    // staging plan: A piece k -> halo pixel (tid>>3) + 32k, 8-channel group sub = tid & 7.  Loads go through buffer
    // descriptors over the whole tensors: voffset = tile origin (wave-uniform, carried incrementally: no divisions in the
    // loop) + a per-thread constant, or an out-of-range offset (returns zeros) for padding / partial tiles / masked channels.
    const int sub = tid & 7;
    // channel counts that are not multiples of 64 (Ci, Co % 8 == 0, e.g. the 16-channel decoder tail): the missing
    // 8-channel groups are staged as zeros and their outputs are not written
    const bool a_on = ci0 + sub * 8 < Ci, b_on = co0 + sub * 8 < Co;
    constexpr unsigned OOB = 0x7FFFFFFFu;
    int hyx[KPA];                              // (hy-1) << 16 | (hx-1) & 0xffff of the halo pixel, or INT_MIN: no piece
    int aoff[KPA];                             // byte offset of the piece relative to the tile's origin pixel
#pragma unroll
    for (int k = 0; k < KPA; ++k) {
        int hp = (tid >> 3) + 32 * k;
        int hy = hp / 34, hx = hp - hy * 34;
        hyx[k] = (hp < HPIX && a_on) ? (int)(((unsigned)(hy - 1) << 16) | ((unsigned)(hx - 1) & 0xffffu)) : (int)0x80000000;
        aoff[k] = (((hy - 1) * W + (hx - 1)) * ldx + ci0 + sub * 8) * 2;
    }
    const int bcol = tid >> 3;
    const int boff0 = (bcol * lddy + co0 + sub * 8) * 2;      // + k * W * lddy * 2 for image row k of the tile
    const int a_lds = (sub >> 2) * A_CHUNK + (tid >> 3) * PROW + (sub & 3) * 16;             // + k*32*PROW
    const int b_lds = A_BYTES + (sub >> 2) * B_CHUNK + (tid >> 3) * PROW + (sub & 3) * 16;   // + k*32*PROW
    const long ximg = (long)H * W * ldx, yimg = (long)H * W * lddy;      // elements per image (bytes < 2^31: host check)

    floatx16 acc[9];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    half8 araw[KPA], braw[KPB];
    bool avalid[KPA];

    // tile cursor of the NEXT tile to issue (image, origin row / column)
    int nx_n, nx_y, nx_x;
    {
        const int tpi = tiles_x * tiles_y;
        nx_n = t_begin / tpi;
        const int rm = t_begin - nx_n * tpi;
        nx_y = (rm / tiles_x) * TR;
        nx_x = (rm % tiles_x) * 32;
    }
#define UMI_ISSUE()                                                                                              \
    do {                                                                                                        \
        /* one descriptor per image (wave-uniform, SGPRs): offsets stay 32-bit whatever the batch size */        \
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + nx_n * ximg), 0, (int)(ximg * 2), 0x00020000); \
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + nx_n * yimg), 0, (int)(yimg * 2), 0x00020000); \
        const int org = nx_y * W + nx_x;                                                                        \
        const unsigned abase = (unsigned)org * (unsigned)(ldx * 2), bbase = (unsigned)org * (unsigned)(lddy * 2);  \
        _Pragma("unroll") for (int k = 0; k < KPA; ++k) {                                                       \
            const int gy = nx_y + (hyx[k] >> 16), gx = nx_x + (int)(short)(hyx[k] & 0xffff);                    \
            avalid[k] = hyx[k] != (int)0x80000000 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;  \
            araw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(                          \
                xrs, avalid[k] ? abase + (unsigned)aoff[k] : OOB, 0, 0));                                       \
        }                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < KPB; ++k) {                                                       \
            const bool ok = b_on && nx_y + k < H && nx_x + bcol < W;                                            \
            braw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(                          \
                yrs, ok ? bbase + (unsigned)(boff0 + k * W * lddy * 2) : OOB, 0, 0));                           \
        }                                                                                                       \
        nx_x += 32;                                                                                             \
        if (nx_x >= W) { nx_x = 0; nx_y += TR; if (nx_y >= H) { nx_y = 0; ++nx_n; } }                           \
    } while (0)

    // per-lane fragment addresses for the transposing reads (see file header)
    const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
    const int frag_lane = (8 * (g >> 1) + lq) * PROW + (16 * (g & 1) + 4 * lp) * 2;
    const unsigned char* a_frag = smem + wci * A_CHUNK + frag_lane;             // + ((rr*34) + 16*xh + dx) * PROW
    const unsigned char* b_frag = smem + A_BYTES + wco * B_CHUNK + frag_lane;   // + (r*32 + 16*xh) * PROW

    if (t_begin < t_end) UMI_ISSUE();
#ifdef UMI_STAMP
    UMI_TW(t_loop);
#endif
    for (int tile = t_begin; tile < t_end; ++tile) {
#ifdef UMI_STAMP
        UMI_TW(t0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        UMI_TW(t0b);
#endif
        if (HAS_TX) {
            // re-read the 8 transform rows per tile (L1-resident) instead of pinning 32 registers across the
            // MFMA phase; the opaque zero keeps the loads from being hoisted out of the tile loop
            int opaque = 0;
            asm volatile("" : "+v"(opaque));
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = txs[j * 8 + sub + opaque];
#pragma unroll
            for (int k = 0; k < KPA; ++k)
                if (avalid[k]) {
                    araw[k] = umi_tx8(araw[k], t);
                }
        }
#pragma unroll
        for (int k = 0; k < KPA; ++k)
            if ((tid >> 3) + 32 * k < HPIX) *reinterpret_cast<half8*>(smem + a_lds + k * 32 * PROW) = araw[k];
#pragma unroll
        for (int k = 0; k < KPB; ++k) *reinterpret_cast<half8*>(smem + b_lds + k * 32 * PROW) = braw[k];
#ifdef UMI_STAMP
        UMI_TW(t1);
#endif
        __syncthreads();
#ifdef UMI_STAMP
        UMI_TW(t2);
#endif
        if (tile + 1 < t_end) UMI_ISSUE();

        // (UMI_EXP_W_*: timing-only ablation builds of tools/exp_stamp_wgrad.py, never compiled into the shipped library)
#pragma unroll
        for (int xh = 0; xh < 2; ++xh) {
            half8 bfr[TR];
#pragma unroll
            for (int r = 0; r < TR; ++r) {
#ifndef UMI_EXP_W_NO_FRAG
                bfr[r] = tr_frag(b_frag + (r * 32 + 16 * xh) * PROW);
#else
                bfr[r] = braw[r];
#endif
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                half8 afr[TR + 2];
#pragma unroll
                for (int rr = 0; rr < TR + 2; ++rr) {
#ifndef UMI_EXP_W_NO_FRAG
                    afr[rr] = tr_frag(a_frag + (rr * 34 + 16 * xh + dx) * PROW);
#else
                    afr[rr] = araw[rr];
#endif
                }
#pragma unroll
                for (int r = 0; r < TR; ++r)
#pragma unroll
                    for (int dyi = 0; dyi < 3; ++dyi) {
#ifndef UMI_EXP_W_NO_MFMA
                        acc[dyi * 3 + dx] =
                            __builtin_amdgcn_mfma_f32_32x32x16_f16(afr[r + dyi], bfr[r], acc[dyi * 3 + dx], 0, 0, 0);
#else
                        asm volatile("" ::"v"(afr[r + dyi]), "v"(bfr[r]));
#endif
                    }
            }
        }
#ifdef UMI_STAMP
        UMI_TW(t3);
#endif
        __syncthreads();
#ifdef UMI_STAMP
        UMI_TW(t4);
        seg[0] += t0b - t0; seg[1] += t1 - t0b; seg[2] += t2 - t1; seg[3] += t3 - t2; seg[4] += t4 - t3;
#endif
    }
#undef UMI_ISSUE
#ifdef UMI_STAMP
    UMI_TW(t_ep0);
#endif

    // partial slab: part[((z*9 + tap)*Ci + ci)*Co + co]
    const int co = co0 + wco * 32 + (lane & 31);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int ci = ci0 + wci * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (ci < Ci && co < Co) part[(((long)blockIdx.y * 9 + tap) * Ci + ci) * Co + co] = acc[tap][r];
        }
#ifdef UMI_STAMP
    UMI_TW(t_end_);
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (lane == 0 && bid < 512) {
#pragma unroll
        for (int i = 0; i < 5; ++i) umi_stamp_buf_w[(bid * 4 + wave) * 8 + i] = seg[i];
        umi_stamp_buf_w[(bid * 4 + wave) * 8 + 5] = t_end - t_begin;
        umi_stamp_buf_w[(bid * 4 + wave) * 8 + 6] = t_loop - t_start;
        umi_stamp_buf_w[(bid * 4 + wave) * 8 + 7] = t_end_ - t_ep0;
    }
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Wave-specialised variant of the kernel above: one workgroup of 8 waves per CU, LDS tile double-buffered.
//   waves 0-3 (one per SIMD) = consumers: transposing fragment reads + the 72 MFMAs of a pixel tile, nothing else;
//   waves 4-7 (one per SIMD) = producers: buffer loads -> consumer-side transform -> ds_write of the NEXT tile.
// The hardware interleaves the two waves of a SIMD, so the staging VALU/LDS work runs under the MFMAs instead of
// in series with them (in-kernel stamps of the single-role kernel: staging ~45 % of every tile iteration, and the
// kernel is not power-limited: 1.2 kW at 2.4 GHz).  One barrier per tile.  Same math, same partial-slab layout.
constexpr int WS_SMEM = 2 * SMEM;        // 84,992 B (dynamic LDS)

// BNA: the dY operand is not read as stored but formed on the fly from the gradient of the ACTIVATED output (`dy` = dA) and the
// layer's raw output y -- stage 3 of the BatchNorm + ReLU backward (elementwise_f16.hip bn_bwd_apply_v8, same expression, same
// rounding to fp16) -- by the producer waves while they stage the tile; the workgroups of the first input-channel tile also
// write it out (`dz`) for the data-gradient kernel.  Replaces a standalone pass that read y and dA and re-wrote dA in place.
struct BnApply {
    const half_t* y; int ldy; const float4* tx; const float* rstd; const float* sum_dz; const float* sum_dzx; long M;
    half_t* dz; int lddz;
};

template <bool HAS_TX, bool BNA>
__global__ __launch_bounds__(512, 1) void wgrad3x3_ws_kernel(
    const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx, const half_t* __restrict__ dy, int lddy,
    float* __restrict__ part, int N, int H, int W, int Ci, int Co, int tiles_x, int tiles_y, int tiles_total,
    int tiles_per_split, int n_co_t, int fast_ci, BnApply ba) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_ws[];      // [2][SMEM] then txs[64]
    float4* txs = reinterpret_cast<float4*>(smem_ws + WS_SMEM);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    // which channel-tile index runs fastest over consecutive workgroup ids (= over the 8 XCDs): with ci fastest every XCD
    // streams its own input-channel tiles once and all of dy, with co fastest the reverse; the host picks the cheaper one
    const int n_ci_t = (int)gridDim.x / n_co_t;
    const int ci0 = (fast_ci ? (int)blockIdx.x % n_ci_t : (int)blockIdx.x / n_co_t) * 64;
    const int co0 = (fast_ci ? (int)blockIdx.x / n_ci_t : (int)blockIdx.x % n_co_t) * 64;
    const int t_begin = blockIdx.y * tiles_per_split;
    int t_end = t_begin + tiles_per_split;
    if (t_end > tiles_total) t_end = tiles_total;
    const int ntile = t_end - t_begin;
    if (HAS_TX) {
        // stored transposed ([j][sub]) so that the 8 channel groups read 8 adjacent float4 (conflict-free)
        if (tid < 64) txs[(tid & 7) * 8 + (tid >> 3)] = ci0 + tid < Ci ? tx[ci0 + tid] : make_float4(0.f, 1.f, 0.f, 0.f);
        __syncthreads();
    }

    if (producer) {
        // ---------------- producer waves: stage tile i+1 while the consumers work on tile i ----------------------------
        const int ptid = tid - 256;
        const int sub = ptid & 7;
        const bool a_on = ci0 + sub * 8 < Ci, b_on = co0 + sub * 8 < Co;
        constexpr unsigned OOB = 0x7FFFFFFFu;
        int hyx[KPA], aoff[KPA];
#pragma unroll
        for (int k = 0; k < KPA; ++k) {
            int hp = (ptid >> 3) + 32 * k;
            int hy = hp / 34, hx = hp - hy * 34;
            hyx[k] = (hp < HPIX && a_on) ? (int)(((unsigned)(hy - 1) << 16) | ((unsigned)(hx - 1) & 0xffffu)) : (int)0x80000000;
            aoff[k] = (((hy - 1) * W + (hx - 1)) * ldx + ci0 + sub * 8) * 2;
        }
        // Images made of whole tiles (W % 32 == 0, H % TR == 0: every shipped config): only the outermost halo ring of an edge tile
        // can leave the image, so a piece's validity is (its ring bits) & (the tile's edge bits) -- one v_and per piece and
        // tile instead of two coordinate additions and four comparisons, and the tile origin rides in the scalar offset of the
        // loads.  The producers share their SIMD's issue port with the MFMA wave: ~55 fewer vector instructions per tile, -0.9 %
        // on the bench's 17 layers (tools/ab_wgrad.py, same box).
        // (The per-lane offset of a buffer load is range-checked on its own and must not be negative: the halo ring's row -1 /
        //  column -1 are, relative to the tile origin, so the resource starts `aback` bytes in front of the image and every
        //  per-lane offset carries +aback.)
        const bool whole_tiles = (W & 31) == 0 && H % TR == 0;
        const int aback = (W + 1) * ldx * 2;
        int ering[KPA];            // bit 0: left halo column, 1: right, 2: top halo row, 3: bottom, 4: no such piece
        unsigned aoffb[KPA];
#pragma unroll
        for (int k = 0; k < KPA; ++k) {
            int hp = (ptid >> 3) + 32 * k;
            int hy = hp / 34, hx = hp - hy * 34;
            ering[k] = (hp < HPIX && a_on) ? ((hx == 0 ? 1 : 0) | (hx == 33 ? 2 : 0) | (hy == 0 ? 4 : 0) | (hy == TR + 1 ? 8 : 0)) : 16;
            aoffb[k] = (unsigned)(aoff[k] + aback);
        }
        const int bcol = ptid >> 3;
        const int boff0 = (bcol * lddy + co0 + sub * 8) * 2;
        const int a_lds = (sub >> 2) * A_CHUNK + (ptid >> 3) * PROW + (sub & 3) * 16;
        const int b_lds = A_BYTES + (sub >> 2) * B_CHUNK + (ptid >> 3) * PROW + (sub & 3) * 16;
        const long ximg = (long)H * W * ldx, yimg = (long)H * W * lddy;
        // this thread's 8 transform rows stay in registers (a producer wave has no accumulators to make room for; read from
        // LDS per tile they cost 12 ds_read2_b32 with 8-way bank conflicts: measured ~1.5k LDS cycles per tile)
        float4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = HAS_TX ? txs[j * 8 + sub] : make_float4(0.f, 1.f, 0.f, 0.f);
        // BatchNorm-backward constants of this thread's 8 output channels (BNA)
        float4 tb[8];
        float rsb[8], c1[8], c2[8];
        const long ybimg = BNA ? (long)H * W * ba.ldy : 0;
        const int ybo0 = BNA ? (bcol * ba.ldy + co0 + sub * 8) * 2 : 0;
        const bool dz_writer = BNA && ci0 == 0;
        if (BNA) {
            const float invM = 1.f / (float)ba.M;          // on the device, as the standalone kernels compute it
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = co0 + sub * 8 + j;
                const bool in = c < Co;
                tb[j] = in ? ba.tx[c] : make_float4(0.f, 1.f, 0.f, 0.f);
                rsb[j] = in ? ba.rstd[c] : 0.f;
                c1[j] = in ? ba.sum_dz[c] * invM : 0.f;
                c2[j] = in ? ba.sum_dzx[c] * invM : 0.f;
            }
        }
        half8 yraw0[KPB], yraw1[KPB];
        bool bvalid0[KPB], bvalid1[KPB];
        int st_n0 = 0, st_y0 = 0, st_x0 = 0, st_n1 = 0, st_y1 = 0, st_x1 = 0;      // tile origin of each register set (dz stores)
        // two register sets: the loads of tile i+2 stay in flight for a whole iteration while tile i+1 is transformed and
        // written (with a single set the producer's iteration is the exposed global-load latency plus the stores)
        half8 araw0[KPA], braw0[KPB], araw1[KPA], braw1[KPB];
        bool avalid0[KPA], avalid1[KPA];
        int nx_n, nx_y, nx_x;
        {
            const int tpi = tiles_x * tiles_y;
            nx_n = t_begin / tpi;
            const int rm = t_begin - nx_n * tpi;
            nx_y = (rm / tiles_x) * TR;
            nx_x = (rm % tiles_x) * 32;
        }
#ifdef UMI_EXP_WS_NO_LOAD      /* timing-only: every load out of range (returns zeros without touching memory) */
#define UMI_EXP_LOAD_OK(c_) ((c_) && H < 0)
#else
#define UMI_EXP_LOAD_OK(c_) (c_)
#endif
#define UMI_ISSUE_WS(S)                                                                                          \
    do {                                                                                                        \
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + nx_n * ximg), 0, (int)(ximg * 2), 0x00020000); \
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + nx_n * yimg), 0, (int)(yimg * 2), 0x00020000); \
        const int org = nx_y * W + nx_x;                                                                        \
        const unsigned abase = (unsigned)org * (unsigned)(ldx * 2), bbase = (unsigned)org * (unsigned)(lddy * 2);  \
        if (whole_tiles) {                                                                                      \
            const __amdgpu_buffer_rsrc_t xrsb = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(x + nx_n * ximg) - aback), 0, (int)(ximg * 2) + aback, 0x00020000); \
            const int edge = (nx_x == 0 ? 1 : 0) | (nx_x + 32 >= W ? 2 : 0) | (nx_y == 0 ? 4 : 0) | (nx_y + TR >= H ? 8 : 0) | 16; \
            _Pragma("unroll") for (int k = 0; k < KPA; ++k) {                                                   \
                avalid##S[k] = (ering[k] & edge) == 0;                                                          \
                araw##S[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(                   \
                    xrsb, UMI_EXP_LOAD_OK(avalid##S[k]) ? aoffb[k] : OOB, abase, 0));                           \
            }                                                                                                   \
        } else {                                                                                                \
            _Pragma("unroll") for (int k = 0; k < KPA; ++k) {                                                   \
                const int gy = nx_y + (hyx[k] >> 16), gx = nx_x + (int)(short)(hyx[k] & 0xffff);                \
                avalid##S[k] = hyx[k] != (int)0x80000000 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W; \
                araw##S[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(                   \
                    xrs, UMI_EXP_LOAD_OK(avalid##S[k]) ? abase + (unsigned)aoff[k] : OOB, 0, 0));               \
            }                                                                                                   \
        }                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < KPB; ++k) {                                                       \
            const bool ok = b_on && (whole_tiles || (nx_y + k < H && nx_x + bcol < W));                         \
            braw##S[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(                       \
                yrs, UMI_EXP_LOAD_OK(ok) ? (unsigned)boff0 : OOB, bbase + k * W * lddy * 2, 0));                \
            if (BNA) {                                                                                          \
                const __amdgpu_buffer_rsrc_t ybrs = __builtin_amdgcn_make_buffer_rsrc((void*)(ba.y + nx_n * ybimg), 0, (int)(ybimg * 2), 0x00020000); \
                bvalid##S[k] = ok;                                                                              \
                yraw##S[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(                   \
                    ybrs, ok ? (unsigned)org * (unsigned)(ba.ldy * 2) + (unsigned)(ybo0 + k * W * ba.ldy * 2) : OOB, 0, 0)); \
            }                                                                                                   \
        }                                                                                                       \
        if (BNA) { st_n##S = nx_n; st_y##S = nx_y; st_x##S = nx_x; }                                            \
        nx_x += 32;                                                                                             \
        if (nx_x >= W) { nx_x = 0; nx_y += TR; if (nx_y >= H) { nx_y = 0; ++nx_n; } }                           \
    } while (0)
#ifdef UMI_EXP_WS_NO_STORE
#define UMI_STORE_WS(S, buf_)                                                                                    \
    do {                                                                                                        \
        _Pragma("unroll") for (int k = 0; k < KPA; ++k) asm volatile("" ::"v"(araw##S[k]));                     \
        _Pragma("unroll") for (int k = 0; k < KPB; ++k) asm volatile("" ::"v"(braw##S[k]));                     \
    } while (0)
#else
#define UMI_STORE_WS(S, buf_)                                                                                    \
    do {                                                                                                        \
        unsigned char* sb = smem_ws + (buf_) * SMEM;                                                            \
        if (HAS_TX) {                                                                                           \
            _Pragma("unroll") for (int k = 0; k < KPA; ++k) {                                                   \
                const half8 t8 = umi_tx8(araw##S[k], t);                                                        \
                araw##S[k] = avalid##S[k] ? t8 : araw##S[k];      /* out-of-image pieces were loaded as zeros */ \
            }                                                                                                   \
        }                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < KPA; ++k)                                                         \
            if ((ptid >> 3) + 32 * k < HPIX) *reinterpret_cast<half8*>(sb + a_lds + k * 32 * PROW) = araw##S[k]; \
        if (BNA) {                                                                                              \
            _Pragma("unroll") for (int k = 0; k < KPB; ++k) {                                                   \
                const half8 o = umi_bn_dz8(yraw##S[k], braw##S[k], tb, rsb, c1, c2);   /* as bn_bwd_apply_v8 (common.h) */ \
                braw##S[k] = bvalid##S[k] ? o : braw##S[k];           /* pixels outside the image stay zero */   \
                if (dz_writer && bvalid##S[k])                                                                  \
                    *reinterpret_cast<half8*>(ba.dz + ((long)((long)st_n##S * H + st_y##S + k) * W + st_x##S + bcol) * ba.lddz + \
                                              co0 + sub * 8) = braw##S[k];                                      \
            }                                                                                                   \
        }                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < KPB; ++k) *reinterpret_cast<half8*>(sb + b_lds + k * 32 * PROW) = braw##S[k]; \
    } while (0)
#endif

        // tile j lives in register set j & 1 and goes to LDS buffer j & 1; issue order = tile order
        if (ntile > 0) UMI_ISSUE_WS(0);
        if (ntile > 1) UMI_ISSUE_WS(1);
        if (ntile > 0) UMI_STORE_WS(0, 0);
        if (ntile > 2) UMI_ISSUE_WS(0);
        __syncthreads();                                    // tile 0 is in buffer 0
#ifdef UMI_STAMP
        unsigned long long pw = 0, pb = 0;
#define UMI_PS(a_) UMI_TW(a_)
#else
#define UMI_PS(a_)
#endif
        for (int i = 0; i < ntile; i += 2) {
            UMI_PS(q0);
#ifndef UMI_EXP_WS_NO_PROD
            if (i + 1 < ntile) {                            // consumers are on tile i (buffer 0)
                UMI_STORE_WS(1, 1);
                if (i + 3 < ntile) UMI_ISSUE_WS(1);
            }
#endif
            UMI_PS(q1);
            __syncthreads();
            UMI_PS(q2);
            if (i + 1 < ntile) {                            // consumers are on tile i + 1 (buffer 1)
#ifndef UMI_EXP_WS_NO_PROD
                if (i + 2 < ntile) {
                    UMI_STORE_WS(0, 0);
                    if (i + 4 < ntile) UMI_ISSUE_WS(0);
                }
#endif
                UMI_PS(q3);
                __syncthreads();
                UMI_PS(q4);
#ifdef UMI_STAMP
                pw += q3 - q2; pb += q4 - q3;
#endif
            }
#ifdef UMI_STAMP
            pw += q1 - q0; pb += q2 - q1;
#endif
        }
#ifdef UMI_STAMP
        {
            const int bid = blockIdx.y * gridDim.x + blockIdx.x;
            if (lane == 0 && bid < 256) {
                umi_stamp_buf_w[(bid * 8 + wave) * 8 + 0] = pw;
                umi_stamp_buf_w[(bid * 8 + wave) * 8 + 1] = pb;
                umi_stamp_buf_w[(bid * 8 + wave) * 8 + 5] = ntile;
            }
        }
#endif
#undef UMI_PS
#undef UMI_ISSUE_WS
#undef UMI_STORE_WS
        return;
    }

    // ---------------- consumer waves -------------------------------------------------------------------------------
    __builtin_amdgcn_s_setprio(3);                          // the matrix-pipe wave outranks its SIMD's producer wave
    const int wci = wave >> 1, wco = wave & 1;
    // v_mfma_f32_16x16x32_f16: K = 32 pixels (one tile row) per instruction, wave tile = 2 x 2 tiles of 16 ci x 16 co per tap.
    // Same 144 accumulator registers and the same LDS reads per MAC as the 32x32x16 shape, but half the accumulator
    // traffic per MAC: under the power cap the chip holds a higher clock (measured +10 % on this kernel, +2 % on the step).
    // A consumer wave is alone on its SIMD's matrix core: nothing hides its LDS latency but itself, so the A fragments of
    // step s+1 (a step = one tap column x one 16-channel tile: 24 MFMAs) are read while the MFMAs of step s run.
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    floatx4 acc[9][2][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ca = 0; ca < 2; ++ca)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[t][ca][cb][r] = 0.f;
    const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
    const int frag_lane = (8 * g + lq) * PROW + (4 * lp) * 2;      // pixel 8g + lq (+4), channel 4*lp of a 16-channel tile
    __syncthreads();                                        // tile 0 staged
#define UMI_LD_A16(dst, ca_, dx_)                                                                                \
    _Pragma("unroll") for (int rr = 0; rr < TR + 2; ++rr) dst[rr] = tr_frag(a_frag + (rr * 34 + (dx_)) * PROW + (ca_) * 32)
#define UMI_MMA16(af_, ca_, dx_)                                                                                 \
    _Pragma("unroll") for (int r = 0; r < TR; ++r)                                                              \
        _Pragma("unroll") for (int dyi = 0; dyi < 3; ++dyi)                                                     \
            _Pragma("unroll") for (int cb = 0; cb < 2; ++cb)                                                    \
                acc[dyi * 3 + (dx_)][ca_][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af_[r + dyi], bfr[r][cb], acc[dyi * 3 + (dx_)][ca_][cb], 0, 0, 0)
#define UMI_PIN() __builtin_amdgcn_sched_barrier(0)
#ifdef UMI_STAMP
    unsigned long long cw = 0, cb_ = 0, cstart = 0;
#endif
    for (int i = 0; i < ntile; ++i) {
#ifdef UMI_STAMP
        UMI_TW(c0);
#endif
        const unsigned char* a_frag = smem_ws + (i & 1) * SMEM + wci * A_CHUNK + frag_lane;
        const unsigned char* b_frag = smem_ws + (i & 1) * SMEM + A_BYTES + wco * B_CHUNK + frag_lane;
        half8 a0[TR + 2], a1[TR + 2], bfr[TR][2];
#pragma unroll
        for (int r = 0; r < TR; ++r)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) bfr[r][cb] = tr_frag(b_frag + (r * 32) * PROW + cb * 32);
#ifndef UMI_WS_COARSE
        // the 12 transposing reads of step s + 1 are spread between the 24 MFMAs of step s (one read behind every MFMA pair) instead of
        // issued as a burst in front of them: -1.5 % on the 17 layers of the bench (round 3, tools/ab_wgrad.py; UMI_WS_COARSE = the old order)
#define UMI_STEP(ld_, mma_) do { ld_; mma_;                                                              \
            _Pragma("unroll") for (int i_ = 0; i_ < 12; ++i_) {                                                     \
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }  \
            UMI_PIN(); } while (0)
        UMI_LD_A16(a0, 0, 0); UMI_PIN();
        UMI_STEP(UMI_LD_A16(a1, 1, 0), UMI_MMA16(a0, 0, 0));
        UMI_STEP(UMI_LD_A16(a0, 0, 1), UMI_MMA16(a1, 1, 0));
        UMI_STEP(UMI_LD_A16(a1, 1, 1), UMI_MMA16(a0, 0, 1));
        UMI_STEP(UMI_LD_A16(a0, 0, 2), UMI_MMA16(a1, 1, 1));
        UMI_STEP(UMI_LD_A16(a1, 1, 2), UMI_MMA16(a0, 0, 2));
        UMI_MMA16(a1, 1, 2);
#undef UMI_STEP
#else
        UMI_LD_A16(a0, 0, 0); UMI_PIN();
#ifdef UMI_STAMP_STEPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        UMI_TW(cs0);
        cstart += cs0 - c0;
#endif
        UMI_LD_A16(a1, 1, 0); UMI_PIN(); UMI_MMA16(a0, 0, 0); UMI_PIN();
        UMI_LD_A16(a0, 0, 1); UMI_PIN(); UMI_MMA16(a1, 1, 0); UMI_PIN();
        UMI_LD_A16(a1, 1, 1); UMI_PIN(); UMI_MMA16(a0, 0, 1); UMI_PIN();
        UMI_LD_A16(a0, 0, 2); UMI_PIN(); UMI_MMA16(a1, 1, 1); UMI_PIN();
        UMI_LD_A16(a1, 1, 2); UMI_PIN(); UMI_MMA16(a0, 0, 2); UMI_PIN();
        UMI_MMA16(a1, 1, 2);
#endif
#ifdef UMI_STAMP
        UMI_TW(c1);
#endif
        __syncthreads();
#ifdef UMI_STAMP
        UMI_TW(c2);
        cw += c1 - c0; cb_ += c2 - c1;
#endif
    }
#ifdef UMI_STAMP
    {
        const int bid = blockIdx.y * gridDim.x + blockIdx.x;
        if (lane == 0 && bid < 256) {
            umi_stamp_buf_w[(bid * 8 + wave) * 8 + 0] = cw;
            umi_stamp_buf_w[(bid * 8 + wave) * 8 + 1] = cb_;
            umi_stamp_buf_w[(bid * 8 + wave) * 8 + 2] = cstart;
            umi_stamp_buf_w[(bid * 8 + wave) * 8 + 5] = ntile;
        }
    }
#endif
#undef UMI_LD_A16
#undef UMI_MMA16
#undef UMI_PIN
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ca = 0; ca < 2; ++ca)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int co = co0 + wco * 32 + cb * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = ci0 + wci * 32 + ca * 16 + 4 * (lane >> 4) + r;
                    if (ci < Ci && co < Co) part[(((long)blockIdx.y * 9 + tap) * Ci + ci) * Co + co] = acc[tap][ca][cb][r];
                }
            }
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradient of the 2x2 stride-2 pair (ConvTranspose2d(k=2,s=2) and its adjoint), reference Model.py:56-57:
//   dW[tap][cx][cy] = sum over low-res pixels p   X[2p + tap][cx] * tx(Y[p])[cy]
// X = hi-res tensor (2h x 2w, e.g. the gradient of the upsampled map), Y = low-res tensor (the ConvT input, BN+ReLU
// applied on load).  Same transposing-read scheme; the stride-2 pixel walk of X is free because every lane supplies
// its own row address to ds_read_b64_tr_b16.  Tile = 2 low-res rows x 32 pixels; 4 accumulator tiles per wave.
constexpr int T2R = 2;                                   // low-res rows per tile
constexpr int X2_PIX = 2 * T2R * 64;                     // hi-res pixels per tile (4 rows x 64)
constexpr int X2_CHUNK = X2_PIX * PROW + 64;             // +64: chunk pairs on opposite halves of the store bank window
constexpr int Y2_CHUNK = T2R * 32 * PROW + 64;
constexpr int SMEM2 = 2 * X2_CHUNK + 2 * Y2_CHUNK;       // 40960
constexpr int KPX2 = X2_PIX * 8 / 256;                   // 8
constexpr int KPY2 = T2R * 32 * 8 / 256;                 // 2

__device__ __forceinline__ half8 tr_frag_s2(const unsigned char* p) {      // pixel stride 2 along K
    typedef __attribute__((address_space(3))) short4v* lds_ptr;
    short4v r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    short4v r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 8 * PROW));
    half4 h0 = __builtin_bit_cast(half4, r0), h1 = __builtin_bit_cast(half4, r1);
    return __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// YH = 64-channel halves of the low-res operand per workgroup (1: 64 x 64 channel tile, 2: 64 x 128).  With 128 of them every staged
// hi-res pixel (the operand that dominates the staging: 256 pixels x 64 channels per tile) feeds twice as many MFMAs -- 12 staged
// pieces per 32 MFMAs and thread instead of 10 per 16 (round 3; 8 accumulator tiles = 128 registers per wave).
// cpart != nullptr: the workgroups of the first cy tile also sum the hi-res operand's columns over their pixels -- the bias gradient of
// the transposed conv, d bias[cx] = sum over all hi-res pixels of X[.][cx] -- from the pieces they stage anyway: cpart[split][2][Cx]
// (slot 0), finished by reduce_rows2 (a pass of its own over the 537 MB gradient at the bench shape otherwise: umi_colsum).
template <bool HAS_TX, int YH>
__global__ __launch_bounds__(256, 2) void wgradT2x2_mfma_kernel(
    const half_t* __restrict__ x, int ldx, const half_t* __restrict__ y, int ldy, const float4* __restrict__ txy,
    float* __restrict__ part, int N, int h, int w, int Cx, int Cy, int tiles_x, int tiles_y, int tiles_total,
    int tiles_per_split, int n_cy_t, float* __restrict__ cpart) {
    constexpr int SMEM_T = 2 * X2_CHUNK + 2 * YH * Y2_CHUNK;
    constexpr int KPY = KPY2 * YH;                           // piece k: pixel half k & 1... see y_lds
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_T];
    __shared__ float4 txs[64 * YH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wcx = wave >> 1, wcy = wave & 1;
    const int cx0 = (blockIdx.x / n_cy_t) * 64, cy0 = (blockIdx.x % n_cy_t) * (64 * YH);
    if (HAS_TX) {
        if (tid < 64 * YH) txs[(tid >> 6) * 64 + (tid & 7) * 8 + ((tid & 63) >> 3)] = txy[cy0 + tid];      // transposed per half: see wgrad3x3
        __syncthreads();
    }
    const int t_begin = blockIdx.y * tiles_per_split;
    int t_end = t_begin + tiles_per_split;
    if (t_end > tiles_total) t_end = tiles_total;
    const int H2 = 2 * h, W2 = 2 * w;

    const int sub = tid & 7;
    const int x_lds = (sub >> 2) * X2_CHUNK + (tid >> 3) * PROW + (sub & 3) * 16;                  // + k*32*PROW
    const int y_lds = 2 * X2_CHUNK + (sub >> 2) * Y2_CHUNK + (tid >> 3) * PROW + (sub & 3) * 16;   // + (k % KPY2)*32*PROW + (k / KPY2) * 2 * Y2_CHUNK
    const half_t* xin = x + cx0 + sub * 8;
    const half_t* yin = y + cy0 + sub * 8;                   // + (k / KPY2) * 64 channels

    floatx16 acc[4][YH];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < YH; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    half8 xraw[KPX2], yraw[KPY];
    bool yvalid[KPY];
    const bool csum = cpart != nullptr && (int)blockIdx.x % n_cy_t == 0;      // (workgroup-uniform)
    float cs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[j] = 0.f;
    half8 zero8;
#pragma unroll
    for (int j = 0; j < 8; ++j) zero8[j] = (half_t)0.f;

#define UMI_ISSUE2(tile_)                                                                                        \
    do {                                                                                                        \
        const int tt = (tile_);                                                                                 \
        const int n_ = tt / (tiles_x * tiles_y);                                                                \
        const int rm_ = tt - n_ * tiles_x * tiles_y;                                                            \
        const int ty0_ = (rm_ / tiles_x) * T2R, tx0_ = (rm_ % tiles_x) * 32;                                    \
        _Pragma("unroll") for (int k = 0; k < KPX2; ++k) {                                                      \
            int gy = 2 * ty0_ + (k >> 1), gx = 2 * tx0_ + (tid >> 3) + 32 * (k & 1);                            \
            xraw[k] = (gy < H2 && gx < W2)                                                                      \
                          ? *reinterpret_cast<const half8*>(xin + ((long)((long)n_ * H2 + gy) * W2 + gx) * ldx) \
                          : zero8;                                                                              \
        }                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < KPY; ++k) {                                                       \
            int gy = ty0_ + (k % KPY2), gx = tx0_ + (tid >> 3);                                                 \
            yvalid[k] = gy < h && gx < w;                                                                       \
            yraw[k] = yvalid[k] ? *reinterpret_cast<const half8*>(yin + (k / KPY2) * 64 + ((long)((long)n_ * h + gy) * w + gx) * ldy) \
                                : zero8;                                                                        \
        }                                                                                                       \
    } while (0)

    const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
    const int ch_lane = (16 * (g & 1) + 4 * lp) * 2;
    const unsigned char* x_frag = smem + wcx * X2_CHUNK + 2 * (8 * (g >> 1) + lq) * PROW + ch_lane;
    // wave wcy multiplies the 32-channel chunk wcy of every 64-channel half: chunk index 2 * half + wcy
    const unsigned char* y_frag = smem + 2 * X2_CHUNK + wcy * Y2_CHUNK + (8 * (g >> 1) + lq) * PROW + ch_lane;

    if (t_begin < t_end) UMI_ISSUE2(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        if (HAS_TX) {
            int opaque = 0;
            asm volatile("" : "+v"(opaque));
#pragma unroll
            for (int hh = 0; hh < YH; ++hh) {
                float4 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = txs[hh * 64 + j * 8 + sub + opaque];
#pragma unroll
                for (int k = hh * KPY2; k < (hh + 1) * KPY2; ++k)
                    if (yvalid[k]) {
                        yraw[k] = umi_tx8(yraw[k], t);
                    }
            }
        }
#pragma unroll
        for (int k = 0; k < KPX2; ++k) *reinterpret_cast<half8*>(smem + x_lds + k * 32 * PROW) = xraw[k];
        if (csum) {                                      // (pieces outside the image were loaded as zeros)
#pragma unroll
            for (int k = 0; k < KPX2; ++k)
#pragma unroll
                for (int j = 0; j < 8; ++j) cs[j] += (float)xraw[k][j];
        }
#pragma unroll
        for (int k = 0; k < KPY; ++k)
            *reinterpret_cast<half8*>(smem + y_lds + (k % KPY2) * 32 * PROW + (k / KPY2) * 2 * Y2_CHUNK) = yraw[k];
        __syncthreads();
        if (tile + 1 < t_end) UMI_ISSUE2(tile + 1);
#pragma unroll
        for (int r = 0; r < T2R; ++r)
#pragma unroll
            for (int xh = 0; xh < 2; ++xh) {
                half8 bfr[YH];
#pragma unroll
                for (int hh = 0; hh < YH; ++hh) bfr[hh] = tr_frag(y_frag + hh * 2 * Y2_CHUNK + (r * 32 + 16 * xh) * PROW);
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    half8 afr = tr_frag_s2(x_frag + ((2 * r + (tap >> 1)) * 64 + 32 * xh + (tap & 1)) * PROW);
#pragma unroll
                    for (int hh = 0; hh < YH; ++hh) acc[tap][hh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr, bfr[hh], acc[tap][hh], 0, 0, 0);
                }
            }
        __syncthreads();
    }
#undef UMI_ISSUE2
    if (csum) {                                          // 32 pixel lanes per 8-channel group `sub` -> one value per channel
        float* red = reinterpret_cast<float*>(smem);     // [32][64] (the tile loop ended with a barrier)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(tid >> 3) * 64 + sub * 8 + j] = cs[j];
        __syncthreads();
        if (tid < 64) {
            float a = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) a += red[k * 64 + tid];
            cpart[((long)blockIdx.y * 2 + 0) * Cx + cx0 + tid] = a;
        }
    }
#pragma unroll
    for (int hh = 0; hh < YH; ++hh) {
        const int cy = cy0 + hh * 64 + wcy * 32 + (lane & 31);
#pragma unroll
        for (int tap = 0; tap < 4; ++tap)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int cx = cx0 + wcx * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                part[(((long)blockIdx.y * 4 + tap) * Cx + cx) * Cy + cy] = acc[tap][hh][r];
            }
    }
}

void planT(int N, int h, int w, int Cx, int Cy, int* tiles_x, int* tiles_y, int* tiles_total, int* splits, int* tps) {
    *tiles_x = (w + 31) / 32;
    *tiles_y = (h + T2R - 1) / T2R;
    *tiles_total = N * (*tiles_x) * (*tiles_y);
    const long pairs = (long)(Cx / 64) * (Cy % 128 == 0 ? Cy / 128 : Cy / 64);       // workgroup tiles: 64 x 128 channels where Cy allows
    static const long target2 = [] { const char* e = getenv("UMI_WGRAD_SPLIT_TARGET2"); long v = e ? atol(e) : 0; return v > 0 ? v : 1024L; }();
    long want = (target2 + pairs - 1) / pairs;
    const long slab = 4L * Cx * Cy * 4;
    long cap = (64L << 20) / slab;
    if (cap < 1) cap = 1;
    if (pairs * cap < 512 && pairs < 512) cap = (512 + pairs - 1) / pairs;
    if (want > cap) want = cap;
    if (want > *tiles_total) want = *tiles_total;
    if (want < 1) want = 1;
    *tps = (int)((*tiles_total + want - 1) / want);
    *splits = (*tiles_total + *tps - 1) / *tps;
}

void plan(int N, int H, int W, int Ci, int Co, int* tiles_x, int* tiles_y, int* tiles_total, int* splits, int* tps) {
    *tiles_x = (W + 31) / 32;
    *tiles_y = (H + TR - 1) / TR;
    *tiles_total = N * (*tiles_x) * (*tiles_y);
    const long pairs = (long)((Ci + 63) / 64) * ((Co + 63) / 64);
    // one 8-wave workgroup per CU resident: ONE balanced round of ~256 workgroups measured best (553 / 566 / 572 img/s for
    // targets 1024 / 512 / 256 on one box: every split is a slab written and re-read); UMI_WGRAD_SPLIT_TARGET overrides
    static const long target = [] { const char* e = getenv("UMI_WGRAD_SPLIT_TARGET"); long v = e ? atol(e) : 0; return v > 0 ? v : 256L; }();
    long want = (target + pairs - 1) / pairs;
    const long slab = 9L * Ci * Co * 4;
    long cap = (96L << 20) / slab;                      // split-K slabs are written + re-read: bound that traffic
    if (cap < 1) cap = 1;
    if (pairs * cap < 256 && pairs < 256) cap = (256 + pairs - 1) / pairs;   // but never starve the chip
    if (want > cap) want = cap;
    if (want > *tiles_total) want = *tiles_total;
    if (want < 1) want = 1;
    *tps = (int)((*tiles_total + want - 1) / want);
    *splits = (*tiles_total + *tps - 1) / *tps;
}

// ------------------------------------------------------------------------------------------------------------
// Weight gradient of a pointwise conv / nn.Linear:  dW[ci][co] = sum_p tx(x[p])[ci] * dy[p][co]
// (TransUNet: QKV/out/MLP linears vit_seg_modeling.py:58-62,100-101, patch embedding, bottleneck 1x1 convs).
// Pixels are linear rows (dense NHWC), K tile = 64 rows.  Block tile TM x TM channels, 2 x 2 waves, each wave
// (TM/2)^2 = MT x MT accumulator tiles; both operands by transposing LDS reads.
// GATHER: weight gradient of any R x S conv with stride / padding (TransUNet's stride-2 convs, resnet_skip.py:52-60):
// blockIdx.z = tap, dy rows stay linear, the x row of dy pixel (n, ho, wo) is (n, s*ho + ty - pad, s*wo + tx - pad),
// zero outside the image:  dW[tap][ci][co] = sum_p tx(x[gather(p, tap)])[ci] * dy[p][co].
// GROUP: up to 16 independent problems of ONE shape in a launch (blockIdx.z = problem; operands from the table in the kernel
// arguments), no split-K: every workgroup runs the whole pixel range and stores its tile straight into the parameter-layout
// gradient.  For the twelve encoder layers of a ViT, whose per-layer weight gradients (K = 4,704 tokens, 36..144 channel
// tiles) each fill the chip only with a 7..9-way split -- slabs written and re-read that cost more than the GEMM.
struct WGeo { int Ho, Wo, H, W, S, stride, pad; };
struct WGroup { const half_t* x[16]; const half_t* dy[16]; float* dW[16]; long s_co, s_ci; float scale; };
struct WNoGroup {};
template <int TM, bool HAS_TX, bool GATHER, bool GROUP = false>
__global__ __launch_bounds__(256, 2) void wgrad1x1_mfma_kernel(const half_t* __restrict__ x, int ldx,
                                                               const float4* __restrict__ tx,
                                                               const half_t* __restrict__ dy, int lddy,
                                                               float* __restrict__ part, long M, int Ci, int Co,
                                                               int tiles_total, int tiles_per_split, int n_co_t, WGeo geo,
                                                               typename std::conditional<GROUP, WGroup, WNoGroup>::type grp) {
    if constexpr (GROUP) { x = grp.x[blockIdx.z]; dy = grp.dy[blockIdx.z]; }
    constexpr int MT = TM / 64;                       // 32x32 tiles per wave per dimension
    constexpr int NCH = TM / 32;                      // 32-channel chunks per operand
    constexpr int CHB = 64 * PROW + 64;               // bytes per chunk (64 pixel rows; +64: see A_CHUNK)
    constexpr int PPP = TM / 8;                       // 16-B pieces per pixel
    constexpr int PXS = 256 / PPP;                    // pixels staged per pass
    constexpr int KP = 64 / PXS;                      // passes
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NCH * CHB];
    __shared__ float4 txs[TM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wci = wave >> 1, wco = wave & 1;
    const int ci0 = (blockIdx.x / n_co_t) * TM, co0 = (blockIdx.x % n_co_t) * TM;
    if (HAS_TX) {
        // transposed [j][sub]: conflict-free reads; rows past Ci (partial channel tile) are the identity
        if (tid < TM) txs[(tid & 7) * PPP + (tid >> 3)] = ci0 + tid < Ci ? tx[ci0 + tid] : make_float4(0.f, 1.f, 0.f, 0.f);
        __syncthreads();
    }
    const int t_begin = blockIdx.y * tiles_per_split;
    int t_end = t_begin + tiles_per_split;
    if (t_end > tiles_total) t_end = tiles_total;

    const int sub = tid % PPP, prow = tid / PPP;
    const int lds_off = (sub >> 2) * CHB + prow * PROW + (sub & 3) * 16;        // + k*PXS*PROW ; dy: + NCH*CHB
    constexpr unsigned OOB = 0x7FFFFFFFu;
    const long xrows = GATHER ? (M / ((long)geo.Ho * geo.Wo)) * geo.H * geo.W : M;       // pixel rows of the x tensor
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + ci0), 0, (int)(xrows * ldx * 2 - ci0 * 2 > 0x7FFFFFF0L ? 0x7FFFFFF0L : xrows * ldx * 2 - ci0 * 2), 0x00020000);
    const int tap_y = GATHER ? (int)blockIdx.z / geo.S : 0, tap_x = GATHER ? (int)blockIdx.z - tap_y * geo.S : 0;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + co0), 0, (int)(M * lddy * 2 - co0 * 2 > 0x7FFFFFF0L ? 0x7FFFFFF0L : M * lddy * 2 - co0 * 2), 0x00020000);

    // v_mfma_f32_16x16x32_f16 (K = 32 pixel rows per instruction): 2*MT x 2*MT tiles of 16 x 16 per wave -- the register count
    // of the 32x32x16 tiling, half its accumulator traffic per MAC (higher clock under the power cap, see wgrad3x3_ws_kernel)
    constexpr int T16 = 2 * MT;
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    floatx4 acc[T16][T16];
#pragma unroll
    for (int a = 0; a < T16; ++a)
#pragma unroll
        for (int b = 0; b < T16; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    half8 xraw[KP], yraw[KP];
    bool xval[KP];

#define UMI_ISSUE1(tile_)                                                                                         \
    do {                                                                                                         \
        const long m0_ = (long)(tile_) * 64;                                                                     \
        _Pragma("unroll") for (int k = 0; k < KP; ++k) {                                                         \
            long m = m0_ + prow + k * PXS;                                                                       \
            xval[k] = m < M;                                                                                     \
            unsigned oy = xval[k] ? (unsigned)(m * lddy * 2 + sub * 16) : OOB;                                   \
            long mx = m;                                                                                         \
            if (GATHER && xval[k]) {                                                                             \
                const int hw = geo.Ho * geo.Wo;                                                                  \
                const int n = (int)(m / hw), r = (int)(m - (long)n * hw);                                        \
                const int ho = r / geo.Wo, wo = r - ho * geo.Wo;                                                 \
                const int ys = ho * geo.stride + tap_y - geo.pad, xs = wo * geo.stride + tap_x - geo.pad;        \
                xval[k] = ys >= 0 && ys < geo.H && xs >= 0 && xs < geo.W;                                        \
                mx = ((long)n * geo.H + ys) * geo.W + xs;                                                        \
            }                                                                                                    \
            unsigned ox = xval[k] ? (unsigned)(mx * ldx * 2 + sub * 16) : OOB;                                   \
            xraw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(xrs, ox, 0, 0));           \
            yraw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(yrs, oy, 0, 0));           \
        }                                                                                                        \
    } while (0)

    const int g = lane >> 4, li = lane & 15, lq = li >> 2, lp = li & 3;
    const int frag_lane = (8 * g + lq) * PROW + (4 * lp) * 2;        // pixel row 8g + lq (+4), channel 4*lp of a 16-channel tile
    const unsigned char* a_frag = smem + (wci * MT) * CHB + frag_lane;                  // + (t>>1)*CHB + (t&1)*32 + ks*32*PROW
    const unsigned char* b_frag = smem + NCH * CHB + (wco * MT) * CHB + frag_lane;

    if (t_begin < t_end) UMI_ISSUE1(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        if (HAS_TX) {
            int opaque = 0;
            asm volatile("" : "+v"(opaque));
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = txs[j * PPP + sub + opaque];
#pragma unroll
            for (int k = 0; k < KP; ++k)
                if (xval[k]) {
                    xraw[k] = umi_tx8(xraw[k], t);
                }
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            *reinterpret_cast<half8*>(smem + lds_off + k * PXS * PROW) = xraw[k];
            *reinterpret_cast<half8*>(smem + NCH * CHB + lds_off + k * PXS * PROW) = yraw[k];
        }
        __syncthreads();
        if (tile + 1 < t_end) UMI_ISSUE1(tile + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 af[T16], bf[T16];
#pragma unroll
            for (int t = 0; t < T16; ++t) af[t] = tr_frag(a_frag + (t >> 1) * CHB + (t & 1) * 32 + ks * 32 * PROW);
#pragma unroll
            for (int t = 0; t < T16; ++t) bf[t] = tr_frag(b_frag + (t >> 1) * CHB + (t & 1) * 32 + ks * 32 * PROW);
#pragma unroll
            for (int ta = 0; ta < T16; ++ta)
#pragma unroll
                for (int tb = 0; tb < T16; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[ta], bf[tb], acc[ta][tb], 0, 0, 0);
        }
        __syncthreads();
    }
#undef UMI_ISSUE1
#pragma unroll
    for (int ta = 0; ta < T16; ++ta)
#pragma unroll
        for (int tb = 0; tb < T16; ++tb) {
            // channel counts that are only multiples of 8 (plain 1x1 mode): the operands of the missing channels were read
            // from whatever follows in memory (or zeros past the end), which only reaches accumulators that are not stored
            const int co = co0 + (wco * T16 + tb) * 16 + (lane & 15);
            if (co >= Co) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = ci0 + (wci * T16 + ta) * 16 + 4 * (lane >> 4) + r;
                if constexpr (GROUP) {
                    if (ci < Ci) grp.dW[blockIdx.z][(long)co * grp.s_co + (long)ci * grp.s_ci] = acc[ta][tb][r] * grp.scale;
                } else {
                    if (ci < Ci) part[(((long)blockIdx.y * gridDim.z + blockIdx.z) * Ci + ci) * Co + co] = acc[ta][tb][r];
                }
            }
        }
}

void plan1(long M, int Ci, int Co, int TM, int* tiles_total, int* splits, int* tps, int taps = 1) {
    *tiles_total = (int)((M + 63) / 64);
    const long pairs = (long)((Ci + TM - 1) / TM) * ((Co + TM - 1) / TM) * taps;
    static const long target2 = [] { const char* e = getenv("UMI_WGRAD_SPLIT_TARGET2"); long v = e ? atol(e) : 0; return v > 0 ? v : 1024L; }();
    long want = (target2 + pairs - 1) / pairs;
    const long slab = (long)taps * Ci * Co * 4;
    long cap = (64L << 20) / slab;
    if (cap < 1) cap = 1;
    if (pairs * cap < 512 && pairs < 512) cap = (512 + pairs - 1) / pairs;
    if (want > cap) want = cap;
    if (want > *tiles_total) want = *tiles_total;
    if (want < 1) want = 1;
    *tps = (int)((*tiles_total + want - 1) / want);
    *splits = (*tiles_total + *tps - 1) / *tps;
}

}  // namespace

bool umi_wgrad1x1_mfma_ok(long M, int Ci, int Co, int R, int S, int stride, int pad, int ldx, int lddy, int dtype, int flags,
                          const void* txb) {
    if (flags & UMI_CONV_FORCE_GENERIC) return false;
    if (dtype != UMI_F16 || txb) return false;
    if (R != 1 || S != 1 || stride != 1 || pad != 0) return false;
    if (Ci % 8 || Co % 8 || Ci < 16 || Co < 16 || ldx % 8 || lddy % 8) return false;   // partial 64-channel tiles are masked
    if (M * (long)(ldx > lddy ? ldx : lddy) * 2 >= 0x7FFFFFF0L) return false;     // 32-bit buffer offsets
    return true;
}
// 128 x 128 channel tiles only when there are enough of them to fill the chip without a deep split-K: every split is a
// slab written and re-read by the reduction (a 768 x 768 linear over 4,704 tokens: 36 tiles x 25 splits = 59 MB of
// slabs with 128-tiles, 144 x 8 = 19 MB with 64-tiles)
static int wgrad1x1_tm(int Ci, int Co) {
    return (Ci % 128 == 0 && Co % 128 == 0 && (long)(Ci / 128) * (Co / 128) >= 96) ? 128 : 64;
}

size_t umi_wgrad1x1_mfma_ws_bytes(long M, int Ci, int Co) {
    int tt, splits, tps;
    plan1(M, Ci, Co, wgrad1x1_tm(Ci, Co), &tt, &splits, &tps);
    return (size_t)splits * Ci * Co * sizeof(float);
}

int umi_wgrad1x1_mfma(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co, long s_ci,
                      long s_t, float out_scale, long M, int Ci, int Co, void* ws, size_t ws_bytes, hipStream_t s) {
    const int TM = wgrad1x1_tm(Ci, Co);
    int tt, splits, tps;
    plan1(M, Ci, Co, TM, &tt, &splits, &tps);
    if (ws_bytes < (size_t)splits * Ci * Co * sizeof(float)) return UMI_ERR_WORKSPACE;
    if (((uintptr_t)x | (uintptr_t)dy) & 15) return UMI_ERR_BADARG;
    const int n_co_t = (Co + TM - 1) / TM;
    dim3 grid(((Ci + TM - 1) / TM) * n_co_t, splits), block(256);
    const WGeo geo{1, 1, 1, 1, 1, 1, 0};
#define GO(T_, H_) hipLaunchKernelGGL((wgrad1x1_mfma_kernel<T_, H_, false>), grid, block, 0, s, (const half_t*)x, ldx, (const float4*)txa, (const half_t*)dy, lddy, (float*)ws, M, Ci, Co, tt, tps, n_co_t, geo, WNoGroup{})
    if (TM == 128) { if (txa) GO(128, true); else GO(128, false); }
    else { if (txa) GO(64, true); else GO(64, false); }
#undef GO
    UMI_LAUNCH_CHECK();
    umi_launch_wgrad_reduce((const float*)ws, splits, 1, Ci, Co, dW, s_co, s_ci, s_t, out_scale, s);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

// Weight gradients of `n` pointwise convs / linears of one shape in one launch per 16 (see GROUP above)
int umi_wgrad1x1_mfma_group(int n, const void* const* x, int ldx, const void* const* dy, int lddy, float* const* dW, long s_co,
                            long s_ci, float out_scale, long M, int Ci, int Co, hipStream_t s) {
    const bool t128 = Ci % 128 == 0 && Co % 128 == 0 && (long)(Ci / 128) * (Co / 128) * (n < 16 ? n : 16) >= 96;
    const int TM = t128 ? 128 : 64;
    const int tt = (int)((M + 63) / 64), n_co_t = (Co + TM - 1) / TM;
    const WGeo geo{1, 1, 1, 1, 1, 1, 0};
    for (int g0 = 0; g0 < n; g0 += 16) {
        const int cnt = n - g0 < 16 ? n - g0 : 16;
        WGroup grp;
        for (int i = 0; i < 16; ++i) {
            const int j = g0 + (i < cnt ? i : 0);
            if (((uintptr_t)x[j] | (uintptr_t)dy[j]) & 15) return UMI_ERR_BADARG;
            grp.x[i] = (const half_t*)x[j]; grp.dy[i] = (const half_t*)dy[j]; grp.dW[i] = dW[j];
        }
        grp.s_co = s_co; grp.s_ci = s_ci; grp.scale = out_scale;
        dim3 grid(((Ci + TM - 1) / TM) * n_co_t, 1, cnt), block(256);
        if (TM == 128)
            hipLaunchKernelGGL((wgrad1x1_mfma_kernel<128, false, false, true>), grid, block, 0, s, (const half_t*)nullptr, ldx,
                               (const float4*)nullptr, (const half_t*)nullptr, lddy, (float*)nullptr, M, Ci, Co, tt, tt, n_co_t, geo, grp);
        else
            hipLaunchKernelGGL((wgrad1x1_mfma_kernel<64, false, false, true>), grid, block, 0, s, (const half_t*)nullptr, ldx,
                               (const float4*)nullptr, (const half_t*)nullptr, lddy, (float*)nullptr, M, Ci, Co, tt, tt, n_co_t, geo, grp);
        UMI_LAUNCH_CHECK();
    }
    return UMI_OK;
}

// ---- tap-gather weight gradient (strided / padded R x S convs) ----------------------------------------------------------
bool umi_wgrad_gather_mfma_ok(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo, int ldx,
                              int lddy, int dtype, int flags, const void* txb) {
    if (flags & UMI_CONV_FORCE_GENERIC) return false;
    if (dtype != UMI_F16 || txb) return false;
    if (R * S > 49 || stride < 1 || pad < 0) return false;
    if (Ho != (H + 2 * pad - R) / stride + 1 || Wo != (W + 2 * pad - S) / stride + 1) return false;
    if (Ci % 64 || Co % 64 || ldx % 8 || lddy % 8) return false;
    if ((long)N * Ho * Wo * lddy * 2 >= 0x7FFFFFF0L || (long)N * H * W * ldx * 2 >= 0x7FFFFFF0L) return false;
    return true;
}

size_t umi_wgrad_gather_mfma_ws_bytes(int N, int Ho, int Wo, int Ci, int Co, int R, int S) {
    int tt, splits, tps;
    plan1((long)N * Ho * Wo, Ci, Co, wgrad1x1_tm(Ci, Co), &tt, &splits, &tps, R * S);
    return (size_t)splits * R * S * Ci * Co * sizeof(float);
}

int umi_wgrad_gather_mfma(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co, long s_ci,
                          long s_t, float out_scale, int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad,
                          int Ho, int Wo, void* ws, size_t ws_bytes, hipStream_t s) {
    const int TM = wgrad1x1_tm(Ci, Co), taps = R * S;
    const long M = (long)N * Ho * Wo;
    int tt, splits, tps;
    plan1(M, Ci, Co, TM, &tt, &splits, &tps, taps);
    if (ws_bytes < (size_t)splits * taps * Ci * Co * sizeof(float)) return UMI_ERR_WORKSPACE;
    if (((uintptr_t)x | (uintptr_t)dy) & 15) return UMI_ERR_BADARG;
    const int n_co_t = Co / TM;
    dim3 grid((Ci / TM) * n_co_t, splits, taps), block(256);
    const WGeo geo{Ho, Wo, H, W, S, stride, pad};
#define GO(T_, H_) hipLaunchKernelGGL((wgrad1x1_mfma_kernel<T_, H_, true>), grid, block, 0, s, (const half_t*)x, ldx, (const float4*)txa, (const half_t*)dy, lddy, (float*)ws, M, Ci, Co, tt, tps, n_co_t, geo, WNoGroup{})
    if (TM == 128) { if (txa) GO(128, true); else GO(128, false); }
    else { if (txa) GO(64, true); else GO(64, false); }
#undef GO
    UMI_LAUNCH_CHECK();
    umi_launch_wgrad_reduce((const float*)ws, splits, taps, Ci, Co, dW, s_co, s_ci, s_t, out_scale, s);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

bool umi_wgradT_mfma_ok(int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo, int ldx,
                        int lddy, int dtype, int flags, const void* txa) {
    if (flags & UMI_CONV_FORCE_GENERIC) return false;
    if (dtype != UMI_F16 || txa) return false;
    if (R != 2 || S != 2 || stride != 2 || pad != 0 || H != 2 * Ho || W != 2 * Wo) return false;
    if (Ci % 64 || Co % 64 || ldx % 8 || lddy % 8) return false;
    return true;
}

size_t umi_wgradT_mfma_ws_bytes(int N, int Ho, int Wo, int Ci, int Co) {
    int tx_, ty_, tt, splits, tps;
    planT(N, Ho, Wo, Ci, Co, &tx_, &ty_, &tt, &splits, &tps);
    return (size_t)splits * 4 * Ci * Co * sizeof(float) + (size_t)splits * 2 * Ci * sizeof(float);    // slabs + bias-gradient rows
}

// one-shot request (set by umi_conv_wgrad_bias, consumed by the next umi_wgradT_mfma on this thread): also produce the hi-res
// operand's column sums, scaled like dW
static thread_local float* g_wgT_bias = nullptr;
void umi_wgradT_bias_set(float* bias_out) { g_wgT_bias = bias_out; }
void umi_launch_reduce_rows2(const float* ws, int rows, int C, float* out0, float* out1, float scale, hipStream_t s);

int umi_wgradT_mfma(const void* x, int ldx, const void* dy, int lddy, const void* txb, float* dW, long s_co, long s_ci,
                    long s_t, float out_scale, int N, int Ho, int Wo, int Ci, int Co, void* ws, size_t ws_bytes,
                    hipStream_t s) {
    int tiles_x, tiles_y, tiles_total, splits, tps;
    planT(N, Ho, Wo, Ci, Co, &tiles_x, &tiles_y, &tiles_total, &splits, &tps);
    float* const bias_out = g_wgT_bias;
    g_wgT_bias = nullptr;
    if (ws_bytes < (size_t)splits * 4 * Ci * Co * sizeof(float) + (bias_out ? (size_t)splits * 2 * Ci * sizeof(float) : 0)) return UMI_ERR_WORKSPACE;
    if (((uintptr_t)x | (uintptr_t)dy) & 15) return UMI_ERR_BADARG;
    float* const cpart = bias_out ? (float*)ws + (size_t)splits * 4 * Ci * Co : nullptr;
    const bool wide = Co % 128 == 0;
    const int n_cy_t = wide ? Co / 128 : Co / 64;
    dim3 grid((Ci / 64) * n_cy_t, splits), block(256);
#define UMI_GO_T(HT, YH_)                                                                                        \
    hipLaunchKernelGGL((wgradT2x2_mfma_kernel<HT, YH_>), grid, block, 0, s, (const half_t*)x, ldx, (const half_t*)dy, lddy, \
                       (const float4*)txb, (float*)ws, N, Ho, Wo, Ci, Co, tiles_x, tiles_y, tiles_total, tps, n_cy_t, cpart)
    if (txb) { if (wide) UMI_GO_T(true, 2); else UMI_GO_T(true, 1); }
    else { if (wide) UMI_GO_T(false, 2); else UMI_GO_T(false, 1); }
#undef UMI_GO_T
    if (bias_out) {
        UMI_LAUNCH_CHECK();
        umi_launch_reduce_rows2(cpart, splits, Ci, bias_out, nullptr, out_scale, s);
    }
    UMI_LAUNCH_CHECK();
    umi_launch_wgrad_reduce((const float*)ws, splits, 4, Ci, Co, dW, s_co, s_ci, s_t, out_scale, s);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

bool umi_wgrad3x3_mfma_ok(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo,
                          int ldx, int lddy, int dtype, int flags, const void* txb) {
    if (flags & UMI_CONV_FORCE_GENERIC) return false;
    if (dtype != UMI_F16 || txb) return false;
    if (R != 3 || S != 3 || stride != 1 || pad != 1 || Ho != H || Wo != W) return false;
    if (Ci % 8 || Co % 8 || ldx % 8 || lddy % 8) return false;      // partial 64-channel tiles are masked in the kernel
    if ((long)H * W * (ldx > lddy ? ldx : lddy) * 2 >= 0x7FFFFFF0L) return false;   // 32-bit offsets inside one image
    return true;
}

size_t umi_wgrad3x3_mfma_ws_bytes(int N, int H, int W, int Ci, int Co) {
    int tx_, ty_, tt, splits, tps;
    plan(N, H, W, Ci, Co, &tx_, &ty_, &tt, &splits, &tps);
    return (size_t)splits * 9 * Ci * Co * sizeof(float);
}

static int wgrad3x3_launch(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co,
                           long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci, int Co, void* ws,
                           size_t ws_bytes, hipStream_t s, const BnApply* bna);

int umi_wgrad3x3_mfma(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co,
                      long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci, int Co, void* ws,
                      size_t ws_bytes, hipStream_t s) {
    return wgrad3x3_launch(x, ldx, txa, dy, lddy, dW, s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, ws, ws_bytes, s, nullptr);
}

// weight gradient fused with stage 3 of the BatchNorm + ReLU backward of the conv's own output (see BnApply above)
int umi_wgrad3x3_mfma_bnapply(const void* x, int ldx, const void* txa, const void* da, int ldda, const void* ybn, int ldybn,
                              const void* txbn, const float* rstd, const float* sum_dz, const float* sum_dzx, void* dz,
                              int lddz, float* dW, long s_co, long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci,
                              int Co, void* ws, size_t ws_bytes, hipStream_t s) {
    const BnApply b{(const half_t*)ybn, ldybn, (const float4*)txbn, rstd, sum_dz, sum_dzx, (long)N * H * W,
                    (half_t*)dz, lddz};
    return wgrad3x3_launch(x, ldx, txa, da, ldda, dW, s_co, s_ci, s_t, out_scale, N, H, W, Ci, Co, ws, ws_bytes, s, &b);
}

static int wgrad3x3_launch(const void* x, int ldx, const void* txa, const void* dy, int lddy, float* dW, long s_co,
                           long s_ci, long s_t, float out_scale, int N, int H, int W, int Ci, int Co, void* ws,
                           size_t ws_bytes, hipStream_t s, const BnApply* bna) {
    int tiles_x, tiles_y, tiles_total, splits, tps;
    plan(N, H, W, Ci, Co, &tiles_x, &tiles_y, &tiles_total, &splits, &tps);
    if (ws_bytes < (size_t)splits * 9 * Ci * Co * sizeof(float)) return UMI_ERR_WORKSPACE;
    if (((uintptr_t)x | (uintptr_t)dy) & 15) return UMI_ERR_BADARG;
    const int n_co_t = (Co + 63) / 64;
    dim3 grid(((Ci + 63) / 64) * n_co_t, splits), block(256);
    // fabric bytes per pixel tile: x (with its halo, 1.59x) once per XCD that touches a ci tile, dy once per XCD that touches a
    // co tile: ci fastest costs 1.59 Ci + 8 Co, co fastest 8 * 1.59 Ci + Co
    static const int force_fast = [] { const char* e = getenv("UMI_WGRAD_FAST_CI"); return e ? atoi(e) : -1; }();
    const int fast_ci = force_fast >= 0 ? force_fast : (Co < 1.59 * Ci ? 1 : 0);
    static const bool classic = [] { const char* e = getenv("UMI_WGRAD_CLASSIC"); return e && e[0] == '1'; }();
    if (bna && classic) return UMI_ERR_UNSUPPORTED;
    if (!classic) {
        constexpr int dyn = WS_SMEM + 64 * (int)sizeof(float4);
        static const int attr_rc = [] {
            int a = (int)hipFuncSetAttribute((const void*)wgrad3x3_ws_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
            int b = (int)hipFuncSetAttribute((const void*)wgrad3x3_ws_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
            int c = (int)hipFuncSetAttribute((const void*)wgrad3x3_ws_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
            int d = (int)hipFuncSetAttribute((const void*)wgrad3x3_ws_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, dyn);
            return a ? a : (b ? b : (c ? c : d));
        }();
        if (attr_rc) return attr_rc;
        dim3 block_ws(512);
        const BnApply ba = bna ? *bna : BnApply{nullptr, 0, nullptr, nullptr, nullptr, nullptr, 1, nullptr, 0};
#define GO_WS(HT, BN_)                                                                                              \
        hipLaunchKernelGGL((wgrad3x3_ws_kernel<HT, BN_>), grid, block_ws, dyn, s, (const half_t*)x, ldx, (const float4*)txa,   \
                           (const half_t*)dy, lddy, (float*)ws, N, H, W, Ci, Co, tiles_x, tiles_y, tiles_total, tps, n_co_t, fast_ci, ba)
        if (bna) { if (txa) GO_WS(true, true); else GO_WS(false, true); }
        else { if (txa) GO_WS(true, false); else GO_WS(false, false); }
#undef GO_WS
    } else if (txa)
        hipLaunchKernelGGL(wgrad3x3_mfma_kernel<true>, grid, block, 0, s, (const half_t*)x, ldx, (const float4*)txa,
                           (const half_t*)dy, lddy, (float*)ws, N, H, W, Ci, Co, tiles_x, tiles_y, tiles_total, tps,
                           n_co_t, fast_ci);
    else
        hipLaunchKernelGGL(wgrad3x3_mfma_kernel<false>, grid, block, 0, s, (const half_t*)x, ldx, (const float4*)txa,
                           (const half_t*)dy, lddy, (float*)ws, N, H, W, Ci, Co, tiles_x, tiles_y, tiles_total, tps,
                           n_co_t, fast_ci);
    UMI_LAUNCH_CHECK();
    umi_launch_wgrad_reduce((const float*)ws, splits, 9, Ci, Co, dW, s_co, s_ci, s_t, out_scale, s);
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
