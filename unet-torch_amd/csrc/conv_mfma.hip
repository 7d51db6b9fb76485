// conv3x3 (stride 1, pad 1) forward / dgrad on the gfx950 matrix cores, fp16 storage, fp32 accumulate.
//
// Implicit GEMM, D[co][pixel] += W[co][k] * X[k][pixel] with v_mfma_f32_32x32x16_f16:
//   * A operand = weights  (M = 32 output channels, K = 16 input channels of one tap)
//   * B operand = pixels   (N = 32 consecutive pixels of one image row, same K)
// so the im2col shift of a tap is just an LDS address offset ("im2col in register").
//
// Workgroup = 256 threads = 4 waves, 2 workgroups per CU.  Tile = TH x 32 output pixels x BN channels:
//   BN = 128: TH = 8  (waves 2 x 2)      BN = 64: TH = 16 (waves 4 x 1)
// every wave owns 4 image rows x 64 channels = acc[2][4] 32x32 tiles (128 accumulator registers).
//
// Per 16-input-channel chunk the workgroup stages, through registers,
//   * the (TH+2) x 34 halo tile of the input with the producer's BatchNorm+ReLU applied on the fly
//     (consumer-side transform, zero padding applied AFTER it) and
//   * the chunk's 9 taps x BN x 16 weights
// into LDS with 48-byte rows (32 B data + 16 B pad: an odd number of 16-B slots, so the 16-lane
// groups of ds_read_b128 fall on distinct slots -> conflict free for both operands).
// Global loads of chunk c+1 are issued before the MFMA phase of chunk c and land in registers while it
// runs (issue-early / write-late staging); the second resident workgroup covers what is left.
// Each tap column dx reuses 6 pixel-row fragments for its 3 taps x 4 rows.
//
// Epilogue: accumulators -> fp16 -> LDS tile [pixel][BN] -> coalesced 16-B global stores, and the
// per-channel sum / sum-of-squares of the *stored* values (BatchNorm statistics) are taken column-wise
// from that LDS tile and written as one deterministic partial row per pixel tile.
#include "common.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifdef UMI_STAMP
// diagnostic build only: per-wave cycle sums of the main-loop segments (never compiled into the shipped library)
__device__ unsigned long long umi_stamp_buf[2048 * 8];
#define UMI_T(var)                                                                        \
    unsigned long long var;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0)
extern "C" int umi_debug_read_stamps(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(umi_stamp_buf), sizeof(umi_stamp_buf));
}
#endif

namespace {

constexpr int HALO_W = 34;      // 32 + 2
constexpr int ROWB = 48;        // LDS bytes per pixel / weight row (16 halfs + 16 B pad)

template <int TH, int BN>
struct Cfg {
    static constexpr int WN = BN / 64;
    static constexpr int WM = 4 / WN;
    static_assert(TH / WM == 4, "every wave owns 4 image rows");
    static constexpr int HALO_PIX = (TH + 2) * HALO_W;
    static constexpr int HB = HALO_PIX * ROWB;
    static constexpr int WB = 9 * BN * ROWB;
    static constexpr int P = TH * 32;
    static constexpr int ERS = BN * 2 + 16;          // epilogue LDS row stride (bytes)
    static constexpr int EB = P * ERS;
    static constexpr int SMEM = (HB + WB) > EB ? (HB + WB) : EB;
    static constexpr int NPH = 2 * HALO_PIX;         // 16-B pieces of the halo tile per chunk
    static constexpr int KPH = (NPH + 255) / 256;
    static constexpr int NPW = 9 * BN * 2;           // 16-B pieces of the weights per chunk
    static constexpr int KPW = (NPW + 255) / 256;
};

// Epilogue reductions written as one partial row per pixel tile, part[tile][2][Co]:
//   EPI 1: sum / sum of squares of the stored outputs (BatchNorm statistics of THIS conv's output, forward);
//   EPI 2: this launch is the data gradient that produces d(activated output) of a BatchNorm+ReLU layer whose raw output
//          is `bn.y`: sum dz and sum dz*xhat of that layer (dz = stored value * [tx(y) > lo]), i.e. stage 1 of its
//          BatchNorm backward without re-reading the gradient tensor.
struct BnRed { const half_t* y; int ld; const float4* tx; const float* rstd; };

// OPT 0: the round-1 schedule.  OPT 1 (round 3): branch-free staging (padding / surplus pieces go to a dummy LDS address fixed in
// the prologue, weight-piece validity folded into a per-lane base offset, the tap step in the scalar offset) and a pinned
// ds_read / MFMA interleave in the MFMA phase (fragment reads issued two MFMA pairs ahead of their use instead of
// "read, wait lgkmcnt(0), 4 MFMAs").  Same MFMA order, bit-identical outputs.
template <int TH, int BN, bool HAS_TX, int EPI, int OPT>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(
    const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx, const half_t* __restrict__ wp8,
    half_t* __restrict__ y, int ldy, float* __restrict__ part, int N, int H, int W, int Ci, int Co, int tiles_x,
    int tiles_y, int n_co, int xcd_chunk, BnRed bn) {
    using C = Cfg<TH, BN>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::SMEM];
    // consumer-transform rows of the current / next chunk (16 channels x float4), refilled two chunks ahead so the
    // transform never waits on a global load (measured with in-kernel stamps: 8 dependent tx loads per chunk cost
    // ~1,650 of the ~7,200 cycles of a main-loop iteration)
    __shared__ float4 txbuf[2][16];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % C::WN, wm = wave / C::WN;
#ifdef UMI_STAMP
    UMI_T(t_start);
#endif

    if (OPT == 2) {
        // experiment: the two workgroups of a CU start together and so reach their prologues, staging phases and epilogues
        // together.  The workgroup whose wave 0 sits in an odd wave slot waits half a workgroup lifetime, once, in the first
        // round of the launch; successors inherit the offset.
        if (blockIdx.x < 512) {
            __shared__ int stag;
            if (tid == 0) {
                unsigned hwid;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                stag = (int)(hwid & 1u);
            }
            __syncthreads();
            if (stag) {
                const unsigned long long t0 = __builtin_amdgcn_s_memtime();
                const unsigned long long wait = 6000ull + 1400ull * (unsigned)(Ci >> 4);
                while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
            }
        }
    }
    // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so id -> work item
    // is permuted to give every XCD a contiguous range of (pixel tile, channel block) items: the channel blocks of one
    // pixel tile then share an L2 and the input tile crosses the fabric once instead of once per channel block
    int wid = blockIdx.x;
    if (xcd_chunk > 0 && wid < 8 * xcd_chunk) wid = (wid & 7) * xcd_chunk + (wid >> 3);
    const int cb = wid % n_co;
    const int pt = wid / n_co;
    const int n = pt / (tiles_x * tiles_y);
    const int rem = pt - n * tiles_x * tiles_y;
    const int ty0 = (rem / tiles_x) * TH, tx0 = (rem % tiles_x) * 32;
    const int c0 = cb * BN;
    const int cvalid = Co - c0 < BN ? Co - c0 : BN;   // output channels of this tile that exist (Co % 8 == 0, e.g. Co = 16)
    // Staging-thread -> LDS-row mapping.  ds_write_b128 is serviced in groups of 8 consecutive lanes against 32 banks: with
    // the 48-byte rows, 8 lanes covering 4 rows x both halves collide 2-way on half the banks (measured: 25-30 % of the
    // kernel's LDS-array cycles were SQ_LDS_BANK_CONFLICT, profiles/r01_conv_fwd_lds_mfma.json); 8 lanes covering 8
    // consecutive rows of ONE half fall on 32 distinct banks.  Global coalescing is unchanged: a wave still touches the same
    // 32 rows x 32 bytes.
    const int q = (tid >> 3) & 1;                  // which 8-channel half of the 16-channel chunk this thread stages
    const int srow = ((tid >> 4) << 3) | (tid & 7);   // this thread's row among the 128 staged per pass

    // ---- per-thread staging plan (fixed across chunks) ------------------------------------------
    // Loads go through buffer descriptors: an out-of-range voffset returns zeros, so zero padding and partial
    // tiles need no branches around the loads.  The per-chunk offset rides in the scalar soffset.
    // halo piece k of this thread: index i = tid + 256k -> halo pixel hp = (tid>>1) + 128k, channel half q
    constexpr unsigned OOB = 0x7FFFFFFFu;
    unsigned hoff[C::KPH];                         // byte offset inside image n, or OOB (zero padding / no piece)
#pragma unroll
    for (int k = 0; k < C::KPH; ++k) {
        int hp = srow + 128 * k;
        int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        bool inimg = (hp < C::HALO_PIX) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        hoff[k] = inimg ? (unsigned)(gy * W + gx) * (unsigned)(ldx * 2) + q * 16 : OOB;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (long)n * H * W * ldx), 0, (int)((long)H * W * ldx * 2), 0x00020000);
    const int hl_base = srow * ROWB + q * 16;                 // + k * 128 * ROWB
    // OPT 1: the LDS byte address every halo piece is stored at, chunk after chunk.  Pieces outside the image (zero padding)
    // and surplus pieces (hp >= HALO_PIX) are stored -- transformed garbage and all -- into the 16 pad bytes of a row, which
    // nothing reads; the padding slots themselves are zeroed once, here, and never written again.
    int hl[C::KPH];
    if (OPT >= 1) {
#pragma unroll
        for (int k = 0; k < C::KPH; ++k) {
            const int hp = srow + 128 * k;
            const int hpc = hp < C::HALO_PIX ? hp : C::HALO_PIX - 1;
            hl[k] = hoff[k] != OOB ? hp * ROWB + q * 16 : hpc * ROWB + 32;
            if (hp < C::HALO_PIX && hoff[k] == OOB) {
                half8 z;
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (half_t)0.f;
                *reinterpret_cast<half8*>(smem + hp * ROWB + q * 16) = z;
            }
        }
    }
    // weight piece k: row rc = (tid>>1) + 128k of the [9*BN] rows -> tap = k*(128/BN) + (tid>>1)/BN
    constexpr int TSTEP = 128 / BN;
    const int tap0 = srow / BN, wcol = srow % BN;
    const bool wok = wcol < cvalid;                // weight rows past Co read as zeros
    const int Ci8 = Ci >> 3;
    const unsigned wbase = (unsigned)(((tap0 * Ci8 + q) * Co + wcol) * 16);
    const unsigned wstep = (unsigned)(TSTEP * Ci8 * Co * 16);   // bytes per k step
    const int wstep_s = __builtin_amdgcn_readfirstlane((int)wstep);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(wp8 + (long)c0 * 8), 0, (int)((long)9 * Ci * Co * 2 - (long)c0 * 16), 0x00020000);
    const int wl_base = C::HB + srow * ROWB + q * 16;          // + k * 128 * ROWB
    // OPT 1: validity of a weight piece is a property of the lane (channel past Co) except for the last piece of the BN = 64
    // tile (tap 8 + tap0), so it lives in the per-lane base offset; WBAD + any tap step stays beyond the buffer's range.
    constexpr unsigned WBAD = 0x40000000u;
    constexpr bool W_LAST_PARTIAL = (C::KPW - 1) * TSTEP + (TSTEP - 1) >= 9;      // BN = 64: piece 4 exists for tap0 = 0 only
    const unsigned wbase_v = wok ? wbase : WBAD;
    const unsigned wbase_l = (wok && tap0 + (C::KPW - 1) * TSTEP < 9) ? wbase : WBAD;
    const int wl_last = (tap0 + (C::KPW - 1) * TSTEP < 9) ? wl_base + (C::KPW - 1) * 128 * ROWB : C::HB + srow * ROWB + 32;

    floatx16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    half8 hraw[C::KPH], wraw[C::KPW];
    float4 txr = make_float4(0.f, 1.f, 0.f, 0.f);
#define UMI_ISSUE(c_)                                                                                              \
    do {                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < C::KPH; ++k)                                                        \
            hraw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(xrs, hoff[k], (c_) * 32, 0)); \
        if (OPT >= 1) {                                                                                           \
            _Pragma("unroll") for (int k = 0; k < C::KPW; ++k)                                                    \
                wraw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(                        \
                    wrs, (W_LAST_PARTIAL && k == C::KPW - 1) ? wbase_l : wbase_v,                                 \
                    (c_) * 2 * Co * 16 + k * wstep_s, 0));                                                        \
        } else {                                                                                                  \
            _Pragma("unroll") for (int k = 0; k < C::KPW; ++k)                                                    \
                wraw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(                        \
                    wrs, (wok && tap0 + k * TSTEP < 9) ? wbase + k * wstep : OOB, (c_) * 2 * Co * 16, 0));        \
        }                                                                                                         \
    } while (0)

    // fragment base addresses (bytes)
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int b_base = ((wm * 4) * HALO_W + lrow) * ROWB + lhalf * 16;               // + ((nt+dy)*34 + dx)*48
    const int a_base = C::HB + (wn * 64 + lrow) * ROWB + lhalf * 16;                 // + (tap*BN + mt*32)*48

    const int nchunks = Ci >> 4;
    // (Rotating the chunk order per workgroup to spread the weight reads over L2 channels was measured and is
    //  SLOWER: -12 % on 1024->1024; simultaneous readers of one panel share L2 lines.)
#define UMI_CHUNK(i_) (i_)
    // OPT 1: the transform rows are handled by the whole of wave 0 (4 lanes per row, same value): a scalar branch, no
    // exec-mask juggling in the loop
    const bool tx_wave = OPT >= 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) == 0 : tid < 16;
    const int txi = OPT >= 1 ? (lane & 15) : tid;                      // row of the chunk this thread carries
    const int txs = (txi & 7) * 2 + (txi >> 3);                        // its slot: [j][q]
    if (HAS_TX) {
        if (tx_wave) {
            // stored transposed ([j][q]: the two channel halves of a chunk side by side) so that the per-chunk reads of
            // lanes q = 0 / 1 fall on different LDS banks
            txbuf[0][txs] = tx[txi];
            if (nchunks > 1) txbuf[1][txs] = tx[16 + txi];
        }
        __syncthreads();
    }
    UMI_ISSUE(UMI_CHUNK(0));
    if (HAS_TX && nchunks > 2 && tx_wave) txr = tx[2 * 16 + txi];
#ifdef UMI_STAMP
    unsigned long long seg[5] = {0, 0, 0, 0, 0};
    UMI_T(t_loop);
#endif
    for (int ci_ = 0; ci_ < nchunks; ++ci_) {
        const int c = UMI_CHUNK(ci_);
#ifdef UMI_STAMP
        UMI_T(t0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        UMI_T(t0b);
#endif
        // ---- registers -> (transform) -> LDS ----------------------------------------------------
        if (HAS_TX) {
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = txbuf[ci_ & 1][j * 2 + q];
#pragma unroll
            for (int k = 0; k < C::KPH; ++k) {
                if (OPT >= 1) hraw[k] = umi_tx8(hraw[k], t);
                else if (hoff[k] != OOB) {
                    hraw[k] = umi_tx8(hraw[k], t);
                }
            }
        }
#ifndef UMI_EXP_NO_STAGE
        if (OPT >= 1) {
#pragma unroll
            for (int k = 0; k < C::KPH; ++k) *reinterpret_cast<half8*>(smem + hl[k]) = hraw[k];
#pragma unroll
            for (int k = 0; k < C::KPW; ++k)
                *reinterpret_cast<half8*>(smem + ((W_LAST_PARTIAL && k == C::KPW - 1) ? wl_last : wl_base + k * 128 * ROWB)) = wraw[k];
        } else {
#pragma unroll
        for (int k = 0; k < C::KPH; ++k)
            if (srow + 128 * k < C::HALO_PIX) *reinterpret_cast<half8*>(smem + hl_base + k * 128 * ROWB) = hraw[k];
#pragma unroll
        for (int k = 0; k < C::KPW; ++k)
            if (tap0 + k * TSTEP < 9) *reinterpret_cast<half8*>(smem + wl_base + k * 128 * ROWB) = wraw[k];
        }
#else
#pragma unroll
        for (int k = 0; k < C::KPH; ++k) asm volatile("" ::"v"(hraw[k]));
#pragma unroll
        for (int k = 0; k < C::KPW; ++k) asm volatile("" ::"v"(wraw[k]));
#endif
#ifdef UMI_STAMP
        UMI_T(t1);
#endif
        __syncthreads();
#ifdef UMI_STAMP
        UMI_T(t2);
#endif
        if (HAS_TX && tx_wave && ci_ + 2 < nchunks) {
            txbuf[ci_ & 1][txs] = txr;                        // every thread is past its reads of this buffer (barrier above)
            if (ci_ + 3 < nchunks) txr = tx[(ci_ + 3) * 16 + txi];
        }
        if (ci_ + 1 < nchunks) UMI_ISSUE(UMI_CHUNK(ci_ + 1));

        // ---- MFMA phase: 9 taps x (2 x 4) tiles -----------------------------------------------
        // raised wave priority for the MFMA phase: when the two waves of a SIMD compete, the one feeding the matrix pipe wins
        // over the other workgroup's staging VALU work (measured +3..5 %, +15 % on the 64-channel layers)
        __builtin_amdgcn_s_setprio(3);
#ifndef UMI_EXP_NO_MFMA
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            half8 bf[6];
#pragma unroll
            for (int rr = 0; rr < 6; ++rr)
                bf[rr] = *reinterpret_cast<const half8*>(smem + b_base + (rr * HALO_W + dx) * ROWB);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                half8 af[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    af[mt] = *reinterpret_cast<const half8*>(smem + a_base + ((dy * 3 + dx) * BN + mt * 32) * ROWB);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt], bf[nt + dy], acc[mt][nt], 0, 0, 0);
            }
        }
#endif
        if (OPT >= 1) {
            // issue order of the phase's 36 fragment reads and 72 MFMAs: 8 reads up front, then one read behind every MFMA pair
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i_ = 0; i_ < 28; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        __builtin_amdgcn_s_setprio(0);
#ifdef UMI_STAMP
        UMI_T(t3);
#endif
        __syncthreads();
#ifdef UMI_STAMP
        UMI_T(t4);
        seg[0] += t0b - t0; seg[1] += t1 - t0b; seg[2] += t2 - t1; seg[3] += t3 - t2; seg[4] += t4 - t3;
#endif
    }
#ifdef UMI_STAMP
    UMI_T(t_ep0);
#endif

    // ---- epilogue: acc -> fp16 LDS tile [pixel][BN] ----------------------------------------------
    // EPI 3 (inference): this conv's own BatchNorm (running statistics) + ReLU applied here, on the fp32 accumulators, so the
    // tensor is stored ACTIVATED and its consumers load it as it is (no statistics, no transform on load)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = wn * 64 + mt * 32 + g * 8 + lhalf * 4;
            float4 ot[4];
            if (EPI == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    ot[j] = co + j < cvalid ? bn.tx[c0 + co + j] : make_float4(0.f, 1.f, 0.f, 0.f);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int pix = (wm * 4 + nt) * 32 + lrow;
                half4 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = acc[mt][nt][g * 4 + j];
                    if (EPI == 3) v = umi_tx(v, ot[j]);
                    h[j] = (half_t)v;
                }
                *reinterpret_cast<half4*>(smem + pix * C::ERS + co * 2) = h;
            }
        }
    __syncthreads();

    // coalesced 16-B stores of the tile.  piece i = tid + 256k -> pixel p = i / PPR, 16-B column j = i % PPR; with PPR a
    // power of two p = p0 + k*PSTEP, so rows/cols advance by constants (no divisions in the loop).
    constexpr int PPR = BN / 8;                     // 16-B pieces per pixel row (16 or 8)
    constexpr int PSTEP = 256 / PPR;                // pixels advanced per k (16 or 32)
    constexpr int NK = C::P / PSTEP;                // 16
    const bool full_tile = (ty0 + TH <= H) && (tx0 + 32 <= W);
    // The same pass feeds the epilogue reductions (EPI): a thread keeps one 8-channel column group j for all its pixels, so
    // the per-channel sums accumulate in registers from the values it is storing anyway (no second LDS pass).
    constexpr int CG = PPR;                         // column groups (16 or 8)
    constexpr int SL = PSTEP;                       // pixel slices = threads per column group (16 or 32)
    const int cg = tid % PPR, sl = tid / PPR;
    float s[8], s2[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) s[jj] = s2[jj] = 0.f;
    {
        const int j = cg, p0 = sl;
        half_t* ybase = y + ((long)((long)n * H + ty0) * W + tx0) * ldy + c0 + j * 8;
        const unsigned char* sbase = smem + p0 * C::ERS + j * 16;
        const bool col_ok = j * 8 < cvalid;
        float4 t[8];
        float rs_[8];
        const half_t* yb = nullptr;
        if (EPI == 2 && col_ok) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) { t[jj] = bn.tx[c0 + j * 8 + jj]; rs_[jj] = bn.rstd[c0 + j * 8 + jj]; }
            yb = bn.y + ((long)((long)n * H + ty0) * W + tx0) * bn.ld + c0 + j * 8;
        }
        // one pixel of the pass: LDS -> global, and the epilogue reduction on the values on their way out
#define UMI_EPI_PIXEL(k_)                                                                                              \
        do {                                                                                                          \
            const int p = p0 + (k_) * PSTEP;                                                                          \
            const int row = p >> 5, col = p & 31;                                                                     \
            uint4 v = *reinterpret_cast<const uint4*>(sbase + (k_) * PSTEP * C::ERS);                                 \
            *reinterpret_cast<uint4*>(ybase + ((long)row * W + col) * ldy) = v;                                       \
            if (EPI == 1) {                                                                                           \
                const half8 hv = __builtin_bit_cast(half8, v);                                                        \
                _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) { float f = (float)hv[jj]; s[jj] += f; s2[jj] = fmaf(f, f, s2[jj]); } \
            } else if (EPI == 2) {                                                                                    \
                const half8 hv = __builtin_bit_cast(half8, v);                                                        \
                const half8 yv = *reinterpret_cast<const half8*>(yb + ((long)row * W + col) * bn.ld);                 \
                _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                    \
                    const float yy = (float)yv[jj];                                                                   \
                    const float dz = umi_tx_pre(yy, t[jj]) > t[jj].w ? (float)hv[jj] : 0.f;                           \
                    s[jj] += dz;                                                                                      \
                    s2[jj] = fmaf(dz, (yy - t[jj].x) * rs_[jj], s2[jj]);                                              \
                }                                                                                                     \
            }                                                                                                         \
        } while (0)
        // OPT 1: whole tiles (the benchmark's case) take a branch-free, fully unrolled pass: the 16 LDS reads go out together
        const bool whole = OPT >= 1 && __builtin_amdgcn_readfirstlane((int)(full_tile && cvalid == BN)) != 0;
        if (whole) {
#pragma unroll
            for (int k = 0; k < NK; ++k) UMI_EPI_PIXEL(k);
        } else {
#pragma unroll 8
            for (int k = 0; k < NK; ++k) {
                const int p_ = p0 + k * PSTEP;
                if (col_ok && (full_tile || (ty0 + (p_ >> 5) < H && tx0 + (p_ & 31) < W))) UMI_EPI_PIXEL(k);
            }
        }
#undef UMI_EPI_PIXEL
    }

    if (EPI == 1 || EPI == 2) {
        __syncthreads();                            // every thread is done with the tile: reuse it for the slice sums
        float* rs = reinterpret_cast<float*>(smem);         // [2][SL][BN]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            rs[(0 * SL + sl) * BN + cg * 8 + j] = s[j];
            rs[(1 * SL + sl) * BN + cg * 8 + j] = s2[j];
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, cc = tid % BN;
            float a = 0.f;
#pragma unroll 8
            for (int k = 0; k < SL; ++k) a += rs[(which * SL + k) * BN + cc];
            if (cc < cvalid) part[((long)pt * 2 + which) * Co + c0 + cc] = a;
        }
    }
#ifdef UMI_STAMP
    UMI_T(t_end);
    if (lane == 0 && blockIdx.x < 512) {
#pragma unroll
        for (int i = 0; i < 5; ++i) umi_stamp_buf[(blockIdx.x * 4 + wave) * 8 + i] = seg[i];
        umi_stamp_buf[(blockIdx.x * 4 + wave) * 8 + 5] = nchunks;
        umi_stamp_buf[(blockIdx.x * 4 + wave) * 8 + 6] = t_loop - t_start;
        umi_stamp_buf[(blockIdx.x * 4 + wave) * 8 + 7] = t_end - t_ep0;
    }
#endif
}

template <int TH, int BN, int OPT>
int launch(const void* x, int ldx, const void* tx, const void* wp8, void* y, int ldy, float* part, int N, int H, int W,
           int Ci, int Co, const BnRed* bnred, hipStream_t s, const void* out_tx = nullptr) {
    const int tiles_x = (W + 31) / 32, tiles_y = (H + TH - 1) / TH, n_co = (Co + BN - 1) / BN;
    const long nblk = (long)N * tiles_x * tiles_y * n_co;
    dim3 grid((unsigned)nblk), block(256);
    static const bool xcd_off = [] { const char* e = getenv("UMI_CONV_NO_XCD_ORDER"); return e && e[0] == '1'; }();
    const int xcd_chunk = (n_co > 1 && !xcd_off) ? (int)(nblk / 8) : 0;     // ids beyond 8 * chunk (the remainder) keep their order
    const BnRed bn = bnred ? *bnred : BnRed{nullptr, 0, (const float4*)out_tx, nullptr};
#define GO(HT, EP)                                                                                               \
    hipLaunchKernelGGL((conv3x3_mfma_kernel<TH, BN, HT, EP, OPT>), grid, block, 0, s, (const half_t*)x, ldx,          \
                       (const float4*)tx, (const half_t*)wp8, (half_t*)y, ldy, part, N, H, W, Ci, Co, tiles_x,   \
                       tiles_y, n_co, xcd_chunk, bn)
    if (out_tx) { if (tx) GO(true, 3); else GO(false, 3); }
    else if (bnred) { if (tx) GO(true, 2); else GO(false, 2); }
    else if (tx) { if (part) GO(true, 1); else GO(true, 0); }
    else    { if (part) GO(false, 1); else GO(false, 0); }
#undef GO
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

#undef UMI_ISSUE
#undef UMI_CHUNK
}  // namespace

// Shapes the MFMA path takes; everything else goes to the generic kernel.
bool umi_conv3x3_mfma_ok(int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo,
                         int ldx, int ldy, int in_dtype, int out_dtype, int flags, const float* bias) {
    if (flags & (UMI_CONV_UPSAMPLE2 | UMI_CONV_FORCE_GENERIC)) return false;
    if (in_dtype != UMI_F16 || out_dtype != UMI_F16 || bias) return false;
    if (R != 3 || S != 3 || stride != 1 || pad != 1 || Ho != H || Wo != W) return false;
    if (Ci % 16 || Co % 8 || ldx % 8 || ldy % 8) return false;
    if ((long)N * H * W * (long)(ldx > ldy ? ldx : ldy) >= (1L << 40)) return false;
    return true;
}

#ifdef UMI_FORCE_BN64
static bool use_bn128(int Co) { (void)Co; return false; }
#else
static bool use_bn128(int Co) { return Co % 128 == 0; }
#endif

// Which schedule serves the 3x3 MFMA path: 1 = the round-1 schedule (OPT 0), 2 = branch-free staging + pinned read / MFMA
// interleave (OPT 1), 3 = 2 + staggered start of the two workgroups of a CU (OPT 2, experiment).  Bit-identical outputs.  Process-wide tuning knob (env UMI_CONV3X3_IMPL at load,
// umi_tune_conv3x3_impl at run time for same-process A/B timing).  The rejected round-2 restructurings (LDS-DMA weights,
// persistent workgroups, 8-wave anti-phase) live in tools/experiments/conv_variants/ with the A/B files that retired them.
static int g_impl = [] { const char* e = getenv("UMI_CONV3X3_IMPL"); return e ? atoi(e) : 2; }();
extern "C" int umi_tune_conv3x3_impl(int impl) {
    const int old = g_impl;
    if (impl >= 1 && impl <= 3) g_impl = impl;
    return old;
}

static int pick_th(int Co) { return use_bn128(Co) ? 8 : 16; }

int umi_conv3x3_mfma_stat_rows(int N, int H, int W, int Ci, int Co, int ldx) {
    const int th = pick_th(Co);
    return N * ((W + 31) / 32) * ((H + th - 1) / th);
}

#define UMI_GO(...)                                                                          \
    do {                                                                                     \
        if (g_impl == 1) {                                                                   \
            if (use_bn128(Co)) return launch<8, 128, 0>(__VA_ARGS__);                        \
            return launch<16, 64, 0>(__VA_ARGS__);                                           \
        }                                                                                    \
        if (g_impl == 3) {                                                                   \
            if (use_bn128(Co)) return launch<8, 128, 2>(__VA_ARGS__);                        \
            return launch<16, 64, 2>(__VA_ARGS__);                                           \
        }                                                                                    \
        if (use_bn128(Co)) return launch<8, 128, 1>(__VA_ARGS__);                            \
        return launch<16, 64, 1>(__VA_ARGS__);                                               \
    } while (0)

int umi_conv3x3_mfma(const void* x, int ldx, const void* tx, const void* wp8, void* y, int ldy, float* stat_part,
                     int N, int H, int W, int Ci, int Co, hipStream_t s) {
    UMI_GO(x, ldx, tx, wp8, y, ldy, stat_part, N, H, W, Ci, Co, nullptr, s);
}

// data gradient + stage 1 of the BatchNorm backward of the layer whose activated-output gradient it produces (EPI 2)
int umi_conv3x3_mfma_bnred(const void* dy, int lddy, const void* wp8, void* da, int ldda, const void* ybn, int ldybn,
                           const void* txbn, const float* rstd, float* part, int N, int H, int W, int Ci, int Co,
                           hipStream_t s) {
    const BnRed bn{(const half_t*)ybn, ldybn, (const float4*)txbn, rstd};
    UMI_GO(dy, lddy, nullptr, wp8, da, ldda, part, N, H, W, Ci, Co, &bn, s);
}

// inference: conv + this layer's BatchNorm (running statistics) + ReLU on store (EPI 3)
int umi_conv3x3_mfma_act(const void* x, int ldx, const void* tx, const void* wp8, const void* out_tx, void* y, int ldy, int N,
                         int H, int W, int Ci, int Co, hipStream_t s) {
    UMI_GO(x, ldx, tx, wp8, y, ldy, nullptr, N, H, W, Ci, Co, nullptr, s, out_tx);
}
#undef UMI_GO
