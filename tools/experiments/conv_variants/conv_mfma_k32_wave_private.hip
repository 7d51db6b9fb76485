// conv3x3 (stride 1, pad 1) forward / dgrad on the gfx950 matrix cores, fp16 storage, fp32 accumulate.
//
// TWO FORMS of one kernel template (template parameter K32, picked per shape by UMI_GO at the bottom of this file):
//   * K32 = true  (Co % 128 == 0 and Ci % 32 == 0; 13 of the U-Net's 17 DoubleConv layers): v_mfma_f32_16x16x32_f16, K = the 32
//     staged channels of one tap, weights staged per tap column into two swizzled buffers -- described at `struct Cfg`;
//   * K32 = false (everything else, and UMI_CONV3X3_IMPL=2): the form described next.
// Shared by both: the staging plan (buffer loads, out-of-range offset = zeros), the consumer-side BatchNorm+ReLU transform,
// the halo tile in LDS, the epilogue through an LDS tile and the epilogue reductions (EPI).
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
// Round 3 (profiles/r03_conv_fwd_ab_*.txt; same MFMA order as the round-1 kernel, bit-identical outputs):
//   * staging is branch-free: zero-padding / surplus pieces are stored -- transformed garbage and all -- into the 16 pad
//     bytes of an LDS row that nothing reads, the padding slots themselves are zeroed once per tile; weight-piece validity
//     is folded into a per-lane base offset and the tap step rides in the scalar offset;
//   * the MFMA phase's issue order is pinned (sched_group_barrier): 8 fragment reads up front, then one read behind every
//     MFMA pair, so a read is in flight for >= 4 MFMAs before its use.  hipcc's own order was "read, s_waitcnt lgkmcnt(0),
//     4 MFMAs": one exposed LDS latency per 128 matrix-pipe cycles;
//   * whole tiles take a branch-free, fully unrolled store pass in the epilogue;
//   * measured and NOT kept (tools/experiments/conv_variants/conv_mfma_persist.hip, profiles/r03_conv_fwd_persistent_*): persistent
//     workgroups (2 per CU) that prefetch the next tile's first chunk behind the epilogue's store pass.  With a static tile
//     order the workgroup lifetimes of a launch spread by +-15 % and the launch lasts as long as the slowest; with tiles drawn
//     from per-XCD counters the lifetimes even out and the launch is still 2-10 % slower than one workgroup per tile: issuing
//     the prefetch's 10-12 loads blocks for 1.5-6 k cycles behind the epilogue stores of the CU, which the hardware's own
//     dispatch of a fresh workgroup hides just as well.
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

// SC = 16-channel chunks per staged halo tile ("super-chunk").  SC = 2: the halo tile is staged for 32 input channels at a
// time (64 B of every pixel's 128-B line per load instead of 32 B, every line requested half as often), the weights still per
// 16-channel chunk.  Used by the BN = 64 tile, i.e. the 512 x 512 layers, whose input lines do not survive in the XCD's 4 MB
// L2 from one chunk to the next (64 resident tiles x 78 KB): measured fabric-side fetch 1.87 x the input of 64 -> 64 at
// 512 x 512 and 2.08 x for 128 -> 64 (profiles/r03_conv_fwd_traffic.json) against 1.2 x for the halo alone.
//
// K32 (needs SC = 2): the matrix cores run v_mfma_f32_16x16x32_f16 with K = the 32 staged channels of ONE tap.  The weights of
// a 32-channel super-chunk (9 x BN x 64 B) do not fit next to the halo tile, so they are staged per tap COLUMN dx (3 taps x BN
// rows of 64 B, double-buffered: stage s + 1 is written while stage s is multiplied): three stages per super-chunk, one
// barrier each plus one behind the halo write, 96 MFMAs (1,536 matrix-pipe cycles) and 26 fragment reads per wave and stage.
// The unpadded 64-B weight rows keep the two buffers within 2 workgroups per CU; slot kg of row R sits at kg ^ ((R >> 1) & 3),
// which spreads the 16 rows x 4 k-groups of a fragment read (and the 8 rows of a staging write) over all banks.  Every wave owns 32 output channels x 8 image rows (2 x 16 accumulator tiles of 16 x 16): the
// 6 weight fragments of a stage stay in registers while the 20 pixel fragments (10 halo rows x 2 halves) stream through, each
// used by up to 3 taps x 2 channel tiles.  Same LDS reads per MAC as the 32x32x16 form, half the accumulator traffic per MAC:
// under the power cap the chip holds a higher clock (timing-only ablation +9 % on the Co >= 256 layers,
// profiles/r03_conv_fwd_ab_mfma_shape.txt; MI355X_MICROARCH.md, DVFS give-back item 7).
template <int TH, int BN, int SC = 1, bool K32 = false>
struct Cfg {
    static_assert(!K32 || SC == 2, "K32 stages 32-channel halo tiles");
    static constexpr int WN = BN / 64;
    static constexpr int WM = 4 / WN;
    static_assert(TH / WM == 4, "every wave owns 4 image rows");
    static constexpr int HALO_PIX = (TH + 2) * HALO_W;
    static constexpr int NQ = 2 * SC;                // 8-channel groups per halo row
    static constexpr int HROWB = NQ * 16 + 16;       // LDS bytes per halo pixel: data + 16 B pad (an odd number of 16-B slots)
    static constexpr int HPASS = 256 / NQ;           // halo rows staged per pass of the 256 threads
    static constexpr int HB = HALO_PIX * HROWB;
    static constexpr int WROWB = K32 ? 64 : ROWB;    // LDS bytes per weight row (K32: no pad, the 16-B slots of a row are swizzled)
    static constexpr int WSTG = 3 * BN * 64;         // K32: bytes of one stage's weights; two buffers
#ifdef UMI_K32_WP2
    // K32, second form: TWO halo buffers and ONE weight stage buffer whose rows every wave stages for itself (see the loop)
    static constexpr int WOFF = K32 ? 2 * HB : HB;
    static constexpr int WB = K32 ? WSTG : 9 * BN * ROWB;
#else
    static constexpr int WOFF = HB;
    static constexpr int WB = K32 ? 2 * WSTG : 9 * BN * ROWB;
#endif
    static constexpr int P = TH * 32;
    static constexpr int ERS = BN * 2 + 16;          // epilogue LDS row stride (bytes)
    static constexpr int EB = P * ERS;
    static constexpr int SMEM = (WOFF + WB) > EB ? (WOFF + WB) : EB;
    static constexpr int KPH = (HALO_PIX + HPASS - 1) / HPASS;   // 16-B halo pieces per thread and super-chunk
    static constexpr int NPW = K32 ? 3 * BN * 4 : 9 * BN * 2;   // 16-B pieces of the weights per chunk (K32: per stage)
    static constexpr int KPW = (NPW + 255) / 256;
    static constexpr int TSTEP = 128 / BN;           // taps advanced per weight piece (1 or 2)
    // BN = 64: the last piece (k = 4) is tap 8 for the lower 64 staging rows and does not exist for the upper 64
    static constexpr bool W_LAST_PARTIAL = !K32 && (KPW - 1) * TSTEP + (TSTEP - 1) >= 9;
    static constexpr int WCO = BN / 32, WRW = 4 / WCO;          // K32: waves over channel groups of 32 x row groups of 8
    static_assert(!K32 || TH == 8 * WRW, "K32: every wave owns 8 image rows");
#ifdef UMI_K32_WP2
    static_assert(!K32 || BN == 128, "wave-private weight rows: four waves x 32 channels");
#endif
    static constexpr int PRE = SC == 2 ? 6 : 8;      // fragment reads issued ahead of the MFMA phase's first MFMA (fewer where registers are short)
};

// Epilogue reductions written as one partial row per pixel tile, part[tile][2][Co]:
//   EPI 1: sum / sum of squares of the stored outputs (BatchNorm statistics of THIS conv's output, forward);
//   EPI 2: this launch is the data gradient that produces d(activated output) of a BatchNorm+ReLU layer whose raw output
//          is `bn.y`: sum dz and sum dz*xhat of that layer (dz = stored value * [tx(y) > lo]), i.e. stage 1 of its
//          BatchNorm backward without re-reading the gradient tensor;
//   EPI 3: inference -- this conv's own BatchNorm (running statistics, rows bn.tx) + ReLU applied on store.
struct BnRed { const half_t* y; int ld; const float4* tx; const float* rstd; };

// a pointer the compiler can prove wave-uniform (buffer descriptors built from it need no waterfall loop)
__device__ __forceinline__ void* umi_uniform_ptr(const void* p) {
    const unsigned long a = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (void*)(((unsigned long)hi << 32) | lo);
}

// one work item = (pixel tile, output-channel block)
struct Tile { int n, ty0, tx0, c0, cvalid, pt; };

template <int TH, int BN, int SC, bool K32, bool HAS_TX, int EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(
    const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx, const half_t* __restrict__ wp8,
    half_t* __restrict__ y, int ldy, float* __restrict__ part, int N, int H, int W, int Ci, int Co, int tiles_x,
    int tiles_y, int n_co, int xcd_chunk, BnRed bn) {
    using C = Cfg<TH, BN, SC, K32>;
    constexpr int HROWB = C::HROWB;
    __shared__ __attribute__((aligned(16))) unsigned char smem[C::SMEM];
    // consumer-transform rows of the current / next chunk (16 channels x float4), refilled two chunks ahead so the
    // transform never waits on a global load (measured with in-kernel stamps: 8 dependent tx loads per chunk cost
    // ~1,650 of the ~7,200 cycles of a main-loop iteration)
    __shared__ float4 txbuf[2][16 * SC];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wn = wave % C::WN, wm = wave / C::WN;
#ifdef UMI_STAMP
    UMI_T(t_start);
    unsigned long long seg[5] = {0, 0, 0, 0, 0}, t_epi = 0, t_pro = 0, eps[4] = {0, 0, 0, 0};
    unsigned n_chunks_done = 0;
#endif

    // Staging-thread -> LDS-row mapping.  ds_write_b128 is serviced in groups of 8 consecutive lanes against 32 banks: with
    // the 48-byte rows, 8 lanes covering 4 rows x both halves collide 2-way on half the banks (measured: 25-30 % of the
    // kernel's LDS-array cycles were SQ_LDS_BANK_CONFLICT, profiles/r01_conv_fwd_lds_mfma.json); 8 lanes covering 8
    // consecutive rows of ONE half fall on 32 distinct banks.  Global coalescing is unchanged: a wave still touches the same
    // 32 rows x 32 bytes.
    const int q = (tid >> 3) & 1;                  // which 8-channel half of the 16-channel chunk this thread stages (weights)
    const int srow = ((tid >> 4) << 3) | (tid & 7);   // this thread's row among the 128 staged per pass (weights)
    // halo: NQ 8-channel groups per row, HPASS rows per pass, same 8-consecutive-rows-per-8-lanes shape
    const int hq = SC == 1 ? q : (tid >> 3) & (C::NQ - 1);
    const int hrow = SC == 1 ? srow : ((tid >> 5) << 3) | (tid & 7);

    // weight piece k: row rc = srow + 128k of the [9*BN] rows -> tap = k * TSTEP + srow / BN
    const int tap0 = srow / BN, wcol = srow % BN;
    const int Ci8 = Ci >> 3;
    const int wstep_s = __builtin_amdgcn_readfirstlane(C::TSTEP * Ci8 * Co * 16);       // bytes per weight piece step
    const int wl_base = C::HB + srow * ROWB + q * 16;          // + k * 128 * ROWB
    const bool w_last_ok = tap0 + (C::KPW - 1) * C::TSTEP < 9;
    const int wl_last = w_last_ok ? wl_base + (C::KPW - 1) * 128 * ROWB : C::HB + srow * ROWB + 32;   // or this row's pad bytes

    // fragment base addresses (bytes)
    const int lrow = lane & 31, lhalf = lane >> 5;
    const int b_base = ((wm * 4) * HALO_W + lrow) * HROWB + lhalf * 16;              // + ((nt+dy)*34 + dx)*HROWB + sub*32
    const int a_base = C::HB + (wn * 64 + lrow) * ROWB + lhalf * 16;                 // + (tap*BN + mt*32)*48
    // K32: wave = (channel group wc of 32, row group wr of 8); lane = (row / pixel l16 of a 16 x 16 tile, 8-channel group lg)
    const int wc = wave % C::WCO, wr = wave / C::WCO;
    const int l16 = lane & 15, lg = lane >> 4;
    const int bk_base = ((wr * 8) * HALO_W + l16) * HROWB + lg * 16;                 // + (hr*34 + half*16 + dx) * HROWB
    const int ak_base = C::WOFF + (wc * 32 + l16) * C::WROWB + ((lg ^ ((l16 >> 1) & 3)) * 16);   // + buffer + (dy*BN + ct*16) * WROWB
    // K32 weight staging: piece k of a stage = row R = wk_row + 64k of the [3 dy][BN] rows, 8-channel group wk_q
#ifdef UMI_K32_WP2
    // wave-private weight staging (BN = 128: wave wc reads rows dy * 128 + wc * 32 + 0..31 of the [3 dy][128] rows and nothing else):
    // lane l carries k-group l >> 4 of row (k & 1) * 16 + (l & 15) of its wave's 32 rows, tap row dy = k >> 1 -- 16 consecutive
    // lanes fetch 256 contiguous bytes of the packed weights -- into slot (l >> 4) ^ ((l >> 1) & 3) of that row (8 consecutive
    // lanes: 8 rows x one slot each, all 32 banks)
    const int wkl_base = C::WOFF + ((tid >> 6) * 32 + (tid & 15)) * C::WROWB + ((((tid & 63) >> 4) ^ ((tid >> 1) & 3)) * 16);
#define UMI_WKL(k_) (wkl_base + ((k_) >> 1) * BN * C::WROWB + ((k_) & 1) * 16 * C::WROWB)
#define UMI_PLAN_WROW(tid_) const int wrow_ = ((tid_) >> 6) * 32 + ((tid_) & 15), wq_ = ((tid_) & 63) >> 4
#define UMI_WROW2 16       /* channel rows from piece k to piece k + 1 of one tap row */
#else
    const int wkl_base = C::HB + (((tid >> 5) << 3) | (tid & 7)) * C::WROWB + ((((tid >> 3) & 3) ^ ((tid >> 1) & 3)) * 16);   // + buffer + k * 64 * WROWB
#define UMI_WKL(k_) (wkl_base + (k_) * 64 * C::WROWB)
#define UMI_PLAN_WROW(tid_) const int wrow_ = (((tid_) >> 5) << 3) | ((tid_) & 7), wq_ = ((tid_) >> 3) & 3
#define UMI_WROW2 64
#endif
    const int wdy_s = __builtin_amdgcn_readfirstlane(3 * Ci8 * Co * 16);             // bytes from tap (dy, dx) to (dy + 1, dx)
    const int nchunks = Ci >> 4;
    // (Rotating the chunk order per workgroup to spread the weight reads over L2 channels was measured and is
    //  SLOWER: -12 % on 1024->1024; simultaneous readers of one panel share L2 lines.)

    // ---- staging plan of the tile whose loads are being issued --------------------------------------------------------
    // Loads go through buffer descriptors: an out-of-range voffset returns zeros, so zero padding and partial
    // tiles need no branches around the loads.  The per-chunk offset (and the weights' tap step) rides in the scalar soffset.
    constexpr unsigned OOB = 0x7FFFFFFFu;
    constexpr unsigned WBAD = 0x40000000u;         // + any tap step stays beyond the weight buffer's range
    unsigned hoff[C::KPH];                         // byte offset inside image n, or OOB (zero padding / no piece)
    // LDS byte address a piece is stored at = its slot (hl_a + k * HPASS rows) or, outside the image, the pad bytes of that
    // row (hl_b + ...), picked per chunk from hoff[k]; the last piece, which some threads do not have at all, keeps its own
    int hl_a, hl_b, hl_last;
    unsigned wbase_v, wbase_l;                     // per-lane weight offset (WBAD for channels past Co / a missing last piece)
    __amdgpu_buffer_rsrc_t xrs, wrs;

    // item -> tile.  XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so
    // id -> work item is permuted to give every XCD a contiguous range of (pixel tile, channel block) items: the channel
    // blocks of one pixel tile then share an L2 and the input tile crosses the fabric once instead of once per channel block
    auto decode = [&](int item) {
        int wid = item;
        if (xcd_chunk > 0 && wid < 8 * xcd_chunk) wid = (wid & 7) * xcd_chunk + (wid >> 3);
        Tile t;
        const int cb = wid % n_co;
        t.pt = wid / n_co;
        t.n = t.pt / (tiles_x * tiles_y);
        const int rem = t.pt - t.n * tiles_x * tiles_y;
        t.ty0 = (rem / tiles_x) * TH;
        t.tx0 = (rem % tiles_x) * 32;
        t.c0 = cb * BN;
        t.cvalid = Co - t.c0 < BN ? Co - t.c0 : BN;   // output channels of this tile that exist (Co % 8 == 0, e.g. Co = 16)
        return t;
    };
    // (the thread's staging coordinates are re-derived from a laundered copy of tid: values the chunk loop does not need
    //  must not stay in registers across it for the sake of the next tile's plan)
#define UMI_PLAN(t_)                                                                                                   \
    do {                                                                                                              \
        int tid_ = threadIdx.x;                                                                                       \
        asm volatile("" : "+v"(tid_));                                                                                \
        const int q = (tid_ >> 3) & 1, srow = ((tid_ >> 4) << 3) | (tid_ & 7);                                        \
        const int hq = SC == 1 ? q : (tid_ >> 3) & (C::NQ - 1), hrow = SC == 1 ? srow : ((tid_ >> 5) << 3) | (tid_ & 7); \
        const int tap0 = srow / BN, wcol = srow % BN;                                                                 \
        const unsigned wbase = (unsigned)(((tap0 * (Ci >> 3) + q) * Co + wcol) * 16);                                 \
        const bool w_last_ok = tap0 + (C::KPW - 1) * C::TSTEP < 9;                                                    \
        _Pragma("unroll") for (int k = 0; k < C::KPH; ++k) {                                                          \
            /* halo piece k of this thread: halo pixel hp = hrow + HPASS * k, channel group hq */                     \
            const int hp = hrow + C::HPASS * k;                                                                       \
            const int hy = hp / HALO_W, hx = hp - hy * HALO_W;                                                        \
            const int gy = (t_).ty0 + hy - 1, gx = (t_).tx0 + hx - 1;                                                 \
            const bool inimg = (hp < C::HALO_PIX) && gy >= 0 && gy < H && gx >= 0 && gx < W;                          \
            hoff[k] = inimg ? (unsigned)(gy * W + gx) * (unsigned)(ldx * 2) + hq * 16 : OOB;                          \
            const int hpc = hp < C::HALO_PIX ? hp : C::HALO_PIX - 1;                                                  \
            if (k == C::KPH - 1) hl_last = inimg ? hp * HROWB + hq * 16 : hpc * HROWB + C::NQ * 16;                   \
        }                                                                                                             \
        hl_a = hrow * HROWB + hq * 16;                                                                                \
        hl_b = hrow * HROWB + C::NQ * 16;                                                                             \
        xrs = __builtin_amdgcn_make_buffer_rsrc(umi_uniform_ptr(x + (long)(t_).n * H * W * ldx), 0,                   \
                                                (int)((long)H * W * ldx * 2), 0x00020000);                            \
        wrs = __builtin_amdgcn_make_buffer_rsrc(umi_uniform_ptr(wp8 + (long)(t_).c0 * 8), 0,                          \
                                                (int)((long)9 * Ci * Co * 2 - (long)(t_).c0 * 16), 0x00020000);       \
        const bool wok = wcol < (t_).cvalid;       /* weight rows past Co read as zeros */                            \
        wbase_v = wok ? wbase : WBAD;                                                                                 \
        wbase_l = (wok && w_last_ok) ? wbase : WBAD;                                                                  \
        if constexpr (K32) {       /* rows wk_row (and wk_row + 64 for BN = 128) of the tile's channels, k-group wk_q */ \
            UMI_PLAN_WROW(tid_);                                                                                      \
            const unsigned wb_ = (unsigned)((wq_ * Co + wrow_) * 16);                                                 \
            wbase_v = wrow_ < (t_).cvalid ? wb_ : WBAD;                                                               \
            wbase_l = wrow_ + UMI_WROW2 < (t_).cvalid ? wb_ + UMI_WROW2 * 16 : WBAD;                                  \
        }                                                                                                             \
    } while (0)
    // the zero-padding slots of the halo tile are written here, once per tile, and never by the chunk loop
#define UMI_ZERO_PADDING()                                                                                             \
    do {                                                                                                              \
        int tid_ = threadIdx.x;                                                                                       \
        asm volatile("" : "+v"(tid_));                                                                                \
        const int hq = SC == 1 ? (tid_ >> 3) & 1 : (tid_ >> 3) & (C::NQ - 1);                                         \
        const int hrow = SC == 1 ? ((tid_ >> 4) << 3) | (tid_ & 7) : ((tid_ >> 5) << 3) | (tid_ & 7);                 \
        _Pragma("unroll") for (int k = 0; k < C::KPH; ++k) {                                                          \
            const int hp = hrow + C::HPASS * k;                                                                       \
            if (hp < C::HALO_PIX && hoff[k] == OOB) {                                                                 \
                unsigned z0 = 0;                                                                                      \
                asm volatile("" : "+v"(z0));        /* made here: a zero vector kept in registers across the tile loop is a spill */ \
                *reinterpret_cast<uint4*>(smem + hp * HROWB + hq * 16) = make_uint4(z0, z0, z0, z0);                  \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

    half8 hraw[C::KPH], wraw[C::KPW];
#define UMI_ISSUE_H(sc_)                                                                                          \
    do {                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < C::KPH; ++k)                                                        \
            hraw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(xrs, hoff[k], (sc_) * (32 * SC), 0)); \
    } while (0)
#define UMI_ISSUE_W(c_)                                                                                           \
    do {                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < C::KPW; ++k)                                                        \
            wraw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(                            \
                wrs, (C::W_LAST_PARTIAL && k == C::KPW - 1) ? wbase_l : wbase_v, (c_) * 2 * Co * 16 + k * wstep_s, 0)); \
    } while (0)

    // K32: the weights of stage (super-chunk sc_, tap column dx_): taps (dy, dx_), channels 32 sc_ .. + 31
#define UMI_ISSUE_WK(sc_, dx_)                                                                                    \
    do {                                                                                                          \
        const int ws_ = (((dx_) * Ci8 + 4 * (sc_)) * Co) * 16;                                                    \
        _Pragma("unroll") for (int k = 0; k < C::KPW; ++k)                                                        \
            wraw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(                            \
                wrs, (BN == 128 && (k & 1)) ? wbase_l : wbase_v, ws_ + (BN == 128 ? k >> 1 : k) * wdy_s, 0));     \
    } while (0)

    // the transform rows are carried by the whole of wave 0 (4 lanes per row, same value): a scalar branch, no exec-mask
    // juggling in the loop.  txbuf[c & 1] holds the rows of chunk c, txr those of chunk c + 2 (indices wrap: the surplus loads
    // of the last two chunks read rows that exist).
    const bool tx_wave = __builtin_amdgcn_readfirstlane(tid >> 6) == 0;
    const int txi = lane & (16 * SC - 1);                               // row of the super-chunk this thread carries
    const int txs = (txi & 7) * C::NQ + (txi >> 3);                     // its slot, [j][q]: the channel groups of a chunk side
                                                                        // by side, so lanes of different q read different banks
    const int nsc = nchunks / SC;                                       // super-chunks (the launcher guarantees SC | nchunks)
    float4 txr = make_float4(0.f, 1.f, 0.f, 0.f);
    int txc = 0;                                                        // chunk whose rows txr holds
    // (rows come through a descriptor: a 64-bit row pointer per lane kept across the loop is a register pair too many)
    const __amdgpu_buffer_rsrc_t txrs = __builtin_amdgcn_make_buffer_rsrc(umi_uniform_ptr(tx), 0, HAS_TX ? Ci * 16 : 0, 0x00020000);
#define UMI_TX_ROWS(sc_) __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(txrs, txi * 16, (sc_) * (16 * SC * 16), 0))

    const Tile cur = decode(blockIdx.x);
    UMI_PLAN(cur);
    UMI_ISSUE_H(0);                     // first: the prologue's one global round trip covers the transform rows as well
    half8 wraw1[K32 ? C::KPW : 1];      // K32: the second stage's weights, in flight with the first's through the prologue only
    if constexpr (K32) {
        UMI_ISSUE_WK(0, 0);
#pragma unroll
        for (int k = 0; k < C::KPW; ++k)
            wraw1[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(
                wrs, (BN == 128 && (k & 1)) ? wbase_l : wbase_v, Ci8 * Co * 16 + (BN == 128 ? k >> 1 : k) * wdy_s, 0));
    } else UMI_ISSUE_W(0);
    UMI_ZERO_PADDING();
    if (HAS_TX) {
        if (tx_wave) {
            txbuf[0][txs] = UMI_TX_ROWS(0);
            txbuf[1][txs] = UMI_TX_ROWS(1 % nsc);
            txc = 2 % nsc;
            txr = UMI_TX_ROWS(txc);
        }
        __syncthreads();
    }
    {
        typedef float floatx4 __attribute__((ext_vector_type(4)));
        floatx16 acc[2][4];                 // 32x32x16 form: [channel tile of 32][image row]
        floatx4 acck[2][16];                // K32 (16x16x32) form: [channel tile of 16][image row * 2 + half row]
        if constexpr (K32) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 16; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acck[a][b][r] = 0.f;
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        }
#ifdef UMI_STAMP
        UMI_T(t_loop);
        if (n_chunks_done == 0) t_pro = t_loop - t_start;
#endif
        if constexpr (!K32) {
        for (int ci_ = 0; ci_ < nchunks; ++ci_) {
            const int sci = ci_ / SC, sub = ci_ % SC;              // super-chunk, 16-channel chunk inside it
            const bool first = SC == 1 || sub == 0;                // this iteration stages a halo tile
#ifdef UMI_STAMP
            UMI_T(t0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            UMI_T(t0b);
#endif
            // ---- registers -> (transform) -> LDS ----------------------------------------------------
            if (first) {
                if (HAS_TX) {
                    float4 t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = txbuf[sci & 1][j * C::NQ + hq];
#pragma unroll
                    for (int k = 0; k < C::KPH; ++k) hraw[k] = umi_tx8(hraw[k], t);
                }
#pragma unroll
                for (int k = 0; k < C::KPH; ++k)
                    *reinterpret_cast<half8*>(smem + (k == C::KPH - 1 ? hl_last : (hoff[k] != OOB ? hl_a : hl_b) + k * C::HPASS * HROWB)) = hraw[k];
            }
#pragma unroll
            for (int k = 0; k < C::KPW; ++k)
                *reinterpret_cast<half8*>(smem + ((C::W_LAST_PARTIAL && k == C::KPW - 1) ? wl_last : wl_base + k * 128 * ROWB)) = wraw[k];
#ifdef UMI_STAMP
            UMI_T(t1);
#endif
            __syncthreads();
#ifdef UMI_STAMP
            UMI_T(t2);
#endif
            if (first) {
                if (HAS_TX && tx_wave) {
                    txbuf[sci & 1][txs] = txr;          // every thread is past its reads of this buffer (barrier above)
                    txc = txc + 1 < nsc ? txc + 1 : 0;
                    txr = UMI_TX_ROWS(txc);
                }
                if (sci + 1 < nsc) UMI_ISSUE_H(sci + 1);   // SC = 2: in flight for two MFMA phases
            }
            if (ci_ + 1 < nchunks) UMI_ISSUE_W(ci_ + 1);

            // ---- MFMA phase: 9 taps x (2 x 4) tiles -----------------------------------------------
            // raised wave priority for the MFMA phase: when the two waves of a SIMD compete, the one feeding the matrix pipe wins
            // over the other workgroup's staging VALU work (measured +3..5 %, +15 % on the 64-channel layers)
            __builtin_amdgcn_s_setprio(3);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                half8 bf[6];
#pragma unroll
                for (int rr = 0; rr < 6; ++rr)
                    bf[rr] = *reinterpret_cast<const half8*>(smem + b_base + (rr * HALO_W + dx) * HROWB + sub * 32);
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
            // issue order of the phase's 36 fragment reads and 72 MFMAs: PRE reads up front, then one read behind every MFMA pair
            constexpr int MPR = 2;
            __builtin_amdgcn_sched_group_barrier(0x100, C::PRE, 0);
#pragma unroll
            for (int i_ = 0; i_ < 36 - C::PRE; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPR, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 36 * MPR - MPR * (36 - C::PRE), 0);
            __builtin_amdgcn_s_setprio(0);
#ifdef UMI_STAMP
            UMI_T(t3);
#endif
            __syncthreads();
#ifdef UMI_STAMP
            UMI_T(t4);
            seg[0] += t0b - t0; seg[1] += t1 - t0b; seg[2] += t2 - t1; seg[3] += t3 - t2; seg[4] += t4 - t3;
            ++n_chunks_done;
#endif
        }
        } else {
        // ---- K32: three tap-column stages per 32-channel super-chunk -------------------------------------------------------
        // Invariant at the top of stage st: weight buffer st & 1 holds stage st (written one stage earlier, behind a barrier), the
        // registers hold stage st + 1, which is written into the other buffer at the END of this stage's MFMA phase -- every wave
        // left that buffer at the barrier that ended stage st - 1 -- and stage st + 2 is requested behind the writes.  The halo tile
        // has one buffer: it is written at the top of a super-chunk's first stage, with a barrier of its own in front of the
        // fragment reads.
#pragma unroll
        for (int k = 0; k < C::KPW; ++k) *reinterpret_cast<half8*>(smem + UMI_WKL(k)) = wraw[k];
#pragma unroll
        for (int k = 0; k < C::KPW; ++k) wraw[k] = wraw1[k];
#ifdef UMI_K32_WP2
        // Second form of the loop.  Weights: a wave reads only its own 32 channels' rows, so it stages exactly those: after the six
        // fragment reads of a stage the rows are dead for it and for everybody, and the NEXT stage's rows (registers, requested one
        // stage ago) overwrite them in place at the end of the phase -- one buffer, no barrier for the weights.  The 24 KB this frees
        // hold a second halo tile: super-chunk s + 1's tile is transformed and written into the other buffer at the end of s's
        // SECOND phase (its loads have had a phase and a half), so a super-chunk boundary is one barrier and nothing else -- the
        // transform and the six LDS writes that sat between that barrier and the first MFMA (11 % of the loop,
        // profiles/r03_conv_fwd_ab_staging_ablations.txt) ride behind matrix instructions.
        {
            // second halo buffer: its padding slots are zeroed once, like the first one's
            int tid_ = threadIdx.x;
            asm volatile("" : "+v"(tid_));
            const int hq_ = (tid_ >> 3) & (C::NQ - 1), hrow_ = ((tid_ >> 5) << 3) | (tid_ & 7);
#pragma unroll
            for (int k = 0; k < C::KPH; ++k) {
                const int hp = hrow_ + C::HPASS * k;
                if (hp < C::HALO_PIX && hoff[k] == OOB) {
                    unsigned z0 = 0;
                    asm volatile("" : "+v"(z0));
                    *reinterpret_cast<uint4*>(smem + C::HB + hp * HROWB + hq_ * 16) = make_uint4(z0, z0, z0, z0);
                }
            }
        }
#define UMI_HALO_TX_PAIR(s_, p_)    /* channels 2p, 2p + 1 of the six staged pieces: the rows of two channels live at a time */ \
        do {                                                                                                          \
            if (HAS_TX) {                                                                                             \
                typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));                                          \
                const float4 t0_ = txbuf[(s_) & 1][(2 * (p_)) * C::NQ + hq], t1_ = txbuf[(s_) & 1][(2 * (p_) + 1) * C::NQ + hq]; \
                _Pragma("unroll") for (int k = 0; k < C::KPH; ++k) {                                                  \
                    u32x4_ in_ = __builtin_bit_cast(u32x4_, hraw[k]);                                                 \
                    in_[p_] = umi_tx8_pair(in_[p_], t0_, t1_);                                                        \
                    hraw[k] = __builtin_bit_cast(half8, in_);                                                         \
                }                                                                                                     \
            }                                                                                                         \
        } while (0)
#define UMI_HALO_WRITE(s_)                                                                                            \
        do {                                                                                                          \
            unsigned char* hb_ = smem + ((s_) & 1) * C::HB;                                                           \
            _Pragma("unroll") for (int k = 0; k < C::KPH; ++k)                                                        \
                *reinterpret_cast<half8*>(hb_ + (k == C::KPH - 1 ? hl_last : (hoff[k] != OOB ? hl_a : hl_b) + k * C::HPASS * HROWB)) = hraw[k]; \
        } while (0)
#define UMI_HALO_TO_LDS(s_)                                                                                           \
        do {                                                                                                          \
            UMI_HALO_TX_PAIR(s_, 0); UMI_HALO_TX_PAIR(s_, 1); UMI_HALO_TX_PAIR(s_, 2); UMI_HALO_TX_PAIR(s_, 3);       \
            UMI_HALO_WRITE(s_);                                                                                       \
        } while (0)
        UMI_HALO_TO_LDS(0);
        __syncthreads();
        if (HAS_TX && tx_wave) {
            txbuf[0][txs] = txr;                 // every thread is past its reads of these rows (barrier above)
            txc = txc + 1 < nsc ? txc + 1 : 0;
            txr = UMI_TX_ROWS(txc);
        }
        if (1 < nsc) UMI_ISSUE_H(1);
        for (int sci = 0; sci < nsc; ++sci) {
            const unsigned char* hcur = smem + (sci & 1) * C::HB;
#define UMI_STAGE(DX_)                                                                                                \
            do {                                                                                                      \
                if ((DX_) == 2 && sci + 2 < nsc) UMI_ISSUE_H(sci + 2);     /* hraw is free since the previous phase's tail */ \
                __builtin_amdgcn_s_setprio(3);                                                                        \
                {                                                                                                     \
                    half8 af[3][2];                                                                                   \
                    const unsigned char* ap = smem + ak_base;                                                         \
                    _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                                  \
                        _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                              \
                            af[dy][ct] = *reinterpret_cast<const half8*>(ap + (dy * BN + ct * 16) * C::WROWB);        \
                    const unsigned char* bp = hcur + bk_base + (DX_) * HROWB;                                         \
                    _Pragma("unroll") for (int hr = 0; hr < 10; ++hr)                                                 \
                        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                               \
                            if ((DX_) == 1 && h == 0 && hr >= 2 && (hr & 1) == 0) UMI_HALO_TX_PAIR(sci + 1, (hr - 2) / 2); \
                            const half8 bf = *reinterpret_cast<const half8*>(bp + (hr * HALO_W + h * 16) * HROWB);    \
                            _Pragma("unroll") for (int dy = 0; dy < 3; ++dy) {                                        \
                                const int r = hr - dy;                                                                \
                                if (r >= 0 && r < 8) {                                                                \
                                    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct)                                  \
                                        acck[ct][r * 2 + h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[dy][ct], bf, acck[ct][r * 2 + h], 0, 0, 0); \
                                }                                                                                     \
                            }                                                                                         \
                        }                                                                                             \
                }                                                                                                     \
                {   /* behind the last fragment read: next stage's weight rows in place, the stage after next requested */ \
                    _Pragma("unroll") for (int k = 0; k < C::KPW; ++k) *reinterpret_cast<half8*>(smem + UMI_WKL(k)) = wraw[k]; \
                    const int ndx = (DX_) == 0 ? 2 : (DX_) - 1;                                                       \
                    int nsci = (DX_) == 0 ? sci : sci + 1;                                                            \
                    nsci = nsci < nsc ? nsci : nsc - 1;                                                               \
                    UMI_ISSUE_WK(nsci, ndx);                                                                          \
                }                                                                                                     \
                if ((DX_) == 1) UMI_HALO_WRITE(sci + 1);    /* (past the last super-chunk: stale registers into a buffer nobody reads) */ \
                __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);                                                   \
                UMI_G(2); UMI_G(2); UMI_G(4); UMI_G(4);                                                               \
                if ((DX_) == 1 && HAS_TX) {     /* + the two transform rows of each pair step, four groups ahead of their use */ \
                    UMI_G(6); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); UMI_G(6); UMI_G(6); UMI_G(6);        \
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); UMI_G(6); UMI_G(6); UMI_G(6);                  \
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); UMI_G(6); UMI_G(6); UMI_G(6);                  \
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); UMI_G(6); UMI_G(6);                            \
                } else {                                                                                              \
                    UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); \
                }                                                                                                     \
                _Pragma("unroll") for (int i_ = 0; i_ < C::KPW; ++i_) {                                               \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                \
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                                \
                }                                                                                                     \
                _Pragma("unroll") for (int i_ = 0; i_ < C::KPW; ++i_) {                                               \
                    __builtin_amdgcn_sched_group_barrier(0x008, (12 - C::KPW) / C::KPW, 0);                           \
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                \
                }                                                                                                     \
                if ((DX_) == 1) __builtin_amdgcn_sched_group_barrier(0x200, C::KPH, 0);                               \
                __builtin_amdgcn_s_setprio(0);                                                                        \
            } while (0)
#define UMI_G(n_) __builtin_amdgcn_sched_group_barrier(0x008, n_, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)
            UMI_STAGE(0);
            UMI_STAGE(1);
            UMI_STAGE(2);
#undef UMI_G
            __syncthreads();                     // everybody has left this super-chunk's halo tile; the next one's is complete
            if (HAS_TX && tx_wave) {
                txbuf[(sci + 1) & 1][txs] = txr; // rows of super-chunk sci + 3 replace those of sci + 1 (its tile was transformed in UMI_STAGE(1))
                txc = txc + 1 < nsc ? txc + 1 : 0;
                txr = UMI_TX_ROWS(txc);
            }
        }
#undef UMI_STAGE
#undef UMI_HALO_TO_LDS
#else
        int sci = 0, dx = 0, wcur = 0;                             // wcur: byte offset of the weight buffer this stage reads
        const int nst = 3 * nsc;
        for (int st = 0; st < nst; ++st) {
            const bool first = dx == 0;                            // this stage also stages the super-chunk's halo tile
#ifdef UMI_STAMP
            UMI_T(t0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            UMI_T(t0b);
#endif
            if (first) {
#ifndef UMI_X_NOTX      /* timing-only ablations of the halo tile's staging: no transform / no LDS writes */
                if (HAS_TX) {
                    float4 t[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) t[j] = txbuf[sci & 1][j * C::NQ + hq];
#pragma unroll
                    for (int k = 0; k < C::KPH; ++k) hraw[k] = umi_tx8(hraw[k], t);
                }
#endif
#ifdef UMI_X_NOHW
#pragma unroll
                for (int k = 0; k < C::KPH; ++k) asm volatile("" ::"v"(hraw[k]));
#else
#pragma unroll
                for (int k = 0; k < C::KPH; ++k)
                    *reinterpret_cast<half8*>(smem + (k == C::KPH - 1 ? hl_last : (hoff[k] != OOB ? hl_a : hl_b) + k * C::HPASS * HROWB)) = hraw[k];
#endif
            }
#ifdef UMI_STAMP
            UMI_T(t1);
#endif
            if (first) __syncthreads();
#ifdef UMI_STAMP
            UMI_T(t2);
#endif
            if (first) {
                if (HAS_TX && tx_wave) {
                    txbuf[sci & 1][txs] = txr;          // every thread is past its reads of this buffer (barrier above)
                    txc = txc + 1 < nsc ? txc + 1 : 0;
                    txr = UMI_TX_ROWS(txc);
                }
                if (sci + 1 < nsc) UMI_ISSUE_H(sci + 1);   // in flight for three MFMA phases
            }
            __builtin_amdgcn_s_setprio(3);
            {
                half8 af[3][2];
                const unsigned char* ap = smem + ak_base + wcur;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        af[dy][ct] = *reinterpret_cast<const half8*>(ap + (dy * BN + ct * 16) * C::WROWB);
                const unsigned char* bp = smem + bk_base + dx * HROWB;
#pragma unroll
                for (int hr = 0; hr < 10; ++hr)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const half8 bf = *reinterpret_cast<const half8*>(bp + (hr * HALO_W + h * 16) * HROWB);
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int r = hr - dy;
                            if (r >= 0 && r < 8) {
#pragma unroll
                                for (int ct = 0; ct < 2; ++ct)
                                    acck[ct][r * 2 + h] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[dy][ct], bf, acck[ct][r * 2 + h], 0, 0, 0);
                            }
                        }
                    }
            }
            // The next stage's weights (registers, loaded one stage ago) go to the OTHER buffer -- free since the barrier that ended
            // the previous stage -- and the stage after that is requested, both at the END of this stage's MFMA phase: behind its
            // last fragment read (hipcc orders an LDS write after every earlier LDS read of the same array, so pinning the writes any
            // earlier makes the whole pinned order unsatisfiable and it is dropped) and between its last twelve MFMAs.  At the top
            // of the stage -- where they used to sit, in front of the phase's first fragment reads in the same LDS queue -- they
            // delayed the first MFMA by their own latency: profiles/r03_conv_fwd_ab_staging_ablations.txt (no staging at all +18 %,
            // no writes +10 %, no loads +7 % on the 17 launches; this placement +1.6...3.2 %, outputs bit-identical).  Branch-free:
            // the last stage writes stale registers into a buffer nobody reads again, the last two re-request an existing stage.
            // (UMI_X_*: timing-only ablation builds, tools/build_variant.py NAME conv_mfma.hip -DUMI_X_...; results are wrong by construction)
            {
#ifdef UMI_X_NOWWRITE
#pragma unroll
                for (int k = 0; k < C::KPW; ++k) asm volatile("" ::"v"(wraw[k]));
#else
#pragma unroll
                for (int k = 0; k < C::KPW; ++k)
                    *reinterpret_cast<half8*>(smem + wkl_base + (C::WSTG - wcur) + k * 64 * C::WROWB) = wraw[k];
#endif
#ifndef UMI_X_NOWLOAD
                const int ndx = dx == 0 ? 2 : dx - 1;
                int nsci = dx == 0 ? sci : sci + 1;                                            // stage st + 2
                nsci = nsci < nsc ? nsci : nsc - 1;
                UMI_ISSUE_WK(nsci, ndx);
#endif
            }
            // issue order: the 6 weight fragments and the first 4 pixel fragments up front, then one pixel fragment behind every
            // (halo row, half) group of MFMAs -- four groups (>= 8 MFMAs) ahead of its use
#define UMI_G(n_) __builtin_amdgcn_sched_group_barrier(0x008, n_, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0)
            // (measured against this order, same box: 8 reads up front -0.7 %, 12 up front -7 % (spills), hipcc's own order -1.2 %)
            __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
            UMI_G(2); UMI_G(2); UMI_G(4); UMI_G(4);
            UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6); UMI_G(6);
            // the last twelve MFMAs carry the weight writes and the requests of the stage after next
            // (tail orders measured, profiles/r03_conv_fwd_ab_staging_ablations.txt: a write behind each of the first KPW MFMAs and a
            //  request behind each following group, as here, 4.824 ms; a write per two MFMAs then all requests 4.859; all writes
            //  first 4.942)
#pragma unroll
            for (int i_ = 0; i_ < C::KPW; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
#pragma unroll
            for (int i_ = 0; i_ < C::KPW; ++i_) {
                __builtin_amdgcn_sched_group_barrier(0x008, (12 - C::KPW) / C::KPW, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
#undef UMI_G
            __builtin_amdgcn_s_setprio(0);
#ifdef UMI_STAMP
            UMI_T(t3);
#endif
            __syncthreads();
#ifdef UMI_STAMP
            UMI_T(t4);
            seg[0] += t0b - t0; seg[1] += t1 - t0b; seg[2] += t2 - t1; seg[3] += t3 - t2; seg[4] += t4 - t3;
            ++n_chunks_done;
#endif
            wcur = C::WSTG - wcur;
            if (++dx == 3) { dx = 0; ++sci; }
        }
#endif
        }
#ifdef UMI_STAMP
        UMI_T(t_ep0);
#endif

        // ---- epilogue: acc -> fp16 LDS tile [pixel][BN] ----------------------------------------------
        // EPI 3 (inference): this conv's own BatchNorm (running statistics) + ReLU applied here, on the fp32 accumulators, so the
        // tensor is stored ACTIVATED and its consumers load it as it is (no statistics, no transform on load)
        const int n = cur.n, ty0 = cur.ty0, tx0 = cur.tx0, c0 = cur.c0, cvalid = cur.cvalid, pt = cur.pt;
        if constexpr (K32) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int co = wc * 32 + ct * 16 + lg * 4;          // accumulator rows = 4 consecutive channels per lane
                float4 ot[4];
                if (EPI == 3) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        ot[j] = co + j < cvalid ? bn.tx[c0 + co + j] : make_float4(0.f, 1.f, 0.f, 0.f);
                }
#pragma unroll
                for (int pt_ = 0; pt_ < 16; ++pt_) {
                    const int pix = (wr * 8 + (pt_ >> 1)) * 32 + (pt_ & 1) * 16 + l16;
                    half4 h;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = acck[ct][pt_][j];
                        if (EPI == 3) v = umi_tx(v, ot[j]);
                        h[j] = (half_t)v;
                    }
                    *reinterpret_cast<half4*>(smem + pix * C::ERS + co * 2) = h;
                }
            }
        } else
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
#ifdef UMI_STAMP
        UMI_T(t_e1);
        eps[0] += t_e1 - t_ep0;
#endif

        // coalesced 16-B stores of the tile.  piece i = tid + 256k -> pixel p = i / PPR, 16-B column j = i % PPR; with PPR a
        // power of two p = p0 + k*PSTEP, so rows/cols advance by constants (no divisions in the loop).
        constexpr int PPR = BN / 8;                     // 16-B pieces per pixel row (16 or 8)
        constexpr int PSTEP = 256 / PPR;                // pixels advanced per k (16 or 32)
        constexpr int NK = C::P / PSTEP;                // 16
        const bool full_tile = (ty0 + TH <= H) && (tx0 + 32 <= W);
        // The same pass feeds the epilogue reductions (EPI): a thread keeps one 8-channel column group j for all its pixels, so
        // the per-channel sums accumulate in registers from the values it is storing anyway (no second LDS pass).
        constexpr int SL = PSTEP;                       // pixel slices = threads per column group (16 or 32)
        const int cg = tid % PPR, sl = tid / PPR;
        float s[8], s2[8];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) s[jj] = s2[jj] = 0.f;
        {
            const int j = cg, p0 = sl;
            const unsigned char* sbase = smem + p0 * C::ERS + j * 16;
            const bool col_ok = j * 8 < cvalid;
            float4 t[8];
            float rs_[8];
            if (EPI == 2 && col_ok) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) { t[jj] = bn.tx[c0 + j * 8 + jj]; rs_[jj] = bn.rstd[c0 + j * 8 + jj]; }
            }
            // the reduction of one stored pixel (8 channels)
#define UMI_EPI_REDUCE(v_, yv_)                                                                                        \
            do {                                                                                                      \
                if (EPI == 1) {                                                                                       \
                    const half8 hv = __builtin_bit_cast(half8, v_);                                                   \
                    _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) { float f = (float)hv[jj]; s[jj] += f; s2[jj] = fmaf(f, f, s2[jj]); } \
                } else if (EPI == 2) {                                                                                \
                    const half8 hv = __builtin_bit_cast(half8, v_);                                                   \
                    _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                \
                        const float yy = (float)(yv_)[jj];                                                            \
                        const float dz = umi_tx_pre(yy, t[jj]) > t[jj].w ? (float)hv[jj] : 0.f;                       \
                        s[jj] += dz;                                                                                  \
                        s2[jj] = fmaf(dz, (yy - t[jj].x) * rs_[jj], s2[jj]);                                          \
                    }                                                                                                 \
                }                                                                                                     \
            } while (0)
            // whole tiles (the benchmark's case) take a branch-free, fully unrolled pass through buffer descriptors: pixel
            // p = p0 + k * PSTEP is row (k * PSTEP) >> 5, column p0 + ((k * PSTEP) & 31) of the tile, so the k-dependent part
            // of the address is a scalar offset (no 64-bit address per store, no registers held across the tile loop)
            const bool whole = __builtin_amdgcn_readfirstlane((int)(full_tile && cvalid == BN)) != 0;
            const long tile_pix = ((long)n * H + ty0) * W + tx0;
            half_t* ybase = y + tile_pix * ldy + c0 + j * 8;
            // (1) EPI 2: the BatchNorm layer's raw outputs for this thread's 16 pixels, all loads up front
            half8 yv[NK];
            if (EPI == 2 && whole) {
                const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(umi_uniform_ptr(bn.y + tile_pix * bn.ld + c0), 0, 0x7FFFF000, 0x00020000);
                const unsigned bvo = (unsigned)(p0 * bn.ld + j * 8) * 2u;
#pragma unroll
                for (int k = 0; k < NK; ++k)
                    yv[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(
                        brs, bvo, ((((k * PSTEP) >> 5) * W + ((k * PSTEP) & 31)) * bn.ld) * 2, 0));
            }
#ifdef UMI_STAMP
            UMI_T(t_e2);
            eps[1] += t_e2 - t_e1;
#endif
            // (2) LDS -> global, and the epilogue reduction on the values on their way out
            if (whole) {
                const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(umi_uniform_ptr(y + tile_pix * ldy + c0), 0, 0x7FFFF000, 0x00020000);
                const unsigned yvo = (unsigned)(p0 * ldy + j * 8) * 2u;
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const uint4 v = *reinterpret_cast<const uint4*>(sbase + k * PSTEP * C::ERS);
                    typedef unsigned u32x4_t __attribute__((__vector_size__(4 * sizeof(unsigned))));
                    const u32x4_t vv = __builtin_bit_cast(u32x4_t, v);
                    __builtin_amdgcn_raw_buffer_store_b128(vv, yrs, yvo, ((((k * PSTEP) >> 5) * W + ((k * PSTEP) & 31)) * ldy) * 2, 0);
                    // Observed on gfx950 (ROCm 7.2): a 16-byte buffer store WITH an SGPR soffset followed at once by a VALU write
                    // of its data registers stores the new value of dwords 1.. (hipcc pads this hazard only for stores without
                    // an soffset register).  The data stays live, untouched, for two more wait states.
                    asm volatile("s_nop 1" ::"v"(vv));
                    UMI_EPI_REDUCE(v, yv[k]);
                }
            } else {
                const half_t* yb = EPI == 2 ? bn.y + tile_pix * bn.ld + c0 + j * 8 : nullptr;
#pragma unroll 8
                for (int k = 0; k < NK; ++k) {
                    const int p = p0 + k * PSTEP;
                    const int row = p >> 5, col = p & 31;
                    if (col_ok && (full_tile || (ty0 + row < H && tx0 + col < W))) {
                        const uint4 v = *reinterpret_cast<const uint4*>(sbase + k * PSTEP * C::ERS);
                        *reinterpret_cast<uint4*>(ybase + ((long)row * W + col) * ldy) = v;
                        half8 yv1;
                        if (EPI == 2) yv1 = *reinterpret_cast<const half8*>(yb + ((long)row * W + col) * bn.ld);
                        UMI_EPI_REDUCE(v, yv1);
                    }
                }
            }
#undef UMI_EPI_REDUCE
#ifdef UMI_STAMP
            UMI_T(t_e3);
            eps[2] += t_e3 - t_e2;
#endif
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
        t_epi += t_end - t_ep0;
#endif
    }
#ifdef UMI_STAMP
    if (lane == 0 && blockIdx.x < 512) {
#pragma unroll
        for (int i = 0; i < 5; ++i) umi_stamp_buf[(blockIdx.x * 4 + wave) * 8 + i] = seg[i];
        umi_stamp_buf[(blockIdx.x * 4 + wave) * 8 + 5] = n_chunks_done;
        umi_stamp_buf[(blockIdx.x * 4 + wave) * 8 + 6] = t_pro;
        // (the first three segment sums are replaced by the epilogue's parts when UMI_STAMP_EPI is set)
#ifdef UMI_STAMP_EPI
        for (int i = 0; i < 3; ++i) umi_stamp_buf[(blockIdx.x * 4 + wave) * 8 + i] = eps[i];
#endif
        umi_stamp_buf[(blockIdx.x * 4 + wave) * 8 + 7] = t_epi * (unsigned long long)nchunks / (n_chunks_done ? n_chunks_done : 1);   // per tile
    }
#endif
}

template <int TH, int BN, int SC, bool K32 = false>
int launch(const void* x, int ldx, const void* tx, const void* wp8, void* y, int ldy, float* part, int N, int H, int W,
           int Ci, int Co, const BnRed* bnred, hipStream_t s, const void* out_tx = nullptr) {
    const int tiles_x = (W + 31) / 32, tiles_y = (H + TH - 1) / TH, n_co = (Co + BN - 1) / BN;
    const long nblk = (long)N * tiles_x * tiles_y * n_co;
    if (nblk >= (1L << 31)) return UMI_ERR_UNSUPPORTED;
    dim3 grid((unsigned)nblk), block(256);
    static const bool xcd_off = [] { const char* e = getenv("UMI_CONV_NO_XCD_ORDER"); return e && e[0] == '1'; }();
    const int xcd_chunk = (n_co > 1 && !xcd_off) ? (int)(nblk / 8) : 0;     // ids beyond 8 * chunk (the remainder) keep their order
    const BnRed bn = bnred ? *bnred : BnRed{nullptr, 0, (const float4*)out_tx, nullptr};
#define GO(HT, EP)                                                                                               \
    hipLaunchKernelGGL((conv3x3_mfma_kernel<TH, BN, SC, K32, HT, EP>), grid, block, 0, s, (const half_t*)x, ldx, \
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

#undef UMI_ISSUE_H
#undef UMI_TX_ROWS
#undef UMI_ISSUE_W
#undef UMI_ISSUE_WK
#undef UMI_PLAN
#undef UMI_ZERO_PADDING
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

// Tuning knob kept for same-process A/B experiments (tools/ab_conv.py): the library has ONE conv3x3 MFMA kernel; variants are
// built with tools/build_variant.py and selected per arm.  The round-1 schedule, the round-2 restructurings (LDS-DMA weights,
// persistent DMA-fed workgroups, 8-wave anti-phase) and the round-3 persistent form are in git history /
// tools/experiments/conv_variants/ with the A/B files that retired them (profiles/r02_conv_fwd_ab_*, r03_conv_fwd_ab_*).
static int g_impl = [] { const char* e = getenv("UMI_CONV3X3_IMPL"); return e ? atoi(e) : 1; }();
extern "C" int umi_tune_conv3x3_impl(int impl) {
    const int old = g_impl;
    if (impl >= 1 && impl <= 8) g_impl = impl;
    return old;
}

static int pick_th(int Co) { return use_bn128(Co) ? 8 : 16; }

int umi_conv3x3_mfma_stat_rows(int N, int H, int W, int Ci, int Co, int ldx) {
    const int th = pick_th(Co);
    return N * ((W + 31) / 32) * ((H + th - 1) / th);
}

#define UMI_GO(...)                                                                          \
    do {                                                                                     \
        if (use_bn128(Co) && Ci % 32 == 0 && g_impl != 2) return launch<8, 128, 2, true>(__VA_ARGS__); \
        if (use_bn128(Co)) return launch<8, 128, 1>(__VA_ARGS__);                            \
        if (Ci % 32 == 0 && g_impl != 2) return launch<16, 64, 2>(__VA_ARGS__);              \
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
