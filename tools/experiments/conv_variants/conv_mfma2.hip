// conv3x3 (stride 1, pad 1) forward / dgrad on the gfx950 matrix cores, second generation ("v2").
//
// Same implicit GEMM and the same MFMA order as conv_mfma.hip (v_mfma_f32_32x32x16_f16, A = weights, B = 32 pixels of one
// image row, tap shift = LDS address offset), so outputs are bit-identical to the first kernel.  What changed is how the
// operands reach LDS:
//   * the chunk's 9 taps x 64 output channels x 16 input channels of weights (77 % of the staged bytes of the old BN = 128
//     tile) no longer pass through registers and ds_write_b128: they are copied L2 -> LDS by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`), one 1-KB run [64 co][8 ci] per wave-instruction.  The packed weight layout
//     [tap][Ci/8][Co][8] is already the lane-linear image the DMA needs, and stored as [tap][k-half][co][16 B] the
//     A-fragment ds_read_b128 are conflict-free without padding;
//   * LDS is double-buffered for both operands (2 x 18.4 KB weights + 2 x 19.6 KB halo), so there is ONE barrier per
//     16-channel chunk and the halo transform / ds_write of chunk c+1 is issued between the MFMAs of chunk c instead of
//     in a serial staging phase;
//   * tile = 16 x 32 pixels x 64 output channels per 4-wave workgroup (every wave 4 image rows x 64 channels =
//     acc[2][4]), 76.5 KB of LDS -> two workgroups per CU.
#include "common.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifdef UMI2_STAMP
// diagnostic build only (tools/build_variant.py -DUMI2_STAMP): per-wave cycle sums of the main loop's two waits
__device__ unsigned long long umi2_stamp_buf[2 * 4096 * 8];
#define UMI2_T(var)                                                                       \
    unsigned long long var;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0)
extern "C" int umi_debug_read_stamps2(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(umi2_stamp_buf), sizeof(umi2_stamp_buf));
}
#endif

namespace {

constexpr int TH = 16, BN = 64;
constexpr int HALO_W = 34;
constexpr int HALO_PIX = (TH + 2) * HALO_W;       // 612
constexpr int KPH = 5;                            // 16-B halo pieces per thread and chunk: 5 x 128 pixel slots >= 612
constexpr int HPLANE = KPH * 128 * 16;            // bytes of one k-half plane of the halo tile (640 pixel slots, 28 unused)
constexpr int WBUF = 18 * 1024;                   // [tap][k-half][64 co][8 halfs]
constexpr int HBUF = 2 * HPLANE;                  // [k-half][halo pixel][8 halfs]
// DMA targets first: their LDS address goes through M0 (kept below 64 KB)
constexpr int OFF_DUMMY = 0;                      // 1 KB that the surplus DMA of waves 2 and 3 points at (never read)
constexpr int OFF_W = 1024;
constexpr int OFF_H = OFF_W + 2 * WBUF;
constexpr int OFF_TX = OFF_H + 2 * HBUF;
constexpr int SMEM = OFF_TX + 2 * 16 * 16;
constexpr int P = TH * 32;
constexpr int ERS = BN * 2 + 16;                  // epilogue LDS row stride (bytes)
static_assert(P * ERS <= OFF_TX, "epilogue tile must fit in the staging buffers");
static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");

struct BnRed2 { const half_t* y; int ld; const float4* tx; const float* rstd; };

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: 64 lanes x 16 B from buffer `rs` (byte offset voff per lane + soff) to LDS bytes [lds, lds + 1024).
// Inline asm so that hipcc keeps it out of its vmcnt bookkeeping (cdna_hip_programming.md 5.7 / 'Three .s-level traps' (b)):
// the kernel waits for these copies itself.  M0 carries the LDS address and is restored; s_nop 4 covers a descriptor /
// soffset SGPR written by v_readfirstlane just before.
__device__ __forceinline__ void umi_dma16(unsigned lds, unsigned voff, u32x4 rs, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds), "v"(voff), "s"(rs), "s"(soff)
                 : "memory");
}

__device__ __forceinline__ u32x4 umi_make_rsrc(const void* p, unsigned bytes) {
    const unsigned long a = (unsigned long)p;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

template <bool HAS_TX, int EPI, int STAGE_AT, bool PAIR>
__global__ __launch_bounds__(256, 2) void conv3x3_v2_kernel(
    const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx, const half_t* __restrict__ wp8,
    half_t* __restrict__ y, int ldy, float* __restrict__ part, int N, int H, int W, int Ci, int Co, int tiles_x,
    int tiles_y, int n_co, int xcd_chunk, BnRed2 bn) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    float4* txbuf = reinterpret_cast<float4*>(smem + OFF_TX);      // [2][16], layout [j][q] as in conv_mfma.hip
    const unsigned smem_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef UMI2_STAMP
    UMI2_T(t_k0);
#endif

    int wid = blockIdx.x;
    if (xcd_chunk > 0 && wid < 8 * xcd_chunk) wid = (wid & 7) * xcd_chunk + (wid >> 3);
    const int cb = wid % n_co;
    const int pt = wid / n_co;
    const int n = pt / (tiles_x * tiles_y);
    const int rem = pt - n * tiles_x * tiles_y;
    const int ty0 = (rem / tiles_x) * TH, tx0 = (rem % tiles_x) * 32;
    const int c0 = cb * BN;
    const int cvalid = Co - c0 < BN ? Co - c0 : BN;

    // ---- halo staging plan (same thread -> row mapping as conv_mfma.hip: 8 consecutive lanes store 8 consecutive pixels)
    const int q = (tid >> 3) & 1;
    const int srow = ((tid >> 4) << 3) | (tid & 7);
    constexpr unsigned OOB = 0x7FFFFFFFu;
    unsigned hoff[KPH];
#pragma unroll
    for (int k = 0; k < KPH; ++k) {
        int hp = srow + 128 * k;
        int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        int gy = ty0 + hy - 1, gx = tx0 + hx - 1;
        bool inimg = (hp < HALO_PIX) && gy >= 0 && gy < H && gx >= 0 && gx < W;
        hoff[k] = inimg ? (unsigned)(gy * W + gx) * (unsigned)(ldx * 2) + q * 16 : OOB;
#ifdef UMI2_H_OOB
        hoff[k] = OOB - (unsigned)(inimg ? 16 : 0);          // every load out of range (no memory traffic), the select above still "transforms"
#endif
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(x + (long)n * H * W * ldx), 0, (int)((long)H * W * ldx * 2), 0x00020000);
    const int hl_base = OFF_H + q * HPLANE + srow * 16;              // + buf * HBUF + k * 128 * 16

    // ---- weights: LDS-DMA, run j = tap * 2 + k-half (1 KB = 64 co x 16 B); wave w copies runs w, w+4, w+8, w+12, w+16
    // (waves 2 and 3 have no fifth run: theirs is out of range on the source side and lands in the dummy KB)
    const int Ci8 = Ci >> 3;
    const u32x4 wrs = umi_make_rsrc(wp8 + (long)c0 * 8, (unsigned)((long)9 * Ci * Co * 2 - (long)c0 * 16));
    const unsigned wvoff = lane < cvalid ? (unsigned)lane * 16u : OOB;   // channels past Co: out of range (their rows are never stored)
#ifdef UMI2_NO_DMA
#define UMI_DMA_W(c_, buf_) do {} while (0)
#else
#define UMI_DMA_W(c_, buf_)                                                                                        \
    do {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                            \
            const int j = wave + 4 * i;                                                                            \
            const bool real = (i < 4) || (j < 18);                                                                 \
            umi_dma16(smem_base + (real ? OFF_W + (buf_) * WBUF + j * 1024 : OFF_DUMMY), real ? wvoff : OOB, wrs,   \
                      (unsigned)((((j >> 1) * Ci8 + 2 * (c_) + (j & 1)) * Co) * 16));                              \
        }                                                                                                          \
    } while (0)

#endif

    half8 hraw2[2][KPH];          // [set][piece]; without PAIR only set 0 is used
#define hraw hraw2[hs_]
#ifdef UMI2_NO_HALO
#define UMI_ISSUE_H(c_, set_) do {} while (0)
#define UMI_STAGE_H(buf_, tb_, set_) do {} while (0)
#else
#define UMI_ISSUE_H(c_, set_)                                                                                      \
    do {                                                                                                           \
        constexpr int hs_ = (set_);                                                                                \
        _Pragma("unroll") for (int k = 0; k < KPH; ++k)                                                            \
            hraw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(xrs, hoff[k], (c_) * 32, 0)); \
    } while (0)
    // registers -> (transform with the rows in txbuf[tb_]) -> halo buffer buf_   (branch-free: padding pieces keep their zeros)
#ifdef UMI2_NO_TX
#define UMI2_TXON false
#else
#define UMI2_TXON HAS_TX
#endif
#ifdef UMI2_NO_HWRITE
#define UMI2_HW(p_, v_) asm volatile("" ::"v"(v_))
#else
#define UMI2_HW(p_, v_) *reinterpret_cast<half8*>(p_) = v_
#endif
#define UMI_STAGE_H(buf_, tb_, set_)                                                                               \
    do {                                                                                                           \
        constexpr int hs_ = (set_);                                                                                \
        if (UMI2_TXON) {                                                                                           \
            float4 t[8];                                                                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) t[j] = txbuf[(tb_) * 16 + j * 2 + q];                    \
            _Pragma("unroll") for (int k = 0; k < KPH; ++k) {                                                      \
                const half8 v = umi_tx8(hraw[k], t);                                                               \
                hraw[k] = hoff[k] != OOB ? v : hraw[k];                                                            \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < KPH; ++k)                                                            \
            UMI2_HW(smem + hl_base + (buf_) * HBUF + k * 128 * 16, hraw[k]);                                       \
    } while (0)
#endif

    floatx16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int lrow = lane & 31, lhalf = lane >> 5;
    const int b_base = OFF_H + lhalf * HPLANE + ((wave * 4) * HALO_W + lrow) * 16;     // + buf*HBUF + (rr*34 + dx)*16
    const int a_base = OFF_W + lhalf * 1024 + lrow * 16;                               // + buf*WBUF + tap*2048 + mt*512

    const int nchunks = Ci >> 4;
    const int txl = ((lane & 7) << 1) | ((lane >> 3) & 1);                  // txbuf slot of tx row (lane & 15): [j][q]
    float4 txr = make_float4(0.f, 1.f, 0.f, 0.f);
    int txrow = lane & 15;

    // ---- prologue: chunk 0 into buffer 0, chunk 1's loads in flight ------------------------------
    UMI_ISSUE_H(0, 0);
    UMI_DMA_W(0, 0);
    if (HAS_TX) {
        // every wave writes the same 16 rows (4 lanes per row): no divergent region in the loop below
        txbuf[txl] = tx[lane & 15];
        if (nchunks > 1) txbuf[16 + txl] = tx[16 + (lane & 15)];
        if (nchunks > 2) txr = tx[2 * 16 + (lane & 15)];
        __syncthreads();
    }
    UMI_STAGE_H(0, 0, 0);
    if (nchunks > 1) UMI_ISSUE_H(1, PAIR ? 1 : 0);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(0) : "memory");

    // timing-only ablation switches (tools/build_variant.py; results are wrong by construction, never in the shipped library)
#ifdef UMI2_NO_LDSREAD
    half8 fake;
#pragma unroll
    for (int j = 0; j < 8; ++j) fake[j] = (half_t)(float)(lane + j);
#define UMI2_LDS(p_) fake
#else
#define UMI2_LDS(p_) (*reinterpret_cast<const half8*>(p_))
#endif
#ifdef UMI2_NO_MFMA
#define UMI2_MFMA(acc_, a_, b_) asm volatile("" ::"v"(a_), "v"(b_))
#else
#define UMI2_MFMA(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, acc_, 0, 0, 0)
#endif
#ifdef UMI2_NO_BARRIER
#define UMI2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#elif defined(UMI2_STAMP)
#define UMI2_BARRIER() do { UMI2_T(tb0_); asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); UMI2_T(tb1_); st_bar += tb1_ - tb0_; } while (0)
#else
#define UMI2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
#ifdef UMI2_STAMP
    unsigned long long tseg_;
#define UMI2_SEG_BEGIN() do { UMI2_T(ts_); tseg_ = ts_; } while (0)
#define UMI2_SEG(acc_) do { UMI2_T(ts_); acc_ += ts_ - tseg_; tseg_ = ts_; } while (0)
    unsigned long long st_vm = 0, st_bar = 0, seg0 = 0, seg1 = 0, seg2 = 0, seg3 = 0, seg4 = 0;
#define UMI2_WAIT_VM() do { UMI2_T(tv0_); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); UMI2_T(tv1_); st_vm += tv1_ - tv0_; } while (0)
    UMI2_T(t_loop0);
#else
#define UMI2_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define UMI2_SEG_BEGIN() do {} while (0)
#define UMI2_SEG(acc_) do {} while (0)
#endif

    // One chunk: MFMAs of chunk c_ from buffers b_, with (N1: chunk c_+1 exists) the DMA of its weights at the top and the
    // staging of its halo after tap column STAGE_AT, and (N2: chunk c_+2 exists) the loads of that chunk's halo behind it.
#define UMI_BODY(c_, b_, N1, N2, PAR, NT)                                                                                \
    do {                                                                                                           \
        UMI2_SEG_BEGIN();                                                                                          \
        if (N1) UMI_DMA_W((c_) + 1, (b_) ^ 1);                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                             \
        UMI2_SEG(seg0);                                                                                            \
        _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) {                                                         \
            half8 bf[6];                                                                                           \
            _Pragma("unroll") for (int rr = 0; rr < 6; ++rr)                                                       \
                bf[rr] = UMI2_LDS(smem + b_base + (b_) * HBUF + (rr * HALO_W + dx) * 16);                          \
            _Pragma("unroll") for (int dy = 0; dy < 3; ++dy) {                                                     \
                half8 af[2];                                                                                       \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                   \
                    af[mt] = UMI2_LDS(smem + a_base + (b_) * WBUF + (dy * 3 + dx) * 2048 + mt * 512);              \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                   \
                    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                               \
                        UMI2_MFMA(acc[mt][nt], af[mt], bf[nt + dy]);                                               \
            }                                                                                                      \
            if (dx == 0) UMI2_SEG(seg1); else if (dx == 1) UMI2_SEG(seg2); else UMI2_SEG(seg3);                    \
            if (N1 && dx == STAGE_AT) {                                                                            \
                /* everything this wave has in flight is due now: the halo of chunk c+1, its transform rows, and the DMA of \
                   chunk c+1's weights (which must have landed before the barrier below).  The kernel does not rely on       \
                   hipcc's own vmcnt bookkeeping for the DMA (it cannot see it). */                                         \
                UMI2_WAIT_VM();                                                                                    \
                UMI_STAGE_H((b_) ^ 1, (b_) ^ 1, PAIR ? ((PAR) ^ 1) : 0);                                           \
                /* the next loads depend on these registers, so no pass can hoist them above the wait */           \
                _Pragma("unroll") for (int k = 0; k < KPH; ++k) asm volatile("" : "+v"(hoff[k]));                  \
                if (HAS_TX && NT) {                                                                                \
                    txbuf[(b_) * 16 + txl] = txr;            /* rows of chunk c+2; this buffer's rows (chunk c) were last read during c-1 */ \
                    int cn = (c_) + 3 < nchunks ? (c_) + 3 : nchunks - 1;                                          \
                    asm volatile("" : "+v"(txrow));                                                                \
                    txr = tx[cn * 16 + txrow];                                                                     \
                }                                                                                                  \
                if (!PAIR) { if (N2) UMI_ISSUE_H((c_) + 2, 0); }                                                   \
                else if ((PAR) == 0 && N2) {        /* both register sets are free: two chunks' loads back to back, the second \
                                                       touches the lines the first one just brought into L1 */          \
                    _Pragma("unroll") for (int k = 0; k < KPH; ++k) {        /* piece-major: the two loads of a line are adjacent */ \
                        hraw2[0][k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(xrs, hoff[k], ((c_) + 2) * 32, 0)); \
                        hraw2[1][k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(xrs, hoff[k], ((c_) + 3) * 32, 0)); \
                    }                                                                                              \
                }                                                                                                  \
                UMI2_SEG(seg4);                                                                                    \
            }                                                                                                      \
        }                                                                                                          \
        __builtin_amdgcn_s_setprio(0);                                                                             \
        /* own LDS writes done; every wave's DMA of chunk c+1 was waited for at its staging point (vmcnt(0) there)  */ \
        UMI2_BARRIER();                                                                                            \
    } while (0)

    if (!PAIR) {
        int c = 0;
        for (; c + 2 < nchunks; ++c) {
            const int b = c & 1;
            UMI_BODY(c, b, true, true, 0, true);
        }
        if (c + 1 < nchunks) {
            const int b = c & 1;
            UMI_BODY(c, b, true, false, 0, false);
            ++c;
        }
        {
            const int b = c & 1;
            UMI_BODY(c, b, false, false, 0, false);
        }
    } else {
        // nchunks is even (the launcher guarantees it): chunk c lives in LDS buffers / register set c & 1
        int c = 0;
        for (; c + 2 < nchunks; c += 2) {
            UMI_BODY(c, 0, true, true, 0, true);
            UMI_BODY(c + 1, 1, true, false, 1, true);
        }
        UMI_BODY(c, 0, true, false, 0, false);
        UMI_BODY(c + 1, 1, false, false, 1, false);
    }

#ifdef UMI2_STAMP
    UMI2_T(t_loop1);
#endif
    // ---- epilogue (as conv_mfma.hip): acc -> fp16 LDS tile [pixel][BN] -> 16-B stores + per-channel sums -------
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int pix = (wave * 4 + nt) * 32 + lrow;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = (half_t)acc[mt][nt][g * 4 + j];
                const int co = mt * 32 + g * 8 + lhalf * 4;
                *reinterpret_cast<half4*>(smem + pix * ERS + co * 2) = h;
            }
        }
    __syncthreads();

    constexpr int PPR = BN / 8;                     // 8
    constexpr int PSTEP = 256 / PPR;                // 32
    constexpr int NK = P / PSTEP;                   // 16
    const bool full_tile = (ty0 + TH <= H) && (tx0 + 32 <= W);
    constexpr int SL = PSTEP;
    const int cg = tid % PPR, sl = tid / PPR;
    float s[8], s2[8];
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) s[jj] = s2[jj] = 0.f;
    {
        const int j = cg, p0 = sl;
        half_t* ybase = y + ((long)((long)n * H + ty0) * W + tx0) * ldy + c0 + j * 8;
        const unsigned char* sbase = smem + p0 * ERS + j * 16;
        const bool col_ok = j * 8 < cvalid;
        float4 t[8];
        float rs_[8];
        const half_t* yb = nullptr;
        if (EPI == 2 && col_ok) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) { t[jj] = bn.tx[c0 + j * 8 + jj]; rs_[jj] = bn.rstd[c0 + j * 8 + jj]; }
            yb = bn.y + ((long)((long)n * H + ty0) * W + tx0) * bn.ld + c0 + j * 8;
        }
#pragma unroll 8
        for (int k = 0; k < NK; ++k) {
            const int p = p0 + k * PSTEP;
            const int row = p >> 5, col = p & 31;
            if (col_ok && (full_tile || (ty0 + row < H && tx0 + col < W))) {
                uint4 v = *reinterpret_cast<const uint4*>(sbase + k * PSTEP * ERS);
                *reinterpret_cast<uint4*>(ybase + ((long)row * W + col) * ldy) = v;
                if (EPI == 1) {
                    const half8 hv = __builtin_bit_cast(half8, v);
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) { float f = (float)hv[jj]; s[jj] += f; s2[jj] = fmaf(f, f, s2[jj]); }
                } else if (EPI == 2) {
                    const half8 hv = __builtin_bit_cast(half8, v);
                    const half8 yv = *reinterpret_cast<const half8*>(yb + ((long)row * W + col) * bn.ld);
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const float yy = (float)yv[jj];
                        const float dz = umi_tx_pre(yy, t[jj]) > t[jj].w ? (float)hv[jj] : 0.f;
                        s[jj] += dz;
                        s2[jj] = fmaf(dz, (yy - t[jj].x) * rs_[jj], s2[jj]);
                    }
                }
            }
        }
    }

    if (EPI) {
        __syncthreads();
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
#ifdef UMI2_STAMP
    UMI2_T(t_k1);
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned long long* o = umi2_stamp_buf + (blockIdx.x * 4 + wave) * 8;
        o[0] = t_loop0 - t_k0; o[1] = t_loop1 - t_loop0; o[2] = t_k1 - t_loop1; o[3] = st_vm; o[4] = st_bar; o[5] = nchunks;
        unsigned long long* o2 = umi2_stamp_buf + 4096 * 8 + (blockIdx.x * 4 + wave) * 8;
        o2[0] = seg0; o2[1] = seg1; o2[2] = seg2; o2[3] = seg3; o2[4] = seg4;
    }
#endif
}

#undef UMI_BODY
#undef hraw
#undef UMI2_LDS
#undef UMI2_MFMA
#undef UMI2_BARRIER
#undef UMI2_WAIT_VM
#undef UMI2_SEG
#undef UMI2_SEG_BEGIN
#undef UMI_DMA_W
#undef UMI_ISSUE_H
#undef UMI_STAGE_H

}  // namespace

int umi_conv3x3_mfma2_stat_rows(int N, int H, int W) { return N * ((W + 31) / 32) * ((H + TH - 1) / TH); }

// bn_y: nullptr for the plain forward; else {y, ld, tx, rstd} of the BatchNorm layer whose stage-1 backward sums the
// epilogue emits (conv_mfma.hip's EPI 2).  variant: where in the chunk the next halo tile is staged (tuning knob).
int umi_conv3x3_mfma2(const void* x, int ldx, const void* tx, const void* wp8, void* y, int ldy, float* part, int N, int H,
                      int W, int Ci, int Co, const void* bn_y, int bn_ld, const void* bn_tx, const float* bn_rstd,
                      int variant, hipStream_t s) {
    const int tiles_x = (W + 31) / 32, tiles_y = (H + TH - 1) / TH, n_co = (Co + BN - 1) / BN;
    const long nblk = (long)N * tiles_x * tiles_y * n_co;
    dim3 grid((unsigned)nblk), block(256);
    static const bool xcd_off = [] { const char* e = getenv("UMI_CONV_NO_XCD_ORDER"); return e && e[0] == '1'; }();
    const int xcd_chunk = (n_co > 1 && !xcd_off) ? (int)(nblk / 8) : 0;
    const BnRed2 bn{(const half_t*)bn_y, bn_ld, (const float4*)bn_tx, bn_rstd};
#define GO(HT, EP, SA, PR)                                                                                       \
    hipLaunchKernelGGL((conv3x3_v2_kernel<HT, EP, SA, PR>), grid, block, 0, s, (const half_t*)x, ldx,            \
                       (const float4*)tx, (const half_t*)wp8, (half_t*)y, ldy, part, N, H, W, Ci, Co, tiles_x,   \
                       tiles_y, n_co, xcd_chunk, bn)
#define GO2(HT, EP)                                                                                              \
    do {                                                                                                         \
        if (variant == 2 && (Ci & 31) == 0) GO(HT, EP, 1, true);                                                 \
        else if (variant == 1) GO(HT, EP, 1, false);                                                             \
        else GO(HT, EP, 0, false);                                                                               \
    } while (0)
    if (bn_y) { if (tx) GO2(true, 2); else GO2(false, 2); }
    else if (tx) { if (part) GO2(true, 1); else GO2(true, 0); }
    else    { if (part) GO2(false, 1); else GO2(false, 0); }
#undef GO2
#undef GO
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
