// conv3x3 (stride 1, pad 1) forward / dgrad on the gfx950 matrix cores, fourth generation: persistent 8-wave workgroups
// whose two wave groups run in ANTI-PHASE ("ping-pong").
//
// Stamps of the persistent 4-wave kernel (conv_mfma3.hip) showed every wave alternating ~2.3 k cycles of MFMA issue with
// ~2.4 k cycles of non-MFMA work per 16-channel chunk (weight DMA issue, halo transform + ds_write, load issue, barrier), and
// the two independent workgroups of a CU drifting so that the matrix pipe idles whenever both waves of a SIMD are in their
// non-MFMA part (measured 6.2 k cycles per chunk against 4.6 k of MFMA).  Here ONE workgroup of 8 waves owns the CU and a
// tile of 32 rows x 32 pixels x 64 output channels; waves w and w + 4 share a SIMD and belong to different groups:
//   group A (waves 0-3, rows 0-15):  stage next halo, issue loads + the weight DMA  ->  72 MFMAs        -> barrier
//   group B (waves 4-7, rows 16-31): 72 MFMAs                                      ->  stage, issue loads -> barrier
// so on every SIMD one wave feeds the matrix pipe while the other does everything else; the program order alone creates the
// anti-phase, there is one workgroup barrier per chunk as before.  Both groups read the same weight buffers (one DMA stream,
// issued by group A), each stages its own 18 x 34 halo.  The epilogue is wave-private as in conv_mfma3.hip but through a
// dedicated 4.6-KB LDS region per wave (one image row per pass), so it needs no barrier at all and runs in the wave's
// non-MFMA half: group B right after its last MFMA of the tile, group A after the tile's last barrier, beside group B's first
// MFMAs of the next tile.  LDS: 36.9 KB weights (2 buffers) + 81.9 KB halo (2 groups x 2 buffers) + 36.9 KB epilogue regions.
// MFMA order per output element is unchanged: outputs are bit-identical to the other three kernels.
#include "common.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));


namespace {

constexpr int TH = 16, BN = 64;
constexpr int HALO_W = 34;
constexpr int HALO_PIX = (TH + 2) * HALO_W;       // 612
constexpr int KPH = 5;                            // 16-B halo pieces per thread and chunk: 5 x 128 pixel slots >= 612
constexpr int HPLANE = KPH * 128 * 16;            // bytes of one k-half plane of the halo tile (640 pixel slots, 28 unused)
constexpr int WBUF = 18 * 1024;                   // [tap][k-half][64 co][8 halfs]
constexpr int HBUF = 2 * HPLANE;                  // [k-half][halo pixel][8 halfs]
constexpr int OFF_DUMMY = 0;                      // 1 KB that the surplus DMA of waves 2 and 3 points at (never read)
constexpr int OFF_W = 1024;                       // DMA targets low: their LDS address goes through M0
constexpr int OFF_H = OFF_W + 2 * WBUF;           // [group][buffer][HBUF]
constexpr int ERS = BN * 2 + 16;                  // epilogue LDS row stride (bytes): 64 channels + 16 B pad
constexpr int EREG = 32 * ERS;                    // one wave's epilogue region: one image row of 32 pixels
constexpr int OFF_E = OFF_H + 4 * HBUF;
constexpr int OFF_TX = OFF_E + 8 * EREG;
constexpr int SMEM = OFF_TX + 2 * 16 * 16;
static_assert(SMEM <= 160 * 1024, "one workgroup per CU");
constexpr unsigned OOB = 0x7FFFFFFFu;

struct BnRed4 { const half_t* y; int ld; const float4* tx; const float* rstd; };

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

// Work list: `nblk` items (32 x 32-pixel tile x 64-channel block, channel blocks of one pixel tile adjacent) are cut into 8
// contiguous ranges, one per XCD label (blockIdx & 7: workgroups that share an L2); inside a range the `slots` workgroups of
// that label take items round-robin, so at any moment an XCD works on neighbouring tiles.
template <bool HAS_TX, int EPI>
__global__ __launch_bounds__(512, 2) void conv3x3_pp_kernel(
    const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx, const half_t* __restrict__ wp8,
    half_t* __restrict__ y, int ldy, float* __restrict__ part, int N, int H, int W, int Ci, int Co, int tiles_x,
    int tiles_y, int n_co, int nblk, int per_xcd, int slots, BnRed4 bn) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    float4* txbuf = reinterpret_cast<float4*>(smem + OFF_TX);      // [2][16], layout [j][q] as in conv_mfma.hip
    const unsigned smem_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..7
    const int grp = wave >> 2, wg = wave & 3;                       // wave group (A = 0: stage first; B = 1: MFMA first), wave in group
    const int gtid = tid & 255;                                     // thread index inside the group
    const int lrow = lane & 31, lhalf = lane >> 5;

    const int xl = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int range_end = (xl + 1) * per_xcd < nblk ? (xl + 1) * per_xcd : nblk;
    int item = xl * per_xcd + slot;
    if (item >= range_end) return;                                  // (whole workgroup: uniform)

    // ---- per-thread halo geometry of the group's own 18 x 34 halo, constant over tiles --------------------------------
    const int q = (gtid >> 3) & 1;
    const int srow = ((gtid >> 4) << 3) | (gtid & 7);
    int hyx[KPH];                                                   // (hy << 8) | hx, or -1 for the 28 unused slots
#pragma unroll
    for (int k = 0; k < KPH; ++k) {
        const int hp = srow + 128 * k;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        hyx[k] = hp < HALO_PIX ? ((hy << 8) | hx) : -1;
    }
    const int hl_base = OFF_H + grp * 2 * HBUF + q * HPLANE + srow * 16;      // + buf * HBUF + k * 128 * 16
    const int pix_bytes = ldx * 2;
    const int tiles_img = tiles_x * tiles_y;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((long)N * H * W * ldx * 2), 0x00020000);
    const u32x4 wrs = umi_make_rsrc(wp8, (unsigned)((long)9 * Ci * Co * 2));
    const int Ci8 = Ci >> 3;
    const int nch = Ci >> 4;                                        // >= 2 (launcher)

    // ---- tile plans: `c_` = the tile being computed, `n_` = the one after it (its first two chunks are loaded early) ----
    unsigned c_hoff[KPH], n_hoff[KPH];
    unsigned c_wvoff, n_wvoff;
    int c_n, c_ty0, c_tx0, c_c0, c_cvalid, c_pt;                    // c_ty0: first row of THIS GROUP's 16 rows
    int n_n, n_ty0, n_tx0, n_c0, n_cvalid, n_pt;
#define UMI_PLAN(P, item_)                                                                                         \
    do {                                                                                                           \
        const int it_ = (item_);                                                                                   \
        if (it_ < range_end) {                                                                                     \
            const int cb_ = it_ % n_co;                                                                            \
            P##pt = it_ / n_co;                                                                                    \
            P##n = P##pt / tiles_img;                                                                              \
            const int rem_ = P##pt - P##n * tiles_img;                                                             \
            P##ty0 = (rem_ / tiles_x) * (2 * TH) + grp * TH;                                                       \
            P##tx0 = (rem_ % tiles_x) * 32;                                                                        \
            P##c0 = cb_ * BN;                                                                                      \
            P##cvalid = Co - P##c0 < BN ? Co - P##c0 : BN;                                                         \
            _Pragma("unroll") for (int k = 0; k < KPH; ++k) {                                                      \
                const int gy = P##ty0 + (hyx[k] >> 8) - 1, gx = P##tx0 + (hyx[k] & 255) - 1;                       \
                const bool in_ = hyx[k] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;                            \
                P##hoff[k] = in_ ? (unsigned)((P##n * H + gy) * W + gx) * (unsigned)pix_bytes + q * 16 : OOB;      \
            }                                                                                                      \
            P##wvoff = lane < P##cvalid ? (unsigned)(P##c0 + lane) * 16u : OOB;                                    \
        } else {                        /* no further tile: every load of the plan is out of range (zeros, no traffic) */ \
            P##pt = P##n = P##ty0 = P##tx0 = P##c0 = 0;                                                            \
            P##cvalid = 0;                                                                                         \
            _Pragma("unroll") for (int k = 0; k < KPH; ++k) P##hoff[k] = OOB;                                      \
            P##wvoff = OOB;                                                                                        \
        }                                                                                                          \
    } while (0)

    // weights of chunk ch_ (lane offsets wv_) -> weight buffer buf_, issued by group A only: run j = tap * 2 + k-half (1 KB =
    // 64 co x 16 B), wave wg copies runs wg, wg+4, wg+8, wg+12, wg+16 (waves 2 and 3 have no fifth run: out of range -> dummy KB)
#define UMI_DMA_W(ch_, buf_, wv_)                                                                                  \
    do {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                            \
            const int j = wg + 4 * i;                                                                              \
            const bool real = (i < 4) || (j < 18);                                                                 \
            umi_dma16(smem_base + (real ? OFF_W + (buf_) * WBUF + j * 1024 : OFF_DUMMY), real ? (wv_) : OOB, wrs,   \
                      (unsigned)((((j >> 1) * Ci8 + 2 * (ch_) + (j & 1)) * Co) * 16));                             \
        }                                                                                                          \
    } while (0)

    half8 hraw[KPH];
#define UMI_ISSUE_H(ch_, HO)                                                                                       \
    do {                                                                                                           \
        _Pragma("unroll") for (int k = 0; k < KPH; ++k)                                                            \
            hraw[k] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(xrs, HO[k], (ch_) * 32, 0)); \
    } while (0)
#define UMI_STAGE_H(buf_, tb_, HO)                                                                                 \
    do {                                                                                                           \
        if (HAS_TX) {                                                                                              \
            float4 t[8];                                                                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) t[j] = txbuf[(tb_) * 16 + j * 2 + q];                    \
            _Pragma("unroll") for (int k = 0; k < KPH; ++k) {                                                      \
                const half8 v = umi_tx8(hraw[k], t);                                                               \
                hraw[k] = HO[k] != OOB ? v : hraw[k];                                                              \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int k = 0; k < KPH; ++k)                                                            \
            *reinterpret_cast<half8*>(smem + hl_base + (buf_) * HBUF + k * 128 * 16) = hraw[k];                    \
    } while (0)

    floatx16 acc[2][4];
    const int b_base = OFF_H + grp * 2 * HBUF + lhalf * HPLANE + ((wg * 4) * HALO_W + lrow) * 16;    // + buf*HBUF + (rr*34 + dx)*16
    const int a_base = OFF_W + lhalf * 1024 + lrow * 16;                               // + buf*WBUF + tap*2048 + mt*512
    const int txl = ((lane & 7) << 1) | ((lane >> 3) & 1);                  // txbuf slot of tx row (lane & 15): [j][q]
    int txrow = lane & 15;
    float4 txr = make_float4(0.f, 1.f, 0.f, 0.f);

    // everything staging does for stream chunk s+1 / s+2 while chunk s is computed (in1 / in2: those chunks still belong to the
    // current tile).  Group A also keeps the shared transform rows rolling (txbuf[par] <- rows of chunk s+2).
#define UMI_STAGE_PART(UPDATE_TX)                                                                                  \
    do {                                                                                                           \
        /* everything this wave has in flight is due now (group A: nothing but the halo of chunk s+1) */           \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
        unsigned ho1[KPH], ho2[KPH];                                                                               \
        _Pragma("unroll") for (int k = 0; k < KPH; ++k) { ho1[k] = in1 ? c_hoff[k] : n_hoff[k]; ho2[k] = in2 ? c_hoff[k] : n_hoff[k]; } \
        UMI_STAGE_H(par ^ 1, par ^ 1, ho1);                                                                        \
        /* the next loads depend on these registers, so no pass can hoist them above the wait */                   \
        _Pragma("unroll") for (int k = 0; k < KPH; ++k) asm volatile("" : "+v"(ho2[k]));                           \
        if (HAS_TX && UPDATE_TX) {                                                                                 \
            txbuf[par * 16 + txl] = txr;             /* rows of stream chunk s+2 (this buffer's rows were last read during s-1) */ \
            asm volatile("" : "+v"(txrow));                                                                        \
            txr = tx[w3 * 16 + txrow];                                                                             \
        }                                                                                                          \
        UMI_ISSUE_H(w2, ho2);                                                                                      \
    } while (0)

#define UMI_MFMA_PART()                                                                                            \
    do {                                                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                             \
        _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) {                                                         \
            half8 bf[6];                                                                                           \
            _Pragma("unroll") for (int rr = 0; rr < 6; ++rr)                                                       \
                bf[rr] = *reinterpret_cast<const half8*>(smem + b_base + par * HBUF + (rr * HALO_W + dx) * 16);    \
            _Pragma("unroll") for (int dy = 0; dy < 3; ++dy) {                                                     \
                half8 af[2];                                                                                       \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                   \
                    af[mt] = *reinterpret_cast<const half8*>(smem + a_base + par * WBUF + (dy * 3 + dx) * 2048 + mt * 512); \
                _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                   \
                    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                               \
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt], bf[nt + dy], acc[mt][nt], 0, 0, 0); \
            }                                                                                                      \
        }                                                                                                          \
        __builtin_amdgcn_s_setprio(0);                                                                             \
    } while (0)

    // tile epilogue of this wave: 4 rows x 32 pixels x 64 channels, one image row per pass through the wave's own LDS region
#define UMI_EPILOGUE()                                                                                             \
    do {                                                                                                           \
        unsigned char* ereg = smem + OFF_E + wave * EREG;                                                          \
        const int j = lane & 7, pl = lane >> 3;      /* this lane's 8-channel column group / pixel slot (of 8) */   \
        const bool col_ok = j * 8 < c_cvalid;                                                                      \
        float s[8], s2[8];                                                                                         \
        _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) s[jj] = s2[jj] = 0.f;                                     \
        float4 t[8];                                                                                               \
        float rs_[8];                                                                                              \
        if (EPI == 2 && col_ok) {                                                                                  \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) { t[jj] = bn.tx[c_c0 + j * 8 + jj]; rs_[jj] = bn.rstd[c_c0 + j * 8 + jj]; } \
        }                                                                                                          \
        const int row0 = c_ty0 + wg * 4;                                                                           \
        const long pix0 = ((long)c_n * H + row0) * W + c_tx0;                                                      \
        _Pragma("unroll") for (int sp = 0; sp < 4; ++sp) {                                                         \
            _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                       \
                _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
                    half4 h;                                                                                       \
                    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) h[jj] = (half_t)acc[mt][sp][g * 4 + jj];      \
                    const int co = mt * 32 + g * 8 + lhalf * 4;                                                    \
                    *reinterpret_cast<half4*>(ereg + lrow * ERS + co * 2) = h;                                     \
                }                                                                                                  \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                        \
                const int p = pl + 8 * k;                 /* pixel column of this image row */                     \
                const uint4 v = *reinterpret_cast<const uint4*>(ereg + p * ERS + j * 16);                          \
                if (col_ok && row0 + sp < H && c_tx0 + p < W) {                                                    \
                    const long pix = pix0 + (long)sp * W + p;                                                      \
                    *reinterpret_cast<uint4*>(y + pix * ldy + c_c0 + j * 8) = v;                                   \
                    if (EPI == 1) {                                                                                \
                        const half8 hv = __builtin_bit_cast(half8, v);                                             \
                        _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) { float f = (float)hv[jj]; s[jj] += f; s2[jj] = fmaf(f, f, s2[jj]); } \
                    } else if (EPI == 2) {                                                                         \
                        const half8 hv = __builtin_bit_cast(half8, v);                                             \
                        const half8 yv = *reinterpret_cast<const half8*>(bn.y + pix * bn.ld + c_c0 + j * 8);       \
                        _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                         \
                            const float yy = (float)yv[jj];                                                        \
                            const float dz = umi_tx_pre(yy, t[jj]) > t[jj].w ? (float)hv[jj] : 0.f;                \
                            s[jj] += dz;                                                                           \
                            s2[jj] = fmaf(dz, (yy - t[jj].x) * rs_[jj], s2[jj]);                                   \
                        }                                                                                          \
                    }                                                                                              \
                }                                                                                                  \
            }                                                                                                      \
        }                                                                                                          \
        if (EPI) {                                                                                                 \
            /* sum over the 8 pixel slots (lane bits 3..5); lanes 0..7 then hold the wave's sums of column group j */ \
            _Pragma("unroll") for (int m = 8; m < 64; m <<= 1)                                                     \
                _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) { s[jj] += __shfl_xor(s[jj], m); s2[jj] += __shfl_xor(s2[jj], m); } \
            if (pl == 0 && col_ok) {                                                                               \
                float* prow = part + ((long)c_pt * 8 + wave) * 2 * Co + c_c0 + j * 8;                              \
                *reinterpret_cast<float4*>(prow) = make_float4(s[0], s[1], s[2], s[3]);                            \
                *reinterpret_cast<float4*>(prow + 4) = make_float4(s[4], s[5], s[6], s[7]);                        \
                *reinterpret_cast<float4*>(prow + Co) = make_float4(s2[0], s2[1], s2[2], s2[3]);                   \
                *reinterpret_cast<float4*>(prow + Co + 4) = make_float4(s2[4], s2[5], s2[6], s2[7]);               \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)

    // ---- workgroup prologue: first tile's chunk 0 staged, chunk 1 in flight, plan of the second tile ready -----------
    UMI_PLAN(c_, item);
    UMI_ISSUE_H(0, c_hoff);
    if (grp == 0) UMI_DMA_W(0, 0, c_wvoff);
    if (HAS_TX) {
        txbuf[txl] = tx[lane & 15];
        txbuf[16 + txl] = tx[16 + (lane & 15)];
        txr = tx[(2 % nch) * 16 + (lane & 15)];
    }
    item += slots;
    UMI_PLAN(n_, item);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // halo registers, transform rows and the weight DMA of chunk 0 (first tile only)
    __syncthreads();                                   // txbuf and the DMA'd weights visible to every wave
    UMI_STAGE_H(0, 0, c_hoff);
    _Pragma("unroll") for (int k = 0; k < KPH; ++k) asm volatile("" : "+v"(c_hoff[k]));
    UMI_ISSUE_H(1, c_hoff);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    int par = 0;                                       // LDS buffer / txbuf parity of the chunk being computed
    int w1 = 1, w2 = 2 % nch, w3 = 3 % nch;            // chunk indices (mod nch) of stream chunks s+1, s+2, s+3
    for (;;) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        for (int c = 0; c < nch; ++c) {
            const bool in1 = c + 1 < nch, in2 = c + 2 < nch;        // does stream chunk s+1 / s+2 still belong to this tile?
            if (grp == 0) {
                // group A: non-MFMA half first (its partner waves on the SIMDs are in their MFMA half)
                UMI_STAGE_PART(true);
                UMI_DMA_W(w1, par ^ 1, in1 ? c_wvoff : n_wvoff);
            }
            UMI_MFMA_PART();
            if (grp == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the weight DMA (issued a whole MFMA half ago) has landed
            } else {
                if (c + 1 == nch) UMI_EPILOGUE();                   // accumulators of this tile are complete
                UMI_STAGE_PART(false);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            par ^= 1;
            w1 = w2; w2 = w3; w3 = w3 + 1 == nch ? 0 : w3 + 1;
        }
        if (grp == 0) UMI_EPILOGUE();                               // beside group B's first MFMA half of the next tile

        if (n_cvalid == 0) break;                          // n_ is the dummy plan: this was the workgroup's last tile
        // next tile becomes current; plan of the one after it
#pragma unroll
        for (int k = 0; k < KPH; ++k) c_hoff[k] = n_hoff[k];
        c_wvoff = n_wvoff; c_n = n_n; c_ty0 = n_ty0; c_tx0 = n_tx0; c_c0 = n_c0; c_cvalid = n_cvalid; c_pt = n_pt;
        item += slots;
        UMI_PLAN(n_, item);
    }
}

#undef UMI_PLAN
#undef UMI_DMA_W
#undef UMI_ISSUE_H
#undef UMI_STAGE_H
#undef UMI_STAGE_PART
#undef UMI_MFMA_PART
#undef UMI_EPILOGUE

}  // namespace

int umi_conv3x3_mfma4_stat_rows(int N, int H, int W) { return 8 * N * ((W + 31) / 32) * ((H + 2 * TH - 1) / (2 * TH)); }

// Shapes the 8-wave kernel takes: as conv_mfma3.hip, and at least 17 image rows (otherwise group B has nothing to compute)
bool umi_conv3x3_mfma4_ok(int N, int H, int W, int Ci, int Co, int ldx) {
    return H > TH && Ci >= 32 && (long)N * H * W * ldx * 2 < (1L << 31) && (long)9 * Ci * Co * 2 < (1L << 31) && Co % 8 == 0;
}

int umi_conv3x3_mfma4(const void* x, int ldx, const void* tx, const void* wp8, void* y, int ldy, float* part, int N, int H,
                      int W, int Ci, int Co, const void* bn_y, int bn_ld, const void* bn_tx, const float* bn_rstd,
                      hipStream_t s) {
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 2 * TH - 1) / (2 * TH), n_co = (Co + BN - 1) / BN;
    const long nblk = (long)N * tiles_x * tiles_y * n_co;
    static const int n_cu = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    const int per_xcd = (int)((nblk + 7) / 8);
    int slots = n_cu / 8;                                         // workgroups per XCD label: one per CU
    if (slots > per_xcd) slots = per_xcd;
    const BnRed4 bn{(const half_t*)bn_y, bn_ld, (const float4*)bn_tx, bn_rstd};
    dim3 grid((unsigned)(8 * slots)), block(512);
#define GO(HT, EP)                                                                                               \
    hipLaunchKernelGGL((conv3x3_pp_kernel<HT, EP>), grid, block, 0, s, (const half_t*)x, ldx, (const float4*)tx, \
                       (const half_t*)wp8, (half_t*)y, ldy, part, N, H, W, Ci, Co, tiles_x, tiles_y, n_co,       \
                       (int)nblk, per_xcd, slots, bn)
    if (bn_y) { if (tx) GO(true, 2); else GO(false, 2); }
    else if (tx) { if (part) GO(true, 1); else GO(true, 0); }
    else    { if (part) GO(false, 1); else GO(false, 0); }
#undef GO
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
