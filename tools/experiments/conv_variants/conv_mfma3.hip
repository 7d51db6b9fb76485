// conv3x3 (stride 1, pad 1) forward / dgrad on the gfx950 matrix cores, third generation: PERSISTENT workgroups.
//
// In-kernel cycle stamps of the second kernel (tools/experiments/stamp_conv2.py) showed that a 16 x 32 x 64 tile spends
// 9-22 k cycles in its prologue (first loads and weights arrive, three serialized memory latencies) and 9-13 k in its
// epilogue, next to ~6 k per 16-channel chunk: for the 64- and 128-channel layers (4 / 8 chunks) more than half of a
// workgroup's life is not its main loop, and the two workgroups of a CU did not hide each other's.  Here a workgroup stays
// on its CU and walks a list of tiles; the chunk stream never stops at a tile boundary:
//   * the loads of the NEXT tile's first two chunks (halo to registers, weights by LDS-DMA) are issued during the last two
//     chunks of the current tile, so a tile's prologue disappears (only the workgroup's first tile pays it);
//   * the epilogue needs no workgroup-wide pass: every wave converts its own 4 rows x 32 pixels x 64 channels through a
//     wave-private LDS region (the buffers the last chunk just released) in two 64-pixel passes, stores 16 B per lane and
//     keeps the BatchNorm partial sums of the values it stores; the per-channel sums are reduced across the wave with
//     shuffles and written as ONE PARTIAL ROW PER WAVE (4 rows per tile): no barrier, no cross-wave reduction;
//   * one extra barrier per tile separates the epilogue's LDS regions from the next chunk's staging.
// Main loop, LDS layout, MFMA order and therefore the output bits are those of conv_mfma2.hip / conv_mfma.hip.
#include "common.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifdef UMI3_STAMP
// diagnostic build only (tools/build_variant.py -DUMI3_STAMP): per-wave cycle sums of the persistent kernel's phases
__device__ unsigned long long umi3_stamp_buf[4096 * 8];
#define UMI3_T(var)                                                                       \
    unsigned long long var;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");           \
    __builtin_amdgcn_sched_barrier(0)
extern "C" int umi_debug_read_stamps3(void* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(umi3_stamp_buf), sizeof(umi3_stamp_buf));
}
#else
#define UMI3_T(var) do {} while (0)
#endif

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
constexpr int OFF_H = OFF_W + 2 * WBUF;
constexpr int OFF_TX = OFF_H + 2 * HBUF;
constexpr int SMEM = OFF_TX + 2 * 16 * 16;
constexpr int ERS = BN * 2 + 16;                  // epilogue LDS row stride (bytes): 64 channels + 16 B pad
constexpr int EREG = 64 * ERS;                    // one wave's epilogue region: 64 pixels
static_assert(2 * EREG <= WBUF && 2 * EREG <= HBUF, "two waves' epilogue regions per released buffer");
static_assert(2 * SMEM <= 160 * 1024, "two workgroups per CU");
constexpr unsigned OOB = 0x7FFFFFFFu;

struct BnRed3 { const half_t* y; int ld; const float4* tx; const float* rstd; };

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

// Work list: `nblk` items (pixel tile x 64-channel block, channel blocks of one pixel tile adjacent) are cut into 8
// contiguous ranges, one per XCD label (blockIdx & 7: workgroups that share an L2); inside a range the `slots` workgroups of
// that label take items round-robin, so at any moment an XCD works on neighbouring tiles.
template <bool HAS_TX, int EPI>
__global__ __launch_bounds__(256, 2) void conv3x3_p_kernel(
    const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx, const half_t* __restrict__ wp8,
    half_t* __restrict__ y, int ldy, float* __restrict__ part, int N, int H, int W, int Ci, int Co, int tiles_x,
    int tiles_y, int n_co, int nblk, int per_xcd, int slots, BnRed3 bn) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    float4* txbuf = reinterpret_cast<float4*>(smem + OFF_TX);      // [2][16], layout [j][q] as in conv_mfma.hip
    const unsigned smem_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lhalf = lane >> 5;

    const int xl = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int range_end = (xl + 1) * per_xcd < nblk ? (xl + 1) * per_xcd : nblk;
    int item = xl * per_xcd + slot;
    if (item >= range_end) return;                                  // (whole workgroup: uniform)

    // ---- per-thread halo geometry, constant over tiles ------------------------------------------------------------
    const int q = (tid >> 3) & 1;
    const int srow = ((tid >> 4) << 3) | (tid & 7);
    int hyx[KPH];                                                   // (hy << 8) | hx, or -1 for the 28 unused slots
#pragma unroll
    for (int k = 0; k < KPH; ++k) {
        const int hp = srow + 128 * k;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        hyx[k] = hp < HALO_PIX ? ((hy << 8) | hx) : -1;
    }
    const int hl_base = OFF_H + q * HPLANE + srow * 16;             // + buf * HBUF + k * 128 * 16
    const int pix_bytes = ldx * 2;
    const int tiles_img = tiles_x * tiles_y;

    // one descriptor each for the whole activation tensor and the whole packed weight tensor (the launcher checked that
    // both are < 2 GB); image / channel-block offsets ride in the per-lane offsets of a tile's plan
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((long)N * H * W * ldx * 2), 0x00020000);
    const u32x4 wrs = umi_make_rsrc(wp8, (unsigned)((long)9 * Ci * Co * 2));
    const int Ci8 = Ci >> 3;
    const int nch = Ci >> 4;                                        // >= 2 (launcher)

    // ---- tile plans: `c_` = the tile being computed, `n_` = the one after it (its first two chunks are loaded early) ----
    unsigned c_hoff[KPH], n_hoff[KPH];
    unsigned c_wvoff, n_wvoff;
    int c_n, c_ty0, c_tx0, c_c0, c_cvalid, c_pt;
    int n_n, n_ty0, n_tx0, n_c0, n_cvalid, n_pt;
#define UMI_PLAN(P, item_)                                                                                         \
    do {                                                                                                           \
        const int it_ = (item_);                                                                                   \
        if (it_ < range_end) {                                                                                     \
            const int cb_ = it_ % n_co;                                                                            \
            P##pt = it_ / n_co;                                                                                    \
            P##n = P##pt / tiles_img;                                                                              \
            const int rem_ = P##pt - P##n * tiles_img;                                                             \
            P##ty0 = (rem_ / tiles_x) * TH;                                                                        \
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

    // weights of chunk ch_ of the plan with lane offsets wv_ -> weight buffer buf_: run j = tap * 2 + k-half (1 KB = 64 co x
    // 16 B), wave w copies runs w, w+4, w+8, w+12, w+16 (waves 2 and 3 have no fifth run: out of range -> the dummy KB)
#define UMI_DMA_W(ch_, buf_, wv_)                                                                                  \
    do {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                            \
            const int j = wave + 4 * i;                                                                            \
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
    // registers -> (transform with the rows in txbuf[tb_]) -> halo buffer buf_; HO = the offsets the loads were issued with
    // (padding pieces keep their zeros)
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
    const int b_base = OFF_H + lhalf * HPLANE + ((wave * 4) * HALO_W + lrow) * 16;     // + buf*HBUF + (rr*34 + dx)*16
    const int a_base = OFF_W + lhalf * 1024 + lrow * 16;                               // + buf*WBUF + tap*2048 + mt*512
    const int txl = ((lane & 7) << 1) | ((lane >> 3) & 1);                  // txbuf slot of tx row (lane & 15): [j][q]
    int txrow = lane & 15;
    float4 txr = make_float4(0.f, 1.f, 0.f, 0.f);

    // ---- workgroup prologue: first tile's chunk 0 staged, chunk 1 in flight, plan of the second tile ready -----------
    UMI_PLAN(c_, item);
    UMI_ISSUE_H(0, c_hoff);
    UMI_DMA_W(0, 0, c_wvoff);
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
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (the DMA of chunk 0 was waited for by __syncthreads)

#ifdef UMI3_STAMP
    unsigned long long st_loop = 0, st_epi = 0, st_plan = 0, st_tiles = 0, st_bar = 0, st_vm = 0;
    UMI3_T(t_start);
#endif
    int par = 0;                                       // LDS buffer / txbuf parity of the chunk being computed
    int w1 = 1, w2 = 2 % nch, w3 = 3 % nch;            // chunk indices (mod nch) of stream chunks s+1, s+2, s+3
    for (;;) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

        UMI3_T(t_l0);
        for (int c = 0; c < nch; ++c) {
            const bool in1 = c + 1 < nch, in2 = c + 2 < nch;        // does stream chunk s+1 / s+2 still belong to this tile?
            UMI_DMA_W(w1, par ^ 1, in1 ? c_wvoff : n_wvoff);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                half8 bf[6];
#pragma unroll
                for (int rr = 0; rr < 6; ++rr)
                    bf[rr] = *reinterpret_cast<const half8*>(smem + b_base + par * HBUF + (rr * HALO_W + dx) * 16);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    half8 af[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        af[mt] = *reinterpret_cast<const half8*>(smem + a_base + par * WBUF + (dy * 3 + dx) * 2048 + mt * 512);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt], bf[nt + dy], acc[mt][nt], 0, 0, 0);
                }
                if (dx == 1) {
                    // everything this wave has in flight is due now: the halo of stream chunk s+1, its transform rows and the
                    // DMA of its weights (must have landed before the barrier below; hipcc cannot see that copy)
                    UMI3_T(tv0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    UMI3_T(tv1);
#ifdef UMI3_STAMP
                    st_vm += tv1 - tv0;
#endif
                    unsigned ho1[KPH], ho2[KPH];
#pragma unroll
                    for (int k = 0; k < KPH; ++k) { ho1[k] = in1 ? c_hoff[k] : n_hoff[k]; ho2[k] = in2 ? c_hoff[k] : n_hoff[k]; }
                    UMI_STAGE_H(par ^ 1, par ^ 1, ho1);
                    // the next loads depend on these registers, so no pass can hoist them above the wait
#pragma unroll
                    for (int k = 0; k < KPH; ++k) asm volatile("" : "+v"(ho2[k]));
                    if (HAS_TX) {
                        txbuf[par * 16 + txl] = txr;             // rows of stream chunk s+2 (this buffer's rows were last read during s-1)
                        asm volatile("" : "+v"(txrow));
                        txr = tx[w3 * 16 + txrow];
                    }
                    UMI_ISSUE_H(w2, ho2);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            UMI3_T(tb0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            UMI3_T(tb1);
#ifdef UMI3_STAMP
            st_bar += tb1 - tb0;
#endif
            par ^= 1;
            w1 = w2; w2 = w3; w3 = w3 + 1 == nch ? 0 : w3 + 1;
        }

        UMI3_T(t_l1);
        // ---- tile epilogue: wave-private, through the LDS buffers the last chunk released (parity par ^ 1) -------------
        {
            // the other workgroup of the CU is in its chunk loop at priority 1: without a raise this wave's VALU / LDS issue
            // starves (stamps: 12.5 k cycles per epilogue at priority 0)
            __builtin_amdgcn_s_setprio(2);
            const int pf = par ^ 1;
            unsigned char* ereg = smem + (wave < 2 ? OFF_W + pf * WBUF + wave * EREG : OFF_H + pf * HBUF + (wave - 2) * EREG);
            const int j = lane & 7, pl = lane >> 3;               // this lane's 8-channel column group / pixel slot (of 8)
            const bool col_ok = j * 8 < c_cvalid;
            float s[8], s2[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) s[jj] = s2[jj] = 0.f;
            float4 t[8];
            float rs_[8];
            if (EPI == 2 && col_ok) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) { t[jj] = bn.tx[c_c0 + j * 8 + jj]; rs_[jj] = bn.rstd[c_c0 + j * 8 + jj]; }
            }
            const int row0 = c_ty0 + wave * 4;
            const long pix0 = ((long)c_n * H + row0) * W + c_tx0;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh) {
                        const int nt = sp * 2 + nh;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            half4 h;
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) h[jj] = (half_t)acc[mt][nt][g * 4 + jj];
                            const int co = mt * 32 + g * 8 + lhalf * 4;
                            *reinterpret_cast<half4*>(ereg + (nh * 32 + lrow) * ERS + co * 2) = h;
                        }
                    }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int p = pl + 8 * k;                     // pixel of this pass: row p >> 5, column p & 31
                    const int row = row0 + sp * 2 + (p >> 5), col = c_tx0 + (p & 31);
                    const uint4 v = *reinterpret_cast<const uint4*>(ereg + p * ERS + j * 16);
                    if (col_ok && row < H && col < W) {
                        const long pix = pix0 + (long)(sp * 2 + (p >> 5)) * W + (p & 31);
                        *reinterpret_cast<uint4*>(y + pix * ldy + c_c0 + j * 8) = v;
                        if (EPI == 1) {
                            const half8 hv = __builtin_bit_cast(half8, v);
#pragma unroll
                            for (int jj = 0; jj < 8; ++jj) { float f = (float)hv[jj]; s[jj] += f; s2[jj] = fmaf(f, f, s2[jj]); }
                        } else if (EPI == 2) {
                            const half8 hv = __builtin_bit_cast(half8, v);
                            const half8 yv = *reinterpret_cast<const half8*>(bn.y + pix * bn.ld + c_c0 + j * 8);
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
                // sum over the 8 pixel slots (lane bits 3..5); lanes 0..7 then hold the wave's sums of column group j
#pragma unroll
                for (int m = 8; m < 64; m <<= 1)
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) { s[jj] += __shfl_xor(s[jj], m); s2[jj] += __shfl_xor(s2[jj], m); }
                if (pl == 0 && col_ok) {
                    float* prow = part + ((long)c_pt * 4 + wave) * 2 * Co + c_c0 + j * 8;
                    *reinterpret_cast<float4*>(prow) = make_float4(s[0], s[1], s[2], s[3]);
                    *reinterpret_cast<float4*>(prow + 4) = make_float4(s[4], s[5], s[6], s[7]);
                    *reinterpret_cast<float4*>(prow + Co) = make_float4(s2[0], s2[1], s2[2], s2[3]);
                    *reinterpret_cast<float4*>(prow + Co + 4) = make_float4(s2[4], s2[5], s2[6], s2[7]);
                }
            }
        }

        __builtin_amdgcn_s_setprio(0);
        UMI3_T(t_e1);
#ifdef UMI3_STAMP
        st_loop += t_l1 - t_l0; st_epi += t_e1 - t_l1; st_tiles += 1;
#endif
        if (n_cvalid == 0) break;                          // n_ is the dummy plan: this was the workgroup's last tile
        // next tile becomes current; plan of the one after it
#pragma unroll
        for (int k = 0; k < KPH; ++k) c_hoff[k] = n_hoff[k];
        c_wvoff = n_wvoff; c_n = n_n; c_ty0 = n_ty0; c_tx0 = n_tx0; c_c0 = n_c0; c_cvalid = n_cvalid; c_pt = n_pt;
        item += slots;
        UMI_PLAN(n_, item);
        // the epilogue regions are staging buffers again from the next chunk on
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        UMI3_T(t_p1);
#ifdef UMI3_STAMP
        st_plan += t_p1 - t_e1;
#endif
    }
#ifdef UMI3_STAMP
    UMI3_T(t_end);
    if (lane == 0 && blockIdx.x < 1024) {
        unsigned long long* o = umi3_stamp_buf + (blockIdx.x * 4 + wave) * 8;
        o[0] = t_end - t_start; o[1] = st_loop; o[2] = st_epi; o[3] = st_plan; o[4] = st_tiles; o[5] = st_bar; o[6] = st_vm; o[7] = nch;
    }
#endif
}

#undef UMI_PLAN
#undef UMI_DMA_W
#undef UMI_ISSUE_H
#undef UMI_STAGE_H

}  // namespace

int umi_conv3x3_mfma3_stat_rows(int N, int H, int W) { return 4 * N * ((W + 31) / 32) * ((H + TH - 1) / TH); }

// Shapes the persistent kernel takes (the others go to conv_mfma.hip): at least two 16-channel chunks, both tensors
// addressable through one 2-GB buffer descriptor, 16-B aligned partial rows.
bool umi_conv3x3_mfma3_ok(int N, int H, int W, int Ci, int Co, int ldx) {
    return Ci >= 32 && (long)N * H * W * ldx * 2 < (1L << 31) && (long)9 * Ci * Co * 2 < (1L << 31) && Co % 8 == 0;
}

int umi_conv3x3_mfma3(const void* x, int ldx, const void* tx, const void* wp8, void* y, int ldy, float* part, int N, int H,
                      int W, int Ci, int Co, const void* bn_y, int bn_ld, const void* bn_tx, const float* bn_rstd,
                      hipStream_t s) {
    const int tiles_x = (W + 31) / 32, tiles_y = (H + TH - 1) / TH, n_co = (Co + BN - 1) / BN;
    const long nblk = (long)N * tiles_x * tiles_y * n_co;
    static const int n_cu = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev);
        return v > 0 ? v : 256;
    }();
    const int per_xcd = (int)((nblk + 7) / 8);
    int slots = (2 * n_cu) / 8;                                   // workgroups per XCD label: two per CU
    if (slots > per_xcd) slots = per_xcd;
    const BnRed3 bn{(const half_t*)bn_y, bn_ld, (const float4*)bn_tx, bn_rstd};
    dim3 grid((unsigned)(8 * slots)), block(256);
#define GO(HT, EP)                                                                                               \
    hipLaunchKernelGGL((conv3x3_p_kernel<HT, EP>), grid, block, 0, s, (const half_t*)x, ldx, (const float4*)tx,  \
                       (const half_t*)wp8, (half_t*)y, ldy, part, N, H, W, Ci, Co, tiles_x, tiles_y, n_co,       \
                       (int)nblk, per_xcd, slots, bn)
    if (bn_y) { if (tx) GO(true, 2); else GO(false, 2); }
    else if (tx) { if (part) GO(true, 1); else GO(true, 0); }
    else    { if (part) GO(false, 1); else GO(false, 0); }
#undef GO
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}
