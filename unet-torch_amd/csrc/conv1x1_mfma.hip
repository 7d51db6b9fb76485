// Pointwise convolutions on the gfx950 matrix cores (fp16 storage, fp32 accumulate):
//   * plain 1x1 conv                                   y[p][co]        = sum_k  tx(x[p][k]) * w[k][co] (+ bias)
//   * ConvTranspose2d(k=2,s=2) forward  (OUT_UPS)      y[2p+t][co]     = sum_k  tx(x[p][k]) * w_t[k][co] + bias
//       = a 1x1 conv with 4*Cout outputs whose epilogue scatters channel block t to sub-pixel t
//   * its data gradient                 (GATHER)       dx[p][ci]       = sum_t sum_k dy[2p+t][k] * w_t[k][ci]
//       = a 1x1 conv whose K axis is the space-to-depth gather of the 4 sub-pixels
//   * any other R x S conv with stride / padding (GATHER): K axis = R*S taps x Ci, source pixel of tap (ty,tx) =
//       (stride*y + ty - pad, stride*x + tx - pad), zero outside the image (TransUNet's stride-2 3x3 and 1x1 convs,
//       resnet_skip.py:52-56,60) and the data gradient of a strided conv (UMI_CONV_DGRAD_STRIDED, "fractional" gather:
//       source = ((y + pad - ty) / stride, (x + pad - tx) / stride) where that is integral)
// (reference Model.py:56-57,67 and the autograd of it; TransUNet's 1x1 convs / patch embedding reuse it).
//
// D[co][pixel] += W[co][k] * X[k][pixel] with v_mfma_f32_32x32x16_f16 (A = weights, B = pixels), same
// structure as conv_mfma.hip with one tap: 256 threads = 4 waves, 2 workgroups per CU, tile = 256 linear
// pixels x BN output channels, K staged in 64-channel chunks through registers (issue-early / write-late)
// into LDS rows of 144 B (128 B data + 16 B pad = 9 slots, odd -> conflict-free ds_read_b128), consumer-side
// BN+ReLU transform applied on the way.  Epilogue through an LDS tile for 16-B coalesced (scattered) stores.
#include "common.h"
#include <stdlib.h>
#include <string.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// epilogue fusion request of umi_linear_fused (api.hip); mode 0 = none
struct UmiLinearEpi { int mode; float p; unsigned seed; const unsigned* seed_dev; void* mask; const void* aux; int ldaux; void* y2; int ldy2;
                      const void* bn_tx; const float* bn_rstd; float* bn_part; };   // mode 3: aux = the BatchNorm layer's raw output

namespace {

constexpr int OUT_UPS_TAPS_MAX = 49;     // most taps a packed weight tensor of this kernel has (R*S <= 49)

// pixels per workgroup tile: template parameter P (256, or 128 when the grid would not fill the chip)
constexpr int CK = 64;        // K chunk
constexpr int ROWB = 160;     // LDS row bytes: 64 halfs + 32 B pad = 10 slots of 16 B, the stride at which the 16x16x32 fragment
                              // reads (lane = row l & 15, k group l >> 4) fall on distinct banks in every ds_read_b128 lane group

struct Geo {                  // geometry of the (optionally strided) source / destination tensors
    int h, w;                 // the pixel grid the GEMM's M index runs over (N*h*w pixels)
    int Hs, Ws, soy, sox;     // source tensor dims (+ window offset) ; GATHER: source pixel = (s*y+ty-pad+soy, s*x+tx-pad+sox)
    int Hd, Wd, doy, dox;     // destination dims (+ offset)          ; for OUT_UPS dest pixel = (2y+dy+doy, 2x+dx+dox)
    int S, stride, pad, frac; // GATHER: taps per row, stride, padding; frac = data gradient of a strided conv
    int accum;                // UMI_CONV_ACCUMULATE: y += result (fp16 add of the stored and the new value)
    // Epilogue fusions of the ViT block's linears (plain dense mode only; reference vit_seg_modeling.py:113-119,177-187), the
    // arithmetic of elementwise_tu_f16.hip's dropout8_fused_kernel on the fp16 values the unfused GEMM would have stored --
    // same random stream (element index, seed), same mask bytes, same roundings, so fused == unfused bit for bit:
    //   epi 1 (fc1):            y = x W + b (kept for the GELU backward),  y2 = dropout(gelu(y)),  mask
    //   epi 2 (fc2 / attn out): y = dropout(x W + b) + aux,                                          mask
    int epi;
    float drop_p;
    unsigned seed;
    const unsigned* seed_dev;
    unsigned char* mask;
    const half_t* aux; int ldaux;
    half_t* y2; int ldy2;
    //   epi 3 (data gradients: ConvTranspose2d's, strided convs'): also stage 1 of the BatchNorm(+ReLU) backward of the layer whose
    //          activated output this gradient belongs to (aux = that layer's raw output, bn_tx / bn_rstd its transform rows and
    //          1/std): bn_part[pixel tile][2][columns] <- sums of dz and dz * xhat over the tile, dz = stored value * [tx(y) > lo]
    const float4* bn_tx; const float* bn_rstd; float* bn_part;
};

__device__ __forceinline__ float c1_gelu(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ unsigned c1_hash32(unsigned a, unsigned b) {            // == elementwise_tu_f16.hip hash32
    unsigned x = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

// Workgroups per CU: the 128 x 64 tiles keep 32 accumulators and fit 128 VGPRs, so FOUR of their workgroups share a CU (4 x 33 KB
// of LDS): the counters of the 4,704-token linears (tools/experiments/pmc_gemm.py) show waves parked at s_waitcnt / barriers
// for half of their cycles with two resident workgroups, MFMA pipes 21 % and LDS 33 % busy -- a synchronisation-bound loop
// that more resident workgroups hide, not a bandwidth-bound one.
#ifndef UMI_C1_OCC128
#define UMI_C1_OCC128 4
#endif
// (timing-only ablation builds of tools/ab_convT.py -- profiles/r03_convT_trio_per_level.txt -- never compiled into the shipped
//  library: UMI_X_NOMFMA drops the matrix instructions, UMI_X_NOSTORE the epilogue's global stores)
#ifdef UMI_X_NOMFMA
#define UMI_X_MMA(c_, a_, b_) do { if ((ks | ct | i) == 0) c_[0] += (float)a_[0] * (float)b_[0]; } while (0)
#else
#define UMI_X_MMA(c_, a_, b_) c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0)
#endif
template <int P, int BN, bool GATHER, bool OUT_UPS, bool HAS_TX>
__global__ __launch_bounds__(256, (P == 128 && BN == 64) ? UMI_C1_OCC128 : 2) void conv1x1_mfma_kernel(
    const half_t* __restrict__ x, int ldx, const float4* __restrict__ tx, const half_t* __restrict__ wp8,
    const float* __restrict__ bias, half_t* __restrict__ y, int ldy, long M, int Kc /*channels per tap of the source*/,
    int Nc /*channels per tap of the destination*/, int n_co, int ntaps, Geo geo, long ntiles) {
    constexpr int WN = BN / 64, WM = 4 / WN, NT = P / (32 * WM);
    constexpr int XB = P * ROWB, WB = BN * ROWB;
    constexpr int ERS = BN * 2 + 16, EB = P * ERS;
    constexpr int SMEM = (XB + WB) > EB ? (XB + WB) : EB;
    constexpr int KPX = P * 8 / 256;          // 8
    constexpr int KPW = BN * 8 / 256;         // 4 or 2
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
    __shared__ float4 txbuf[2][CK];          // transform rows of the current / next K chunk, refilled two chunks ahead
    __shared__ int2 pixinfo[P];              // (image, y << 16 | x) of the tile's pixels on the GEMM grid: computed once in the
                                             // staging plan, so the epilogue's 16 pieces per thread need no divisions

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    // Workgroup b runs on XCD b % 8 (round-robin dispatch), and every XCD has its own L2.  Tiles are numbered so that each XCD
    // gets one contiguous run of them -- with the output-channel tile running fastest, the n_co tiles that share a pixel-row tile
    // (the same 256 or 128 rows of x) run on ONE XCD and x crosses the fabric once, not up to 8 times (counters of the
    // 4,704 x 3,072 -> 768 linear before this: 248 MB of L2 misses for 34 MB of operands, 5.3 TB/s of fabric traffic
    // at the kernel's 47 us -- that, not LDS or MFMA issue, was its bound).
    const int per_xcd = (int)gridDim.x >> 3;                  // the grid is a multiple of 8
    const long tile = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int cb = (int)(tile % n_co);
    const long m0 = (tile / n_co) * P;
    const int c0 = cb * BN;
    const int sub = tid & 7;

    // ---- staging plan --------------------------------------------------------------------------------
    // dense: a plain pointwise conv whose source and destination are whole tensors (token linears, bottleneck 1x1 convs): pixel m
    // is row m of both, none of the index arithmetic below (64-bit divisions: ~2 us of a 17-us launch) is needed
    const bool dense = !GATHER && !OUT_UPS && geo.Hs == geo.h && geo.Ws == geo.w && geo.Hd == geo.h && geo.Wd == geo.w;
    // first image of this tile: source offsets are relative to it (32-bit division wherever the pixel count allows: the 64-bit one
    // expands to ~200 instructions)
    const long hw_ = (long)geo.h * geo.w;
    const long n0img = dense ? 0 : (M <= 0xFFFFFFFFL ? (long)((unsigned)m0 / (unsigned)hw_) : m0 / hw_);
    // K chunks in flight per thread (global -> registers), D - 1 of them while the matrix cores work on another: with the
    // 128-pixel tiles (ViT linears: 4,704 tokens, ~1.7 workgroups per CU) one chunk's MFMA phase is ~500 cycles against
    // ~2,000 of load latency, and with a single chunk in flight the loop ran at the latency (17 us for K = 768, 55 us for
    // K = 3,072: 12 / 48 chunks x ~1.2 us)
#ifdef UMI_C1_DEPTH            /* timing experiments: one depth for every tile */
    constexpr int D = UMI_C1_DEPTH;
#else
    // measured (same box, whole steps): a second set helps the 256 x 64 tiles (U-Net transposed convs 24.6 -> 24.3 ms/step),
    // the 128-pixel tiles are best with one (TransUNet 20.85 ms/step against 21.1 with three, 21.6 with two everywhere) -- they
    // are bound by LDS traffic per MFMA, not by the load latency; 256 x 128 has no registers for a second set
    constexpr int D = (P == 256 && BN == 64) ? 2 : 1;
#endif
    long xoff[KPX];                           // plain: element offset of this thread's pixel k in the source, or -1
    int nH[KPX], by[KPX], bx[KPX];            // GATHER: image row base n*Hs and the tap-0 source coordinate of pixel k
    bool xv[D][KPX];                          // this pixel's piece of the chunk held in register set d exists (image and K range)
    // (image, y, x) of the tile's first pixel: ONE 64-bit division per thread; the other pixels of the tile follow from it with
    // small-integer quotients (float reciprocal + one correction step, exact for operands below 2^17).  Eight 64-bit divisions
    // per thread here were ~40 % of the lifetime of a two-chunk workgroup (ConvTranspose of the 128-channel level).
    const int r0img = dense ? 0 : (int)(m0 - n0img * (long)geo.h * geo.w);
    const int y0img = dense ? 0 : r0img / geo.w, x0img = dense ? 0 : r0img - y0img * geo.w;
    const float inv_w = 1.f / (float)geo.w, inv_h = 1.f / (float)geo.h;
#pragma unroll
    for (int k = 0; k < KPX; ++k) {
        const int off = (tid >> 3) + 32 * k;
        long m = m0 + off;
        xoff[k] = -1;
        nH[k] = 0; by[k] = -(1 << 28); bx[k] = 0;          // far outside: every tap of a pixel beyond M is "padding"
        if (m < M && dense) xoff[k] = (m - m0) * ldx + sub * 8;
        else if (m < M) {
            const int a = x0img + off;                     // < w + P
            int qx = (int)((float)a * inv_w), xx = a - qx * geo.w;
            if (xx < 0) { --qx; xx += geo.w; } else if (xx >= geo.w) { ++qx; xx -= geo.w; }
            const int b = y0img + qx;                      // < h + P
            int qy = (int)((float)b * inv_h), yy = b - qy * geo.h;
            if (yy < 0) { --qy; yy += geo.h; } else if (yy >= geo.h) { ++qy; yy -= geo.h; }
            const int nrel = qy;                           // image index relative to the tile's first image
            if (sub == 0) pixinfo[off] = make_int2((int)n0img + nrel, (yy << 16) | xx);
            if (GATHER) {
                nH[k] = nrel * geo.Hs;
                by[k] = geo.frac ? yy + geo.pad : yy * geo.stride - geo.pad + geo.soy;
                bx[k] = geo.frac ? xx + geo.pad : xx * geo.stride - geo.pad + geo.sox;
            } else xoff[k] = ((long)((long)nrel * geo.Hs + yy) * geo.Ws + xx) * ldx + sub * 8;
        }
    }
    long woff[KPW];                           // element offset of weight row k (chunk 0)
    bool wv[KPW];                             // output channel exists (plain mode accepts Co % 8 == 0: the last tile is partial)
    const int Ntot_ = OUT_UPS ? 4 * Nc : Nc;
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
        int cop = c0 + (tid >> 3) + 32 * k;   // output channel index in [0, taps_out * Nc)
        wv[k] = cop < Ntot_;
        if (!wv[k]) cop = c0;                 // keep the address in range; the row is staged as zeros
        int tap = OUT_UPS ? cop / Nc : 0;
        int co = cop - tap * Nc;
        // OUT_UPS : wp8 = [4][Kc/8][Nc][8]  (tap of the OUTPUT);   GATHER: wp8 = [R*S][Kc/8][Nc][8] (tap of the INPUT)
        woff[k] = ((long)((long)tap * (Kc >> 3) + sub) * Nc + co) * 8;
    }
    const int xl = (tid >> 3) * ROWB + sub * 16;            // + k*32*ROWB
    const int wl = XB + (tid >> 3) * ROWB + sub * 16;       // + k*32*ROWB

    // v_mfma_f32_16x16x32_f16: per wave 4 channel tiles x 2*NT pixel tiles of 16 x 16 (the register count of the 32x32x16
    // tiling, the same LDS reads per MAC, half the accumulator traffic per MAC: higher clock under the power cap)
    floatx4 acc[4][2 * NT];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2 * NT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;

    // (kept as 4 x 32 bit: a half8 carried around the loop is split into sixteen-bit values by the compiler, with the unpacking --
    // and a wait for the load -- right behind the load)
    u32x4 xraw[D][KPX], wraw[D][KPW];
    // Every load is unconditional: a piece outside the image / the K range / the channel range gets an offset past the end
    // of its buffer resource and comes back as zeros.  (`cond ? *p : zero` compiled to a branch per load and a
    // s_waitcnt vmcnt(0) right behind the loads: the whole global-memory latency sat in front of every chunk's MFMA phase --
    // 17 us for the 12 chunks of a 4,704 x 768 x 768 linear.)
    constexpr unsigned OOB = 0x7FFFFFFFu;
    // (the source resource starts at the first image this tile touches -- n0img, below -- so that 31-bit offsets are enough
    // for any tensor whose single images are below 1 GB)
    const long ximg_bytes = (long)geo.Hs * geo.Ws * ldx * 2;
    const long xleft = dense ? (M - m0) * ldx * 2 : ((M <= 0xFFFFFFFFL ? (long)((unsigned)M / (unsigned)hw_) : M / hw_) - n0img) * ximg_bytes;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)x + (dense ? m0 * ldx * 2 : n0img * ximg_bytes)), 0, (int)(xleft > 0x7FFFFFF0L ? 0x7FFFFFF0L : xleft), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)wp8, 0, (int)((long)(OUT_UPS ? 4 : ntaps) * Kc * Nc * 2), 0x00020000);

    const int chunks_per_tap = (Kc + CK - 1) / CK;        // plain mode accepts Kc % 8 == 0: the last chunk is partial
    const int nchunks = GATHER ? ntaps * chunks_per_tap : chunks_per_tap;
#define UMI_ISSUE(c_, S_)                                                                                          \
    do {                                                                                                          \
        const int cc = (c_);                                                                                      \
        const int tap_in = GATHER ? cc / chunks_per_tap : 0;                                                      \
        const int kc = cc - tap_in * chunks_per_tap;                                                              \
        const long xadd = (long)kc * CK;                                                                          \
        const long wadd = ((long)tap_in * (Kc >> 3) + kc * 8) * Nc * 8;                                           \
        const bool kin = cc < nchunks && kc * CK + sub * 8 < Kc;   /* this thread's 8 input channels exist in the chunk */ \
        if (GATHER) {                                                                                             \
            const int ty = tap_in / geo.S, tx_ = tap_in - ty * geo.S;                                             \
            _Pragma("unroll") for (int k = 0; k < KPX; ++k) {                                                     \
                int ys = by[k] + ty, xs = bx[k] + tx_;                                                            \
                bool ok = true;                                                                                   \
                if (geo.frac) {                                                                                   \
                    ys = by[k] - ty; xs = bx[k] - tx_;                                                            \
                    ok = ys >= 0 && xs >= 0 && ys % geo.stride == 0 && xs % geo.stride == 0;                      \
                    ys /= geo.stride; xs /= geo.stride;                                                           \
                }                                                                                                 \
                ok = ok && ys >= 0 && ys < geo.Hs && xs >= 0 && xs < geo.Ws;                                       \
                xv[S_][k] = ok && kin;                                                                            \
                xraw[S_][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(                    \
                    xrs, xv[S_][k] ? (unsigned)((((long)(nH[k] + ys) * geo.Ws + xs) * ldx + sub * 8 + xadd) * 2) : OOB, 0, 0)); \
            }                                                                                                     \
        } else {                                                                                                  \
            _Pragma("unroll") for (int k = 0; k < KPX; ++k) {                                                     \
                xv[S_][k] = xoff[k] >= 0 && kin;                                                                  \
                xraw[S_][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(                    \
                    xrs, xv[S_][k] ? (unsigned)((xoff[k] + xadd) * 2) : OOB, 0, 0));                              \
            }                                                                                                     \
        }                                                                                                         \
        _Pragma("unroll") for (int k = 0; k < KPW; ++k)                                                           \
            wraw[S_][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(                        \
                wrs, (wv[k] && kin) ? (unsigned)((woff[k] + wadd) * 2) : OOB, 0, 0));                             \
    } while (0)

    const int lrow = lane & 15, lgrp = lane >> 4;
    const int b_base = (wm * NT * 32 + lrow) * ROWB + lgrp * 16;          // + pt*16*ROWB + ks*64
    const int a_base = XB + (wn * 64 + lrow) * ROWB + lgrp * 16;          // + ct*16*ROWB + ks*64

#define UMI_TXROW(cc_)                                                                                            \
    (((GATHER ? (cc_) % chunks_per_tap : (cc_)) * CK + tid) < Kc ? tx[(GATHER ? (cc_) % chunks_per_tap : (cc_)) * CK + tid] \
                                                                : make_float4(0.f, 1.f, 0.f, 0.f))
    float4 txr = make_float4(0.f, 1.f, 0.f, 0.f);
    if (HAS_TX) {
        if (tid < CK) {
            // stored transposed ([j][sub]): the 8 lanes of a pixel read 8 adjacent float4 (conflict-free) instead of 8 rows
            // 128 B apart (one bank)
            txbuf[0][(tid & 7) * 8 + (tid >> 3)] = UMI_TXROW(0);
            if (nchunks > 1) txbuf[1][(tid & 7) * 8 + (tid >> 3)] = UMI_TXROW(1);
            if (nchunks > 2) txr = UMI_TXROW(2);
        }
        __syncthreads();
    }
    // one K chunk: (transform,) registers of set S_ -> LDS, refill the set with chunk c + D, MFMA.  Every body issues its loads
    // whether or not chunk c + D exists (past the end they are out-of-range = no memory access, zeros): with the same number of
    // loads in flight on every path the compiler can wait for exactly this set's loads (vmcnt(2 x per-set loads)); with the
    // issue under a condition it fell back to vmcnt(0) at the loop head and drained the whole pipeline once per D chunks.
#define UMI_CHUNK(c_, S_)                                                                                          \
    do {                                                                                                          \
        const int c = (c_);                                                                                       \
        const bool live = c < nchunks;                                                                            \
        if (live) {                                                                                               \
            if (HAS_TX) {                                                                                         \
                float4 t[8];                                                                                      \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) t[j] = txbuf[c & 1][j * 8 + sub];                   \
                _Pragma("unroll") for (int k = 0; k < KPX; ++k)                                                   \
                    if (xv[S_][k]) xraw[S_][k] = __builtin_bit_cast(u32x4, umi_tx8(__builtin_bit_cast(half8, xraw[S_][k]), t)); \
            }                                                                                                     \
            _Pragma("unroll") for (int k = 0; k < KPX; ++k) *reinterpret_cast<u32x4*>(smem + xl + k * 32 * ROWB) = xraw[S_][k]; \
            _Pragma("unroll") for (int k = 0; k < KPW; ++k) *reinterpret_cast<u32x4*>(smem + wl + k * 32 * ROWB) = wraw[S_][k]; \
        }                                                                                                         \
        __syncthreads();                                                                                          \
        if (HAS_TX && tid < CK && c + 2 < nchunks) {                                                              \
            txbuf[c & 1][(tid & 7) * 8 + (tid >> 3)] = txr;     /* all readers of this buffer are past the barrier above */ \
            if (c + 3 < nchunks) txr = UMI_TXROW(c + 3);                                                          \
        }                                                                                                         \
        UMI_ISSUE(c + D, S_);                                                                                     \
        if (live) {                                                                                               \
            __builtin_amdgcn_s_setprio(3);          /* MFMA phase outranks the other workgroup's staging (see conv_mfma.hip) */ \
            _Pragma("unroll") for (int ks = 0; ks < CK / 32; ++ks) {                                              \
                half8 af[4];                                                                                      \
                _Pragma("unroll") for (int ct = 0; ct < 4; ++ct)                                                  \
                    af[ct] = *reinterpret_cast<const half8*>(smem + a_base + ct * 16 * ROWB + ks * 64);           \
                _Pragma("unroll") for (int ph = 0; ph < 2; ++ph) {   /* pixel tiles in two halves: 4 + NT fragments live */ \
                    half8 bf[NT];                                                                                 \
                    _Pragma("unroll") for (int i = 0; i < NT; ++i)                                                \
                        bf[i] = *reinterpret_cast<const half8*>(smem + b_base + (ph * NT + i) * 16 * ROWB + ks * 64); \
                    _Pragma("unroll") for (int ct = 0; ct < 4; ++ct)                                              \
                        _Pragma("unroll") for (int i = 0; i < NT; ++i)                                            \
                            UMI_X_MMA(acc[ct][ph * NT + i], af[ct], bf[i]); \
                }                                                                                                 \
            }                                                                                                     \
            __builtin_amdgcn_s_setprio(0);                                                                        \
        }                                                                                                         \
        __syncthreads();                                                                                          \
    } while (0)

    constexpr int S1 = D > 1 ? 1 : 0, S2 = D > 2 ? 2 : 0;          // (in-range indices for the sets a smaller D does not have)
    UMI_ISSUE(0, 0);
    if (D > 1) UMI_ISSUE(1, S1);
    if (D > 2) UMI_ISSUE(2, S2);
    for (int c0_ = 0; c0_ < nchunks; c0_ += D) {
        UMI_CHUNK(c0_, 0);
        if (D > 1) UMI_CHUNK(c0_ + 1, S1);
        if (D > 2) UMI_CHUNK(c0_ + 2, S2);
    }
#undef UMI_CHUNK
#undef UMI_ISSUE
#undef UMI_TXROW

    // ---- epilogue: (+bias) -> fp16 -> LDS tile [pixel][BN] -> 16-B stores (scattered per tap for OUT_UPS) -----
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int col = wn * 64 + ct * 16 + lgrp * 4;           // accumulator rows = 4 consecutive channels per lane
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cop = c0 + col + j;
                bv[j] = cop < Ntot_ ? bias[OUT_UPS ? cop % Nc : cop] : 0.f;
            }
        }
#pragma unroll
        for (int pt = 0; pt < 2 * NT; ++pt) {
            const int pix = wm * NT * 32 + pt * 16 + lrow;
            half4 h;
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = (half_t)(acc[ct][pt][j] + bv[j]);
            *reinterpret_cast<half4*>(smem + pix * ERS + col * 2) = h;
        }
    }
    __syncthreads();
    constexpr int PPR = BN / 8;                 // 16-B pieces per pixel row
    constexpr int PSTEP = 256 / PPR;            // pixels advanced per trip: the piece column j is fixed per thread
    {
        const int j = tid % PPR, p0 = tid / PPR;
        const int cop = c0 + j * 8;
        const bool col_ok = cop < Ntot_;
        float4 bt[8];
        float brs[8], bs[8], bq[8];
        if (geo.epi == 3) {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                bt[jj] = col_ok ? geo.bn_tx[cop + jj] : make_float4(0.f, 1.f, 0.f, 0.f);
                brs[jj] = col_ok ? geo.bn_rstd[cop + jj] : 0.f;
                bs[jj] = bq[jj] = 0.f;
            }
        }
        int co = cop, tdy = 0, tdx = 0;
        if (OUT_UPS) {
            const int tap = cop / Nc;
            co = cop - tap * Nc;
            tdy = (tap >> 1) + geo.doy;
            tdx = (tap & 1) + geo.dox;
        }
        // epi 3: the BatchNorm layer's raw outputs of this thread's pixels, all loads up front (one dependent load per trip of the
        // store loop otherwise: the ConvT data gradients went 0.54 -> 0.77 ms per step with it, i.e. nothing was gained)
        constexpr int NPS = P / PSTEP;
        half8 ybv[NPS];
        if (geo.epi == 3 && !OUT_UPS) {
#pragma unroll
            for (int k = 0; k < NPS; ++k) {
                const int p = p0 + k * PSTEP;
                if (m0 + p < M && col_ok) {
                    const int2 pi = dense ? make_int2(0, 0) : pixinfo[p];
                    const long drow = dense ? (m0 + p) : ((long)pi.x * geo.Hd + (pi.y >> 16)) * geo.Wd + (pi.y & 0xffff);
                    ybv[k] = *reinterpret_cast<const half8*>(geo.aux + drow * geo.ldaux + cop);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NPS; ++k) {
            const int p = p0 + k * PSTEP;
            if (m0 + p >= M || !col_ok) break;
            const int2 pi = dense ? make_int2(0, 0) : pixinfo[p];
            int yy = pi.y >> 16, xx = pi.y & 0xffff;
            if (OUT_UPS) {
                yy = 2 * yy + tdy;
                xx = 2 * xx + tdx;
                if (yy < 0 || yy >= geo.Hd || xx < 0 || xx >= geo.Wd) continue;
            }
            uint4 v = *reinterpret_cast<const uint4*>(smem + p * ERS + j * 16);
            half_t* dst = dense ? y + (m0 + p) * ldy + co : y + ((long)((long)pi.x * geo.Hd + yy) * geo.Wd + xx) * ldy + co;
            if (geo.accum) {
                // a second gradient contribution lands on the tensor the first one wrote: same rounding as adding two
                // stored fp16 tensors
                const half8 o = *reinterpret_cast<const half8*>(dst);
                v = __builtin_bit_cast(uint4, (half8)(o + __builtin_bit_cast(half8, v)));
            }
            if (geo.epi == 1 || geo.epi == 2) {   // dense mode (the launcher guarantees it): row m0 + p, columns cop .. cop + 7
                const long e0 = (m0 + p) * (long)Ntot_ + cop;
                const unsigned seed = geo.seed + (geo.seed_dev ? geo.seed_dev[0] * 0x9E3779B9u : 0u);
                const float scale = 1.f / (1.f - geo.drop_p);
                const half8 xv = __builtin_bit_cast(half8, v);
                half8 av, o;
                if (geo.epi == 2) av = *reinterpret_cast<const half8*>(geo.aux + (m0 + p) * geo.ldaux + cop);
                unsigned long long mk = 0;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const long e = e0 + jj;
                    const float u = (c1_hash32((unsigned)e, seed ^ (unsigned)(e >> 32)) >> 8) * (1.f / 16777216.f);
                    const unsigned k = u >= geo.drop_p;
                    mk |= (unsigned long long)k << (8 * jj);
                    float f = (float)xv[jj];
                    if (geo.epi == 1) f = c1_gelu(f);
                    f = k ? f * scale : 0.f;
                    if (geo.epi == 2) f += (float)av[jj];
                    o[jj] = (half_t)f;
                }
                *reinterpret_cast<unsigned long long*>(geo.mask + e0) = mk;
                if (geo.epi == 1) *reinterpret_cast<half8*>(geo.y2 + (m0 + p) * geo.ldy2 + cop) = o;
                else v = __builtin_bit_cast(uint4, o);
            }
#ifdef UMI_X_NOSTORE
            if (v.x == 0x12345678u)
#endif
            *reinterpret_cast<uint4*>(dst) = v;
            if (geo.epi == 3 && !OUT_UPS) {  // (plain / tap-gather modes: destination pixel = GEMM pixel, any ld)
                const half8 yv = ybv[k];
                const half8 hv = __builtin_bit_cast(half8, v);
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const float yy_ = (float)yv[jj];
                    const float dz = umi_tx_pre(yy_, bt[jj]) > bt[jj].w ? (float)hv[jj] : 0.f;
                    bs[jj] += dz;
                    bq[jj] = fmaf(dz, (yy_ - bt[jj].x) * brs[jj], bq[jj]);
                }
            }
        }
        if (geo.epi == 3) {
            __syncthreads();                                // every thread is done with the output tile: reuse it for the slice sums
            float* rs_ = reinterpret_cast<float*>(smem);    // [2][PSTEP][BN]
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                rs_[(0 * PSTEP + p0) * BN + j * 8 + jj] = bs[jj];
                rs_[(1 * PSTEP + p0) * BN + j * 8 + jj] = bq[jj];
            }
            __syncthreads();
            if (tid < 2 * BN) {
                const int which = tid / BN, cc = tid % BN;
                float a = 0.f;
#pragma unroll 8
                for (int k = 0; k < PSTEP; ++k) a += rs_[(which * PSTEP + k) * BN + cc];
                if (c0 + cc < Ntot_) geo.bn_part[((m0 / P) * 2 + which) * Ntot_ + c0 + cc] = a;
            }
        }
    }
}

template <int P, int BN>
int launch(bool s2d, bool ups, const void* x, int ldx, const void* tx, const void* wp8, const float* bias, void* y,
           int ldy, long M, int Kc, int Nc, int Ntot, int ntaps, Geo geo, hipStream_t s) {
    const int n_co = (Ntot + BN - 1) / BN;
    const long nblk = ((M + P - 1) / P) * n_co;
    dim3 grid((unsigned)((nblk + 7) / 8 * 8)), block(256);
#define GO(S2D, UPS, HT)                                                                                          \
    hipLaunchKernelGGL((conv1x1_mfma_kernel<P, BN, S2D, UPS, HT>), grid, block, 0, s, (const half_t*)x, ldx,      \
                       (const float4*)tx, (const half_t*)wp8, bias, (half_t*)y, ldy, M, Kc, Nc, n_co, ntaps, geo, nblk)
    if (s2d) { if (tx) GO(true, false, true); else GO(true, false, false); }
    else if (ups) { if (tx) GO(false, true, true); else GO(false, true, false); }
    else { if (tx) GO(false, false, true); else GO(false, false, false); }
#undef GO
    UMI_LAUNCH_CHECK();
    return UMI_OK;
}

}  // namespace

// mode: 0 = plain 1x1, 1 = ConvT(2,2) forward (UMI_CONV_UPSAMPLE2), 2 = tap gather (ConvT data gradient = stride-2 2x2 conv,
// strided R x S convs, data gradient of a strided conv)
int umi_conv1x1_mode(int R, int S, int stride, int pad, int flags) {
    if (flags & UMI_CONV_UPSAMPLE2) return (R == 2 && S == 2) ? 1 : -1;
    if (R * S > 49 || pad < 0 || pad >= 1 << 20) return -1;
    if (flags & UMI_CONV_DGRAD_STRIDED) return stride >= 1 ? 2 : -1;
    if (R == 1 && S == 1 && stride == 1 && pad == 0) return 0;
    if (stride >= 2) return 2;
    return -1;
}

bool umi_conv1x1_mfma_ok(int Ci, int Co, int R, int S, int stride, int pad, int ldx, int ldy, int in_dtype,
                         int out_dtype, int flags) {
    if (flags & UMI_CONV_FORCE_GENERIC) return false;
    if (in_dtype != UMI_F16 || out_dtype != UMI_F16) return false;
    const int mode = umi_conv1x1_mode(R, S, stride, pad, flags);
    if (mode < 0) return false;
    if (ldx % 8 || ldy % 8) return false;
    if ((long)(OUT_UPS_TAPS_MAX) * Ci * Co * 2 >= 0x7FFFFFF0L) return false;        // weights behind one 31-bit buffer resource
    // plain 1x1: any Ci, Co that are multiples of 8 (partial last K chunk / output tile; the attention gates' 32-channel
    // branches); the tap-gather / transposed-conv modes keep whole 64-channel tiles
    if (mode == 0) return Ci % 8 == 0 && Co % 8 == 0 && Ci >= 16 && Co >= 16;
    if (Ci % 64 || Co % 64) return false;
    return true;
}

// tile of a problem with M pixels and Ntot output columns: the largest that still gives every CU two workgroups; small GEMMs (ViT
// linears: 4,704 tokens x 768) take 128-pixel tiles, and 64-channel ones if that is still not enough.
// ~0.6 of the 512 resident workgroup slots.  (400 sent the ViT's Q/K/V linear -- 4,704 x 768 -> 2,304: 342 tiles of 256 x 128 --
// to 256 x 64 tiles: 35.2 us against 29.0, tools/ab_gemm.py; the other linears' picks are the same with either value.)
static void c1_pick_tile(long M, int Ntot, int* P, int* BN) {
    const bool bn128 = Ntot % 128 == 0;
    const long want = 300;
    const long b_256_128 = ((M + 255) / 256) * (Ntot / 128), b_128_128 = ((M + 127) / 128) * (Ntot / 128);
    const long b_256_64 = ((M + 255) / 256) * (Ntot / 64);
    // UMI_C1_TILE=PxBN: tile override for timing experiments (tools/ab_gemm.py); read per call
    if (const char* e = getenv("UMI_C1_TILE")) {
        const int p_ = atoi(e), bn_ = strchr(e, 'x') ? atoi(strchr(e, 'x') + 1) : 0;
        if ((p_ == 256 || p_ == 128) && ((bn_ == 128 && bn128) || bn_ == 64)) { *P = p_; *BN = bn_; return; }
    }
    if (bn128 && b_256_128 >= want) { *P = 256; *BN = 128; return; }
    if (b_256_64 >= want) { *P = 256; *BN = 64; return; }
    if (bn128 && b_128_128 >= want) { *P = 128; *BN = 128; return; }
    *P = 128; *BN = 64;
}
// partial-row count of the epi-3 (BatchNorm-reduce) epilogue for a problem of M pixels: one row per pixel tile
int umi_conv1x1_bnred_rows(long M, int Ntot) {
    int P, BN;
    c1_pick_tile(M, Ntot, &P, &BN);
    return (int)((M + P - 1) / P);
}

int umi_conv1x1_mfma(const void* x, int ldx, const void* tx, const void* wp8, const float* bias, void* y, int ldy,
                     int N, int H, int W, int Ci, int Co, int R, int S, int stride, int pad, int Ho, int Wo, int off_h,
                     int off_w, int out_H, int out_W, int flags, hipStream_t s, const UmiLinearEpi* epi) {
    const int mode = umi_conv1x1_mode(R, S, stride, pad, flags);
    if (epi && epi->mode && epi->mode != 3 && (mode != 0 || (flags & UMI_CONV_ACCUMULATE) || Co % 8)) return UMI_ERR_UNSUPPORTED;
    if (epi && epi->mode == 3 && (mode == 1 || Co % 8)) return UMI_ERR_UNSUPPORTED;        // (not for the scattering ConvT forward)
    Geo geo;
    long M;
    int Kc = Ci, Nc = Co, Ntot = Co, ntaps = 1;
    if (mode == 0) {
        geo = Geo{H, W, H, W, 0, 0, H, W, 0, 0, 1, 1, 0, 0};
        M = (long)N * H * W;
    } else if (mode == 1) {            // input grid HxW -> output (out_H x out_W), 4 taps of Co channels each
        geo = Geo{H, W, H, W, 0, 0, out_H, out_W, off_h, off_w, 1, 1, 0, 0};
        M = (long)N * H * W;
        Ntot = 4 * Co;
    } else {                           // source is HxW, GEMM grid = Ho x Wo, K = R*S taps x Ci
        geo = Geo{Ho, Wo, H, W, 0, 0, Ho, Wo, 0, 0, S, stride, pad, (flags & UMI_CONV_DGRAD_STRIDED) ? 1 : 0};
        M = (long)N * Ho * Wo;
        ntaps = R * S;
    }
    // the gather / scatter modes address a source image with 31-bit byte offsets relative to the tile's first image (two
    // images in reach): refuse what does not fit instead of wrapping (~1 GB per image, no shipped config)
    if (mode != 0 && 2L * H * W * ldx * 2 >= 0x7FFFFFF0L) return UMI_ERR_UNSUPPORTED;
    geo.accum = (flags & UMI_CONV_ACCUMULATE) ? 1 : 0;
    geo.epi = 0; geo.drop_p = 0.f; geo.seed = 0; geo.seed_dev = nullptr; geo.mask = nullptr; geo.aux = nullptr; geo.ldaux = 0;
    geo.y2 = nullptr; geo.ldy2 = 0; geo.bn_tx = nullptr; geo.bn_rstd = nullptr; geo.bn_part = nullptr;
    if (epi && epi->mode) {
        geo.epi = epi->mode; geo.drop_p = epi->p; geo.seed = epi->seed; geo.seed_dev = epi->seed_dev;
        geo.mask = (unsigned char*)epi->mask; geo.aux = (const half_t*)epi->aux; geo.ldaux = epi->ldaux;
        geo.y2 = (half_t*)epi->y2; geo.ldy2 = epi->ldy2;
        geo.bn_tx = (const float4*)epi->bn_tx; geo.bn_rstd = epi->bn_rstd; geo.bn_part = epi->bn_part;
    }
#define GO(P_, BN_) return launch<P_, BN_>(mode == 2, mode == 1, x, ldx, tx, wp8, bias, y, ldy, M, Kc, Nc, Ntot, ntaps, geo, s)
    int tp, tbn;
    c1_pick_tile(M, Ntot, &tp, &tbn);
    if (tp == 256 && tbn == 128) GO(256, 128);
    if (tp == 256) GO(256, 64);
    if (tbn == 128) GO(128, 128);
    GO(128, 64);
#undef GO
}
