// The stem: stage-1 down-sampling conv of the event tensor + its LayerNorm (reference maxvit.py:160-177: Conv2d(20 -> 64,
// k = 7, stride 4, pad 3, no bias) -> channels-last LayerNorm), fed with what the data loader hands over — the uint8
// histogram planes [frame][Cin][h][w] (reference modules/detection.py:133-134 casts them, utils/padding.py:29-44 pads
// bottom / right with zeros) — instead of a channels-last bf16 copy of it.
//
// Why its own kernels: through the GEMM engine the stem costs a prepack pass (2.5 GB read, 5.9 GB written, then 5.9 GB
// saved for the backward), an im2col-gather GEMM with K = 49 x 24 padded taps (3.3 ms) and an im2col^T weight-gradient
// GEMM (4.5 ms, 11 GB fetched) — 9.4 ms of a 95 ms step for 1.2 TFLOP each way.  Here both directions read the uint8
// planes (2.5 GB) directly, and the contraction index is ordered (c, ky, kx) with kx padded 7 -> 8:
//
//   forward   lane = output pixel, its eight contraction slots of k-step ks = the eight consecutive input bytes
//             x = 4 ox - 3 .. 4 ox + 4 of plane row (c, iy = 4 oy - 3 + ky), (c, ky) = row 2 ks + (lane >> 5): TWO aligned
//             dwords per lane straight from global memory (neighbouring lanes overlap by half: whole cache lines per
//             wave), converted in registers; weights [k-step][cout][half][8] stay in LDS for the whole launch (144 KB);
//             a wave owns PB = 4 output rows x 32 pixels x 64 channels (8 accumulator blocks) so that one weight
//             fragment read feeds four MFMAs; the epilogue writes y0 (kept for the LayerNorm backward) and LN(y0).
//   backward  dW^T[(c,ky,kx)][cout] = sum over pixels of patch^T dy: the tile's plane rows are converted ONCE into a
//             bf16 LDS image [row (c,ky)][x'], and the operand "row = (row, kx), contract over 16 pixels" — elements
//             4 px + kx + 1 of an image row, stride 4 — is exactly what the transposing LDS read (ds_read_b64_tr_b16)
//             gathers from 8-byte chunks: no per-fragment VALU work at all.  dy^T comes from the same instruction on a
//             row-major dy tile.  Eight waves split the 35 row-blocks of (4 rows x 8 kx-slots); partial sums per
//             workgroup, folded by stem_wgrad_fold_kernel into the engine's raw layout [cout][(ky, kx, c padded)].
//
// uint8 values are exact in bf16, so the products equal those of the prepack + GEMM route; only the fp32 summation
// order differs.  bf16 only (the fp32 parity mode keeps the GEMM route).
#pragma once
#include "common.hpp"
#include "line_bounce.hpp"
#include "attn_block.hpp"

namespace rvt {

struct StemGeom {
    int F, Cin, cp, h, w, Ho, Wo;          // planes [F][Cin][h][w]; output [F][Ho][Wo][64]; cp = padded Cin of the weights
    int NR, KS, KSP;                       // contraction rows (c, ky), k-steps of two rows, k-steps padded to the pipeline depth
    int XS, OG;                            // 32-pixel segments per output row, groups of PB output rows
    int n_items;                           // workgroup items: (frame, eight consecutive (row group, segment) units)
    FastDiv dOG, d7, dXS;                  // dOG: by the number of eight-unit sets per frame; dXS: by the segments per row
};

constexpr int STEM_K = 7, STEM_STRIDE = 4, STEM_PAD = 3, STEM_CO = 64;
constexpr int STEM_KSP_MAX = 72;           // LDS: 72 k-steps x 64 couts x 32 B = 144 KB  (Cin <= 20)

// 7 taps of plane row bytes (d0 = x 4ox-4 .. 4ox-1, d1 = x 4ox .. 4ox+3) -> the k-step fragment: slot e = x 4ox - 3 + e
__device__ __forceinline__ bf16x8 stem_frag_u8(uint32_t d0, uint32_t d1) {
    bf16x8 f;
    f[0] = (bf16)(float)((d0 >> 8) & 0xffu);
    f[1] = (bf16)(float)((d0 >> 16) & 0xffu);
    f[2] = (bf16)(float)(d0 >> 24);
    f[3] = (bf16)(float)(d1 & 0xffu);
    f[4] = (bf16)(float)((d1 >> 8) & 0xffu);
    f[5] = (bf16)(float)((d1 >> 16) & 0xffu);
    // slots 6, 7 as one dword: bf16 of an integer < 256 is the high half of its float; kx = 7 does not exist (zero, like its weight)
    const uint32_t w3 = __builtin_bit_cast(uint32_t, (float)(d1 >> 24)) >> 16;
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
    u32x4_ q = __builtin_bit_cast(u32x4_, f);
    q[3] = w3;
    return __builtin_bit_cast(bf16x8, q);
}

// two dwords at a 4-byte aligned address as ONE 8-byte load
struct __attribute__((packed, aligned(4))) StemU2 { uint32_t a, b; };

// One wave item of the forward: acc[nb][j] += W[32 nb .., (c,ky,kx)] patch[(c,ky,kx), pixel (oy0 + j, ox)] over all k-steps.
// INTERIOR (wave-uniform): every window row and column of the item exists — loads need no bounds logic and the two dwords of a
// lane are adjacent (one 8-byte load from a per-row uniform base + a 32-bit lane offset).
template <int PB, int D, bool INTERIOR>
__device__ __forceinline__ void stem_fwd_item(f32x16 (&acc)[2][PB], const uint8_t* __restrict__ plane0, const char* wl, const StemGeom& g,
                                              int oy0, int half, int xo0, int xo1, bool x0ok, bool x1ok) {
    const int hw = g.h * g.w, coff_max = (g.Cin - 1) * hw;
    int coff = 0, ky = half, kyw = half * g.w;                      // row r = 2 ks + half -> (c, ky); coff = c h w, kyw = ky w
    uint32_t raw[D][PB][2];
    uint32_t msk[D];                                               // border items: bit j = window row of block j exists
    auto load_step = [&](uint32_t (&dst)[PB][2], uint32_t& m) {
        const int cc = coff < coff_max ? coff : coff_max;           // rows past the last plane: zero weights, any finite data
        m = 0;
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int iy0 = 4 * (oy0 + j) - STEM_PAD;               // uniform
            if (INTERIOR) {
                const StemU2 v = *reinterpret_cast<const StemU2*>(plane0 + iy0 * g.w + (uint32_t)(cc + kyw + xo0));
                dst[j][0] = v.a;
                dst[j][1] = v.b;
            } else {
                // unconditional loads from clamped addresses; the zeros are selected when the fragment is built (a load that
                // is only conditionally needed becomes a branch with a wait inside — no pipelining left)
                const bool rowok = (unsigned)(iy0 + ky) < (unsigned)g.h;
                const uint32_t ro = (uint32_t)(cc + (rowok ? iy0 * g.w + kyw : 0));
                dst[j][0] = *reinterpret_cast<const uint32_t*>(plane0 + (ro + (uint32_t)xo0));
                dst[j][1] = *reinterpret_cast<const uint32_t*>(plane0 + (ro + (uint32_t)xo1));
                m |= rowok ? (1u << j) : 0u;
            }
        }
        ky += 2;
        kyw += 2 * g.w;
        const bool wrap = ky >= STEM_K;
        ky -= wrap ? STEM_K : 0;
        kyw -= wrap ? STEM_K * g.w : 0;
        coff += wrap ? hw : 0;
    };
    drain_vmem();                                                   // (the previous item's stores: see common.hpp)
#pragma unroll
    for (int d = 0; d < D; d++) { load_step(raw[d], msk[d]); sched_fence(); }       // in THIS order: the loop waits for the oldest first
    // weight fragments one k-step ahead as well: read right in front of the MFMAs that need them, every k-step would start
    // with an LDS round trip
    bf16x8 b0 = *reinterpret_cast<const bf16x8*>(wl), b1 = *reinterpret_cast<const bf16x8*>(wl + 1024);
    for (int ks0 = 0; ks0 < g.KSP; ks0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            bf16x8 a[PB];
#pragma unroll
            for (int j = 0; j < PB; j++) {
                uint32_t d0 = raw[d][j][0], d1 = raw[d][j][1];
                if (!INTERIOR) {
                    const bool rowok = (msk[d] >> j) & 1u;
                    d0 = (rowok && x0ok) ? d0 : 0u;
                    d1 = (rowok && x1ok) ? d1 : 0u;
                }
                a[j] = stem_frag_u8(d0, d1);
            }
            // the fences pin the software pipeline: left alone, the scheduler sinks the loads next to their uses (one k-step
            // of latency cover instead of D) and drains the queue at every loop back-edge
            sched_fence();
            load_step(raw[d], msk[d]);                              // rows of k-step ks0 + d + D
            const int kn = ks0 + d + 1 < g.KSP ? ks0 + d + 1 : 0;
            const bf16x8 n0 = *reinterpret_cast<const bf16x8*>(wl + (size_t)kn * 2048);
            const bf16x8 n1 = *reinterpret_cast<const bf16x8*>(wl + (size_t)kn * 2048 + 1024);
            sched_fence();
#pragma unroll
            for (int j = 0; j < PB; j++) {
                mma32(acc[0][j], b0, a[j]);                         // rows = channels, column = this lane's pixel
                mma32(acc[1][j], b1, a[j]);
            }
            b0 = n0;
            b1 = n1;
        }
    }
}

template <int PB, int D>
__global__ void __launch_bounds__(512)
stem_fwd_kernel(const uint8_t* __restrict__ src, const bf16* __restrict__ wp, const float* __restrict__ ln_w,
                const float* __restrict__ ln_b, bf16* __restrict__ y0, bf16* __restrict__ xo, StemGeom g, float eps) {
    typedef bf16 T;
    constexpr int OFF_SCR = STEM_KSP_MAX * 2048 + 2 * STEM_CO * 4;
    __shared__ __attribute__((aligned(16))) char smem[OFF_SCR + 8 * LineBounceT<8>::BYTES];      // (144 KB of weights: room for the 8-token bounce only)
    char* const Wl = smem;
    float* const kst = reinterpret_cast<float*>(smem + STEM_KSP_MAX * 2048);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, half = lane >> 5, wave = tid >> 6;
    const int KK = STEM_K * STEM_K * g.cp;
    // weights: fragment (ks, cout n, half) = taps kx = 0..6 of row r = 2 ks + half = (c, ky)
    for (int f = tid; f < g.KSP * 128; f += 512) {
        const int hf = f & 1, n = (f >> 1) & 63, ks = f >> 7;
        const int r = 2 * ks + hf;
        bf16x8 v = frag_zero<T>();
        if (r < g.NR) {
            uint32_t c, ky;
            g.d7.divmod((uint32_t)r, c, ky);
            const T* q = wp + (size_t)n * KK + (size_t)(ky * STEM_K) * g.cp + c;
#pragma unroll
            for (int e = 0; e < STEM_K; e++) v[e] = q[e * g.cp];
        }
        *reinterpret_cast<bf16x8*>(Wl + (size_t)f * 16) = v;
    }
    for (int i = tid; i < STEM_CO; i += 512) { kst[i] = ln_w[i]; kst[STEM_CO + i] = ln_b[i]; }
    __syncthreads();

    const size_t hw = (size_t)g.h * g.w;
    // Work split (round 6): a workgroup item = eight consecutive (row group, segment) UNITS of one frame in row-major order, one per
    // wave, all in flight at the same time.  A segment's window [128 xs - 4, 128 xs + 128) reaches one dword into the cache line of the
    // segment to its left and three plane rows into the row group above: with the left / upper neighbour running on a neighbouring
    // wave of the SAME workgroup those lines are in flight together.  Measured (profiles/r6/pmc_stem_fwd.txt): 4.36 GB fetched for
    // 2.32 GB of planes and 1.30 ms, against 4.72 GB / 1.34 ms for the round-2 split (a wave walking its row group segment by segment:
    // it came back to the neighbour line 50 KB per wave later) and 4.4 GB for units dealt round-robin over workgroups.  The fabric
    // request counter agrees with FETCH_SIZE (TCC_EA0_RDREQ x 64 B = the raw counter); most of the remaining 1.9 x is NOT the halo
    // (19 / 16 rows, 132 / 128 bytes) and is not explained yet.
    const int wv = wave_uniform(wave);
    for (int item = blockIdx.x; item < g.n_items; item += gridDim.x) {
        uint32_t f, u8;
        g.dOG.divmod((uint32_t)item, f, u8);
        const int unit = 8 * (int)u8 + wv;
        if (unit >= g.OG * g.XS) continue;
        uint32_t ogu, xsu;
        g.dXS.divmod((uint32_t)unit, ogu, xsu);
        const int og = (int)ogu, xs = (int)xsu;
        const int oy0 = PB * og, ox = 32 * xs + li;
        const bool colok = ox < g.Wo;
        // per-lane source columns: d0 = bytes x 4ox-4 .. 4ox-1 (the zero padding at ox = 0), d1 = bytes x 4ox .. 4ox+3 (zeros past
        // the real width w).  A lane whose pixel does not exist only needs SAFE addresses: it is a column nobody stores.
        const bool x0ok = colok && ox > 0 && 4 * ox - 4 < g.w, x1ok = colok && 4 * ox + 3 < g.w;
        const int xo0 = x0ok ? 4 * ox - 4 : 0, xo1 = x1ok ? 4 * ox : 0;
        const uint8_t* const plane0 = src + (size_t)f * g.Cin * hw;
        // wave-uniform: no row of this item's windows falls into the top / bottom padding, no existing pixel touches the left
        // padding or the columns past w
        const int ox_last = (32 * (int)xs + 31 < g.Wo ? 32 * (int)xs + 31 : g.Wo - 1);
        const bool interior = xs > 0 && 4 * ox_last + 3 < g.w && 4 * oy0 - STEM_PAD >= 0 &&
                              4 * (oy0 + PB - 1) - STEM_PAD + STEM_K - 1 < g.h;
        f32x16 acc[2][PB];
#pragma unroll
        for (int nb = 0; nb < 2; nb++)
#pragma unroll
            for (int j = 0; j < PB; j++) acc_zero(acc[nb][j]);
        const char* wl = Wl + (size_t)(li * 2 + half) * 16;
        if (interior) stem_fwd_item<PB, D, true>(acc, plane0, wl, g, oy0, half, xo0, xo1, x0ok, x1ok);
        else stem_fwd_item<PB, D, false>(acc, plane0, wl, g, oy0, half, xo0, xo1, x0ok, x1ok);
        // epilogue: y0 (bf16, what the LayerNorm backward reads) and LN(y0) as 16-byte row pieces
#pragma unroll
        for (int j = 0; j < PB; j++) {
            const int oy = oy0 + j;
            float v[2][2][8];
            acc_to_rows(acc[0][j], v[0]);
            acc_to_rows(acc[1][j], v[1]);
            bf16x8 yb[2][2];
            float s = 0.f;
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    yb[nb][m] = frag_from_float<T>(v[nb][m]);
                    frag_to_float<T>(yb[nb][m], v[nb][m]);          // the statistics see what the backward will see
#pragma unroll
                    for (int e = 0; e < 8; e++) s += v[nb][m][e];
                }
            s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / STEM_CO);
            float ss = 0.f;
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int e = 0; e < 8; e++) { const float dlt = v[nb][m][e] - mean; ss += dlt * dlt; }
            ss += __shfl_xor(ss, 32);
            const float rstd = 1.0f / sqrtf(ss * (1.0f / STEM_CO) + eps);
            // rows leave as full 128-byte lines (line_bounce.hpp): a lane owns a pixel, so 16-byte pieces stored directly are 32 bytes
            // of 32 different lines per instruction; the 32 pixels of the segment are 4 KiB of consecutive memory
            u32x4 pcy[2][2], pcx[2][2];
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    const int c0 = 32 * nb + 16 * m + 8 * half;
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = fmaf((v[nb][m][e] - mean) * rstd, kst[c0 + e], kst[STEM_CO + c0 + e]);
                    pcy[nb][m] = __builtin_bit_cast(u32x4, yb[nb][m]);
                    pcx[nb][m] = __builtin_bit_cast(u32x4, frag_from_float<T>(o));
                }
            const int px0 = 32 * (int)xs;
            const int npx = wave_uniform(oy < g.Ho ? (g.Wo - px0 < 32 ? g.Wo - px0 : 32) : 0);      // existing pixels of the segment (>= 1 unless the row is past Ho)
            const size_t pbase = (((size_t)f * g.Ho + (oy < g.Ho ? oy : 0)) * g.Wo + px0) * STEM_CO;
            LineBounceT<8> lb;
            lb.init(smem + OFF_SCR + wv * LineBounceT<8>::BYTES, lane);
            lb.flush(pp_make_rsrc(y0 + pbase, (unsigned)(npx * STEM_CO * 2)), pcy, STEM_CO * 2, 0);
            lb.flush(pp_make_rsrc(xo + pbase, (unsigned)(npx * STEM_CO * 2)), pcx, STEM_CO * 2, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------ weight gradient
constexpr int STEM_WG_ROWB = 304;          // bytes per LDS image row: 136 bf16 + pad; rows 48 B apart modulo 256 (banks)
constexpr int STEM_WG_ROWS = 144;          // image rows (c, ky)  (Cin <= 20)
constexpr int STEM_WG_DYB = 144;           // bytes per dy tile row (64 bf16 + pad)
constexpr int STEM_WG_JW = 5;              // row-blocks (4 rows x 8 slots) per wave: 8 x 5 >= 35
constexpr int STEM_WG_DPR = 33;            // dwords (4 source bytes each) per image row: x' = 0 .. 131 (the last one read)
constexpr int STEM_WG_RPR = 15;            // image rows per staging round: 15 x 33 = 495 of the 512 threads
constexpr int STEM_WG_ROUNDS = 10;

struct StemWgGeom {
    int F, Cin, h, w, Ho, Wo;
    int NR, NJB;                           // contraction rows (c, ky); row-blocks of 4
    int XS, n_tiles, per_wg;
    FastDiv dXS, dHo, d7;
};

// ws: [workgroup][j = 8 r + e'][cout] partial sums, r = (c, ky) row, e' = kx + 1 (e' = 0 unused)
// LNB: `dy` is the gradient at the LayerNorm OUTPUT and `y0` the conv output: the LayerNorm backward (maxvit.py:177) is done
// while the dy tile is staged (sixteen threads per pixel row hold it anyway) — dy0 never exists in HBM, its parameter
// gradients are accumulated per thread in LDS and added to dln_w / dln_b at the end.
template <bool LNB>
__global__ void __launch_bounds__(512)
stem_wgrad_kernel(const uint8_t* __restrict__ src, const bf16* __restrict__ dy, const bf16* __restrict__ y0,
                  const float* __restrict__ ln_w, float* __restrict__ dln_w, float* __restrict__ dln_b, float* __restrict__ ws,
                  StemWgGeom g, float eps) {
    typedef bf16 T;
    constexpr int IMG = STEM_WG_ROWS * STEM_WG_ROWB, DYT = 32 * STEM_WG_DYB;
    __shared__ __attribute__((aligned(16))) char smem[2 * IMG + 2 * DYT + STEM_WG_ROUNDS * 512 * 4 + (LNB ? 512 * 32 : 16)];
    float* const dlp = reinterpret_cast<float*>(smem + 2 * IMG + 2 * DYT + STEM_WG_ROUNDS * 512 * 4) + 8 * threadIdx.x;   // LNB: this thread's dln_w / dln_b partials
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, half = lane >> 5, wave = tid >> 6;
    const int hw = g.h * g.w;
    uint32_t* const voff = reinterpret_cast<uint32_t*>(smem + 2 * IMG + 2 * DYT) + tid;       // [k][thread]: see below

    // Staging role of this thread: dword column dq of the image rows row0 + 15 k, k = 0..9.  What does not depend on the tile
    // is computed once: the source offset of each of the ten dwords relative to the tile's window origin (kept in an LDS
    // table this thread alone reads: ten registers less), the tap row ky of each (3 bits), and which of them exist at all.
    const int row0 = tid / STEM_WG_DPR, dq = tid - row0 * STEM_WG_DPR;
    uint32_t kys = 0, rmask = 0;
    {
        uint32_t c, ky;
        g.d7.divmod((uint32_t)row0, c, ky);
#pragma unroll
        for (int k = 0; k < STEM_WG_ROUNDS; k++) {
            const bool ok = row0 < STEM_WG_RPR && row0 + STEM_WG_RPR * k < g.NR;
            voff[512 * k] = ok ? (uint32_t)((int)c * hw + (int)ky * g.w + 4 * dq) : 0u;
            kys |= ky << (3 * k);
            rmask |= ok ? (1u << k) : 0u;
            c += 2;                                                // + 15 rows = + 2 planes + 1 tap row
            ky += 1;
            if (ky >= STEM_K) { ky -= STEM_K; c += 1; }
        }
    }
    // dy tile: 32 rows x 128 B = 512 pieces of 8 bytes
    const int dyr = tid >> 4, dyc = tid & 15;

    // operand addressing of the transposing reads (see the header): lane L of a 16-lane group supplies the chunk
    // (row 2 (L>>4 & 1) + (L&3 >> 1), chunk (L&15 >> 2) + (L & 1)) and receives row 2 (L>>4 & 1) + (L&15 >> 3), slot e' = L & 7
    const int a_off = (2 * ((lane >> 4) & 1) + ((lane & 3) >> 1)) * STEM_WG_ROWB + (8 * half + ((lane & 15) >> 2) + (lane & 1)) * 8;
    const int tl = 8 * half + ((lane & 15) >> 2), fl = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int b_off = tl * STEM_WG_DYB + fl * 2;

    f32x16 acc[STEM_WG_JW][2];
#pragma unroll
    for (int i = 0; i < STEM_WG_JW; i++) { acc_zero(acc[i][0]); acc_zero(acc[i][1]); }

    const int t_begin = blockIdx.x * g.per_wg, t_end = (t_begin + g.per_wg < g.n_tiles) ? t_begin + g.per_wg : g.n_tiles;
    // Two tiles of source dwords in flight per thread (register sets A, B): a tile is requested two iterations before it is
    // converted into the LDS image.
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    struct Stage { uint32_t raw[STEM_WG_ROUNDS]; u32x2 dyraw, y0raw; uint32_t ok; };
    if (LNB) { *reinterpret_cast<f32x4*>(dlp) = f32x4{0.f, 0.f, 0.f, 0.f}; *reinterpret_cast<f32x4*>(dlp + 4) = f32x4{0.f, 0.f, 0.f, 0.f}; }
    auto fetch = [&](Stage& st, int tile) {      // global -> registers
        if (tile >= t_end) return;
        uint32_t fo, xs, f, oy;
        g.dXS.divmod((uint32_t)tile, fo, xs);
        g.dHo.divmod(fo, f, oy);
        const int x0 = 128 * (int)xs - 4, iy0 = 4 * (int)oy - STEM_PAD;        // window origin (uniform)
        const uint8_t* const org = src + ((size_t)f * g.Cin * hw + (ptrdiff_t)iy0 * g.w + x0);
        if (x0 >= 0 && x0 + 4 * STEM_WG_DPR <= g.w && iy0 >= 0 && iy0 + STEM_K <= g.h) {
            // interior tile: every dword of the window exists — scalar base + precomputed lane offsets, no bounds logic
#pragma unroll
            for (int k = 0; k < STEM_WG_ROUNDS; k++) st.raw[k] = *reinterpret_cast<const uint32_t*>(org + voff[512 * k]);
            st.ok = rmask;
        } else {
            const int x = x0 + 4 * dq;
            const bool xok = x >= 0 && x < g.w;
            uint32_t okm = 0;
#pragma unroll
            for (int k = 0; k < STEM_WG_ROUNDS; k++) {
                const int iy = iy0 + (int)((kys >> (3 * k)) & 7u);
                const bool ok = ((rmask >> k) & 1u) && xok && (unsigned)iy < (unsigned)g.h;
                st.raw[k] = *reinterpret_cast<const uint32_t*>(ok ? org + voff[512 * k] : src);      // (zeroed when stashed)
                okm |= ok ? (1u << k) : 0u;
            }
            st.ok = okm;
        }
        const int ox = 32 * (int)xs + dyr;
        const bool dok = ox < g.Wo;
        const u32x2 z = {0u, 0u};
        const size_t doff = (((size_t)f * g.Ho + oy) * g.Wo + (dok ? ox : 0)) * STEM_CO + 4 * dyc;
        const u32x2 v = *reinterpret_cast<const u32x2*>(dy + doff);
        st.dyraw = dok ? v : z;
        if (LNB) {
            const u32x2 vy = *reinterpret_cast<const u32x2*>(y0 + doff);
            st.y0raw = dok ? vy : z;
        }
    };
    auto stash = [&](const Stage& st, int tile, int buf) {         // registers -> bf16 LDS image
        if (tile >= t_end) return;
        char* const img = smem + buf * IMG + row0 * STEM_WG_ROWB + dq * 8;
        if (row0 < STEM_WG_RPR) {
#pragma unroll
            for (int k = 0; k < STEM_WG_ROUNDS; k++) {
                if (STEM_WG_RPR * k + STEM_WG_RPR <= STEM_WG_ROWS || row0 + STEM_WG_RPR * k < STEM_WG_ROWS) {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    const uint32_t v = ((st.ok >> k) & 1u) ? st.raw[k] : 0u;
                    bf16x4 q;
                    q[0] = (bf16)(float)(v & 0xffu);
                    q[1] = (bf16)(float)((v >> 8) & 0xffu);
                    q[2] = (bf16)(float)((v >> 16) & 0xffu);
                    q[3] = (bf16)(float)(v >> 24);
                    *reinterpret_cast<bf16x4*>(img + STEM_WG_RPR * k * STEM_WG_ROWB) = q;
                }
            }
        }
        u32x2 dyv = st.dyraw;
        if (LNB) {
            typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
            const bf16x4 yb = __builtin_bit_cast(bf16x4, st.y0raw), gb = __builtin_bit_cast(bf16x4, st.dyraw);
            const f32x4 lw = *reinterpret_cast<const f32x4*>(ln_w + 4 * dyc);
            float x[4], gq[4], dyf[4];
#pragma unroll
            for (int i = 0; i < 4; i++) { x[i] = (float)yb[i]; dyf[i] = (float)gb[i]; }
            const float mean = row16_sum((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / STEM_CO);
            float ss = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) { x[i] -= mean; ss += x[i] * x[i]; }
            const float rstd = 1.0f / sqrtf(row16_sum(ss) * (1.0f / STEM_CO) + eps);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; i++) { x[i] *= rstd; gq[i] = dyf[i] * lw[i]; s1 += gq[i]; s2 += gq[i] * x[i]; }
            s1 = row16_sum(s1) * (1.0f / STEM_CO);
            s2 = row16_sum(s2) * (1.0f / STEM_CO);
            bf16x4 o;
            f32x4 pw = *reinterpret_cast<const f32x4*>(dlp), pb = *reinterpret_cast<const f32x4*>(dlp + 4);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                o[i] = (bf16)(rstd * (gq[i] - s1 - x[i] * s2));
                pw[i] += dyf[i] * x[i];
                pb[i] += dyf[i];
            }
            *reinterpret_cast<f32x4*>(dlp) = pw;
            *reinterpret_cast<f32x4*>(dlp + 4) = pb;
            dyv = __builtin_bit_cast(u32x2, o);
        }
        *reinterpret_cast<u32x2*>(smem + 2 * IMG + buf * DYT + dyr * STEM_WG_DYB + dyc * 8) = dyv;
    };
    auto compute = [&](int buf) {
        const char* const img = smem + buf * IMG;
        const char* const dyt = smem + 2 * IMG + buf * DYT;
        bf16x8 bq[2][2];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int nb = 0; nb < 2; nb++) {
                const char* p = dyt + b_off + 16 * q * STEM_WG_DYB + 64 * nb;
                bq[q][nb] = frag_from_tr<T>(reinterpret_cast<const bf16*>(p), reinterpret_cast<const bf16*>(p + 4 * STEM_WG_DYB));
            }
#pragma unroll
        for (int i = 0; i < STEM_WG_JW; i++) {
            const int jb = wave + 8 * i;
            if (jb < g.NJB) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const char* p = img + a_off + jb * 4 * STEM_WG_ROWB + 128 * q;
                    const bf16x8 a = frag_from_tr<T>(reinterpret_cast<const bf16*>(p), reinterpret_cast<const bf16*>(p + 32));
                    mma32(acc[i][0], a, bq[q][0]);
                    mma32(acc[i][1], a, bq[q][1]);
                }
            }
        }
    };
    // The two waves that share a SIMD (w and w + 4) take the two halves of an iteration in opposite order: one feeds the
    // matrix pipe while the other converts — in lockstep both would convert, then both would queue on the pipe.
    const bool mfma_first = wave < 4;
    Stage sa, sb;
    fetch(sa, t_begin);
    fetch(sb, t_begin + 1);
    stash(sa, t_begin, 0);
    fetch(sa, t_begin + 2);
    lds_barrier();
    for (int tile = t_begin; tile < t_end; tile += 2) {
        if (mfma_first) compute(0);               // tile
        stash(sb, tile + 1, 1);
        fetch(sb, tile + 3);
        if (!mfma_first) compute(0);
        lds_barrier();
        if (mfma_first && tile + 1 < t_end) compute(1);       // tile + 1
        stash(sa, tile + 2, 0);
        fetch(sa, tile + 4);
        if (!mfma_first && tile + 1 < t_end) compute(1);
        lds_barrier();
    }
    float* const out = ws + (size_t)blockIdx.x * (size_t)(g.NJB * 32) * STEM_CO;
#pragma unroll
    for (int i = 0; i < STEM_WG_JW; i++) {
        const int jb = wave + 8 * i;
        if (jb < g.NJB) {
#pragma unroll
            for (int nb = 0; nb < 2; nb++)
#pragma unroll
                for (int r = 0; r < 16; r++) out[(size_t)(32 * jb + acc_row(r, lane)) * STEM_CO + 32 * nb + li] = acc[i][nb][r];
        }
    }
    if (LNB) {                                     // thread (row slot dyr, piece dyc) -> channel sums over the 32 row slots
        __syncthreads();
        if (tid < 2 * STEM_CO) {
            const int c = tid & (STEM_CO - 1), which = tid >> 6;
            const float* p = reinterpret_cast<const float*>(smem + 2 * IMG + 2 * DYT + STEM_WG_ROUNDS * 512 * 4) + 8 * (c >> 2) + 4 * which + (c & 3);
            float sum = 0.f;
            for (int r = 0; r < 32; r++) sum += p[r * 16 * 8];
            atomicAdd((which ? dln_b : dln_w) + c, sum);
        }
    }
}

// dw[cout][(ky 7 + kx) cp + c] += sum over workgroups of ws[wg][8 (c 7 + ky) + kx + 1][cout]   (c < Cin; the padded
// channels of the raw layout are left alone)
__global__ void __launch_bounds__(256)
stem_wgrad_fold_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nwg, int Cin, int cp, int NJB) {
    const int KK = STEM_K * STEM_K * cp;
    const int total = STEM_CO * STEM_K * STEM_K * Cin;
    const size_t stride = (size_t)(NJB * 32) * STEM_CO;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int n = idx & 63, rest = idx >> 6;            // cout fastest: the partials are read in whole 256-byte rows
        const int c = rest % Cin, tap = rest / Cin, ky = tap / STEM_K, kx = tap % STEM_K;
        const float* p = ws + ((size_t)8 * (c * STEM_K + ky) + kx + 1) * STEM_CO + n;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};    // independent chains (see splitk_reduce_kernel)
        int wgi = 0;
        for (; wgi + 8 <= nwg; wgi += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) a[u] += p[(size_t)(wgi + u) * stride];
        }
        for (; wgi < nwg; wgi++) a[0] += p[(size_t)wgi * stride];
        dw[(size_t)n * KK + (size_t)tap * cp + c] += ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
}

}  // namespace rvt
