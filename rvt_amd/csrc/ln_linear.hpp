// LayerNorm + linear in one launch for the qkv projection of a C = 128 block (reference maxvit.py:268 -> :347: `qkv(norm1(x))`):
//
//     u = LN(x; ln_w, ln_b)            (kept: the B operand of the qkv weight gradient; u = x when the block has no norm1)
//     y = u W^T + bias                 (W [N][C], N = 3C)
//
// The op-by-op route runs rvt_layernorm_fwd (read x, write u) and an NT GEMM with K = 128 whose 128-row tiles spend their time on
// the operand staging and the store of 3C columns per token (0.11 + 0.64 ms at 1.9 M tokens; the rows alone are 2.5 GB = 0.31 ms
// of HBM time).  Here the whole weight matrix (96 KiB bf16) is an LDS-DMA image loaded once per persistent workgroup; a wave owns
// 32 token rows in MFMA-operand form (the data flow of mlp_chain.hpp / mlp_stream.hpp), normalises them in registers and walks the
// 12 blocks of 32 output columns: 8 MFMAs per block, bias as the initial accumulator, the block leaves as two 16-byte row pieces
// per lane.  The waves never synchronise after the image has landed; the next tile's rows are in flight while a tile computes.
#pragma once
#include "common.hpp"
#include "mlp_chain.hpp"
#include "mlp_stream.hpp"
#include "ppgemm.hpp"
#include "line_bounce.hpp"
#include "gelu_lut.hpp"

namespace rvt {

template <class T, int C, int N> struct LnLinGeom {
    static constexpr int KT = C / TileGeom<T>::BK;
    static constexpr int W_BYTES = KT * N * 128;          // [N rows][C] as KT sub-tile columns
    static constexpr int K_LNW = 0, K_LNB = C, K_BIAS = 2 * C, NCONST = 2 * C + N;
    static constexpr int OFF_SCR = W_BYTES + NCONST * 4;
    template <int WPB> static constexpr int smem() { return OFF_SCR + WPB * LineBounce::BYTES; }
};

template <class T, int C, int N, int WPB, int MINW>
__global__ void __launch_bounds__(64 * WPB, MINW)
lnlin_fwd_kernel(const T* __restrict__ x, const float* __restrict__ ln_w, const float* __restrict__ ln_b, const T* __restrict__ W,
                 const float* __restrict__ bias, T* __restrict__ u, T* __restrict__ y, int M, float eps) {
    typedef LnLinGeom<T, C, N> G;
    constexpr int KS = C / 16, NB = N / 32;
    constexpr int NPW = (G::W_BYTES / 1024) / WPB;        // 1-KiB pieces of the weight image per wave
    static_assert(NPW * WPB * 1024 == G::W_BYTES, "weight pieces must divide over the waves");
    static_assert(NB % 2 == 0 && KS % 4 == 0, "output blocks are walked in pairs");
    __shared__ __attribute__((aligned(16))) char smem[G::template smem<WPB>()];
    float* const kst = reinterpret_cast<float*>(smem + G::W_BYTES);
    const int tid = threadIdx.x, lane_ = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    {
        const pp_rsrc rw = pp_make_rsrc(W, (unsigned)(N * C * sizeof(T)));
        int v[NPW];
        ms_piece_offsets<T, NPW, WPB>(v, wave, lane_, N, C);
#pragma unroll
        for (int i = 0; i < NPW; i++) pp_glds16(rw, smem, (wave + i * WPB) * 1024, v[i], 0);
    }
    const bool has_ln = ln_w != nullptr;              // (the first block behind a down-sampling conv has no norm1: maxvit.py:229-236)
    if (has_ln)
        for (int i = tid; i < C; i += 64 * WPB) { kst[G::K_LNW + i] = ln_w[i]; kst[G::K_LNB + i] = ln_b[i]; }
    for (int i = tid; i < N; i += 64 * WPB) kst[G::K_BIAS + i] = bias ? bias[i] : 0.f;
    pp_wait_vm<0>();
    __syncthreads();

    const int n_tiles = (M + 31) / 32;
    const int stride = (int)gridDim.x * WPB;
    int tile = blockIdx.x * WPB + wave;
    if (tile >= n_tiles) return;
    // Rows enter and leave through BUFFER accesses whose resource covers exactly the valid rows of the wave's tile: rows beyond M
    // load zeros and their stores are dropped, so the tile loop has no branch and no select on a loaded value.  That matters
    // because loads and stores retire through ONE in-order counter: with straight-line code hipcc waits for the prefetched rows
    // with vmcnt(<everything issued behind them>); behind a conditional store it has to assume the store was skipped and waits
    // for vmcnt(7) - i.e. for every store of the tile.  Everything lane-derived is recomputed per tile from an OPAQUE copy of the
    // lane id, or the LayerNorm constants of the lane's 64 columns are hoisted into 128 registers.
    // (wave_uniform: hipcc clamps with a VALU med3, and a resource word in a vector register costs a waterfall loop per access)
    auto rows_of = [&](int t) { const int r = M - t * 32; return wave_uniform(r < 0 ? 0 : (r > 32 ? 32 : r)); };
    auto load_rows = [&](frag_t<T> (&f)[KS], int t, int lane) __attribute__((always_inline)) {
        const pp_rsrc rx = pp_make_rsrc(x + (size_t)t * 32 * C, (unsigned)(rows_of(t) * C * (int)sizeof(T)));
        const int off = (lane & 31) * C * (int)sizeof(T) + (lane >> 5) * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) f[ks] = __builtin_bit_cast(frag_t<T>, pp_load16(rx, off + 32 * ks));
    };
    frag_t<T> xn[KS];
    load_rows(xn, tile, lane_);
#ifndef RVT_EMU
    // (a use in front of the loop: hipcc then waits for these loads HERE; otherwise the loop top inherits "eight loads may be
    //  pending" from this edge and waits with vmcnt(8..15) on every iteration - for the previous tile's stores)
#pragma unroll
    for (int ks = 0; ks < KS; ks++) asm volatile("" : "+v"(xn[ks]));
#endif
    for (; tile < n_tiles; tile += stride) {
        int lane = lane_;
        opaque_vgpr(lane);
        const int li = lane & 31, half = lane >> 5;
        const int rb = ms_rowbase_h<T>(li, half);
        frag_t<T> uf[KS];
        {
            frag_t<T> xf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) xf[ks] = xn[ks];
            load_rows(xn, tile + stride, lane);           // (beyond the last tile: an empty resource, zeros)
            sched_fence();
            if (has_ln) {
                float mean, rstd;
                mc_layernorm<T, C>(xf, uf, kst + G::K_LNW, kst + G::K_LNB, true, half, eps, mean, rstd);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ks++) uf[ks] = xf[ks];
            }
        }
        const int nrows = rows_of(tile);
        LineBounce lb;
        lb.init(smem + G::OFF_SCR + wave * LineBounce::BYTES, lane);
        {
            const pp_rsrc ru = pp_make_rsrc(u + (size_t)tile * 32 * C, u != nullptr && has_ln ? (unsigned)(nrows * C * (int)sizeof(T)) : 0u);
#pragma unroll
            for (int p = 0; p < KS / 4; p++) {            // 64-column groups of the normalised rows: piece (j, m) = k-step 4 p + 2 j + m
                u32x4 pc[2][2];
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int m = 0; m < 2; m++) pc[j][m] = __builtin_bit_cast(u32x4, uf[4 * p + 2 * j + m]);
                lb.flush(ru, pc, C * (int)sizeof(T), 128 * p);
            }
        }
        const pp_rsrc ry = pp_make_rsrc(y + (size_t)tile * 32 * N, (unsigned)(nrows * N * (int)sizeof(T)));
#pragma unroll
        for (int nb = 0; nb < NB; nb += 2) {
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; j++) acc_load_rows(acc[j], kst + G::K_BIAS + 32 * (nb + j), half);
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
#pragma unroll
                for (int j = 0; j < 2; j++) mma32(acc[j], ms_load_frag<T>(smem + (nb + j) * 32 * 128, N, rb, 2 * ks), uf[ks]);
            u32x4 pc[2][2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float r8[2][8];
                acc_to_rows(acc[j], r8);
#pragma unroll
                for (int m = 0; m < 2; m++) pc[j][m] = __builtin_bit_cast(u32x4, frag_from_float<T>(r8[m]));
            }
            lb.flush(ry, pc, N * (int)sizeof(T), 64 * (nb / 2) * (int)sizeof(T));
            sched_fence();                                // (unrolled for the store count only: without the fence hipcc interleaves all six pairs - 192 accumulator registers)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// fc1 + GELU of the MLP half of a C = 256 block (reference maxvit.py:100-112: `act(fc1(.))`) in the form the op-by-op backward wants:
// g = GELU(x W1^T + b1) (operand of fc2 and of its weight gradient) and gp = GELU'(.) (factor of the fc2 input gradient), so that no
// later kernel re-evaluates erf.
//
// On the NT GEMM engines this product (K = 256: tiles live four k-steps and leave through an LDS staging epilogue, two tensors stored)
// takes 0.78 - 0.89 ms at 484 k tokens for 2.2 GB of rows (0.76 ms with the GELU arithmetic compiled out: it is not VALU-bound there).
// W1 is 512 KiB.  Measured first, and removed (all with the LayerNorm in front fused in): the weights STREAMED through a two-stage ring
// in chunks of 64 hidden columns as in mlp_stream.hpp, a wave = 32 tokens x all 1024 columns - 0.83 ms with the erf evaluation, 0.88
// with the table below, 0.49 ms without gp: ~4 us per (chunk, wave) whatever is inside, the barrier-per-chunk floor of that design;
// and this kernel WITH the LayerNorm recomputed by each of its eight column groups - 1.11 ms (1300 VALU instructions per tile for
// the 128 values of a lane, eight times).  What is left is WEIGHT-STATIONARY: a workgroup keeps 128 hidden columns (64 KiB) of W1
// in LDS for the whole launch and its waves stream token tiles through them exactly as lnlin_fwd_kernel does (no barrier, rows
// prefetched, branch-free buffer stores); the eight column groups of a token stream run side by side on one XCD, so the rows come
// from HBM once and seven times from that XCD's L2.  GELU and GELU' come from ONE 8-byte gather of the nearest-entry pair table.
template <class T, int C, int N> struct LinGeluWsGeom {
    static constexpr int KT = C / TileGeom<T>::BK, NG = 128, GROUPS = N / NG;      // hidden columns per workgroup; column groups
    static constexpr int W_BYTES = KT * NG * 128;
    static constexpr int OFF_LUT = W_BYTES + NG * 4;
    static constexpr int OFF_SCR = OFF_LUT + GELU_NLUT2_BYTES;
    template <int WPB> static constexpr int smem() { return OFF_SCR + WPB * LineBounce::BYTES; }
};

template <class T, int C, int N, int WPB, bool HAS_GP>
__global__ void __launch_bounds__(64 * WPB, 2)
lin_gelu_ws_kernel(const T* __restrict__ x, const T* __restrict__ W, const float* __restrict__ bias, T* __restrict__ g_out,
                   T* __restrict__ gp_out, int M) {
    typedef LinGeluWsGeom<T, C, N> G;
    constexpr int KS = C / 16, NB = G::NG / 32;
    constexpr int NPW = (G::W_BYTES / 1024) / WPB;
    static_assert(NPW * WPB * 1024 == G::W_BYTES && NB % 2 == 0, "weight pieces must divide over the waves");
    __shared__ __attribute__((aligned(16))) char smem[G::template smem<WPB>()];
    float* const kb = reinterpret_cast<float*>(smem + G::W_BYTES);
    float* const lut = reinterpret_cast<float*>(smem + G::OFF_LUT);
    const int tid = threadIdx.x, lane_ = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    // workgroup -> (column group, token stream): streams = grid / 8; with a multiple of 8 streams the 8 groups of stream s sit on
    // XCD s % 8 (workgroup id % 8 - the dispatcher deals workgroups round-robin over the XCDs)
    const int nstreams = (int)gridDim.x / G::GROUPS;
    int grp, stream;
    if (nstreams % 8 == 0) {
        const int j = (int)blockIdx.x >> 3;
        grp = j % G::GROUPS;
        stream = (j / G::GROUPS) * 8 + ((int)blockIdx.x & 7);
    } else {
        grp = (int)blockIdx.x % G::GROUPS;
        stream = (int)blockIdx.x / G::GROUPS;
    }
    {
        const pp_rsrc rw = pp_make_rsrc(W + (size_t)grp * G::NG * C, (unsigned)(G::NG * C * sizeof(T)));
        int v[NPW];
        ms_piece_offsets<T, NPW, WPB>(v, wave, lane_, G::NG, C);
#pragma unroll
        for (int i = 0; i < NPW; i++) pp_glds16(rw, smem, (wave + i * WPB) * 1024, v[i], 0);
    }
    for (int i = tid; i < G::NG; i += 64 * WPB) kb[i] = bias[grp * G::NG + i];
    gelu_nlut2_fill(lut, tid, 64 * WPB);
    pp_wait_vm<0>();
    __syncthreads();

    const int n_tiles = (M + 31) / 32;
    const int stride = nstreams * WPB;
    int tile = stream * WPB + wave;
    if (tile >= n_tiles) return;
    // (rows through buffer accesses with exact-range resources, everything lane-derived from an opaque lane id per tile: see lnlin_fwd_kernel)
    auto rows_of = [&](int t) { const int r = M - t * 32; return wave_uniform(r < 0 ? 0 : (r > 32 ? 32 : r)); };
    auto load_rows = [&](frag_t<T> (&f)[KS], int t, int lane) __attribute__((always_inline)) {
        const pp_rsrc rx = pp_make_rsrc(x + (size_t)t * 32 * C, (unsigned)(rows_of(t) * C * (int)sizeof(T)));
        const int off = (lane & 31) * C * (int)sizeof(T) + (lane >> 5) * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) f[ks] = __builtin_bit_cast(frag_t<T>, pp_load16(rx, off + 32 * ks));
    };
    frag_t<T> xn[KS];
    load_rows(xn, tile, lane_);
#ifndef RVT_EMU
#pragma unroll
    for (int ks = 0; ks < KS; ks++) asm volatile("" : "+v"(xn[ks]));
#endif
    for (; tile < n_tiles; tile += stride) {
        int lane = lane_;
        opaque_vgpr(lane);
        const int li = lane & 31, half = lane >> 5;
        const int rb = ms_rowbase_h<T>(li, half);
        frag_t<T> uf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) uf[ks] = xn[ks];
        load_rows(xn, tile + stride, lane);               // (beyond the last tile: an empty resource) - in front of every store of the tile
        sched_fence();
        const int nrows = rows_of(tile);
        const pp_rsrc rg = pp_make_rsrc(g_out + (size_t)tile * 32 * N, (unsigned)(nrows * N * (int)sizeof(T)));
        const pp_rsrc rp = pp_make_rsrc(gp_out + (size_t)tile * 32 * N, HAS_GP ? (unsigned)(nrows * N * (int)sizeof(T)) : 0u);
        LineBounce lb;
        lb.init(smem + G::OFF_SCR + wave * LineBounce::BYTES, lane);
#pragma unroll
        for (int pair = 0; pair < NB / 2; pair++) {
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; j++) acc_load_rows(acc[j], kb + 32 * (2 * pair + j), half);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                for (int j = 0; j < 2; j++) mma32(acc[j], ms_load_frag<T>(smem + (2 * pair + j) * 32 * 128, G::NG, rb, 2 * ks), uf[ks]);
                if ((ks & 3) == 3) sched_fence();         // (or hipcc requests all the weight fragments of the pair at once)
            }
            // piece m of block j = accumulator registers 8 m .. 8 m + 7 (acc_to_rows): columns 32 j + 16 m + 8 half .. + 7 of the pair
            u32x4 gpc[2][2], ppc[2][2];
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    float x8[8], g8[8], p8[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) x8[e] = acc[j][8 * m + e];
                    gelu_both_nlut2_8(lut, x8, g8, p8);
#pragma unroll
                    for (int w = 0; w < 4; w++) { swap32(g8[w], g8[4 + w]); swap32(p8[w], p8[4 + w]); }
                    gpc[j][m] = __builtin_bit_cast(u32x4, frag_from_float<T>(g8));
                    ppc[j][m] = __builtin_bit_cast(u32x4, frag_from_float<T>(p8));
                }
            lb.flush(rg, gpc, N * (int)sizeof(T), (grp * G::NG + 64 * pair) * (int)sizeof(T));
            if (HAS_GP) lb.flush(rp, ppc, N * (int)sizeof(T), (grp * G::NG + 64 * pair) * (int)sizeof(T));
            sched_fence();
        }
    }
}

}  // namespace rvt
