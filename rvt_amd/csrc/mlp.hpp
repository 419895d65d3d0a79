// Fused MLP half of a PartitionAttentionCl block (reference maxvit.py:269 with the MLP of :100-118):
//
//     xout = xmid + gamma2 * ( GELU( LN2(xmid) W1^T + b1 ) W2^T + b2 )
//
// Op-by-op this chain moves ~17 (forward) + ~18 (input-gradient path of backward) activation rows of C elements per
// token through HBM; at C <= 128 (RVT stages 1-2) every one of those kernels is HBM-bound.  The two kernels here
// keep the LayerNorm output and the data flow between the two linears on chip:
//
//   mlp_fwd_kernel       reads xmid, writes xout and (training) g = GELU(h), gp = GELU'(h)           1 + 8 + 1 rows
//   mlp_bwd_dgrad_kernel reads dxout, gp, xmid; writes dh (needed by the fc1 weight gradient) and
//                        dxmid = dxout + LN2'(dh W1)   — fc2 dgrad * gp, fc1 dgrad and LayerNorm backward in one pass
//
// (weight gradients stay with the split-K TN GEMMs, which read g / dh.)
//
// Structure per workgroup (256 threads = 4 waves as 2x2, persistent over 128-token tiles):
//   tile load in a (row, 8-channel chunk) thread layout -> LayerNorm statistics with wave shuffles -> swizzled LDS A
//   operand; per hidden chunk of JC columns: weight panels -> LDS, MFMA, accumulators -> fp32 LDS staging (64 rows per
//   pass) -> back to the (row, chunk) layout where bias/GELU/gp are applied on 16-byte vectors that go to HBM and, as
//   the next A operand, to LDS; second MFMA; final epilogue (LayerScale+residual / LayerNorm backward) in the same
//   (row, chunk) layout, with the raw input tile still in registers.
// JC = 128 (bf16) / 64 (f32) so that all operands fit the 160 KiB LDS.
#pragma once
#include "common.hpp"
#include "rowops.hpp"

namespace rvt {

// ---- LDS operand matrix: [rows][K] stored as K/BK sub-tiles of [rows][128 bytes], XOR-swizzled like GEMM tiles ----
template <class T> __device__ __forceinline__ char* opm_subtile(char* base, int rows, int kt) { return base + (size_t)kt * rows * 128; }

// store an 8-element fragment at (row, kcol0 = 8*fcg)
template <class T> __device__ __forceinline__ void opm_store_frag(char* base, int rows, int row, int fcg, const frag_t<T>& v) {
    constexpr int FPR = TileGeom<T>::FPR;
    tile_store_frag<T>(opm_subtile<T>(base, rows, fcg / FPR), row, fcg % FPR, v);
}

// address of element (row, kcol) of an LDS operand matrix [rows][K] (K-subtiles of [rows][128 B], swizzled 16-B chunks)
template <class T> __device__ __forceinline__ T* opm_elem_ptr(char* base, int rows, int row, int kcol) {
    constexpr int BK = TileGeom<T>::BK;
    const int kt = kcol / BK, kc = kcol % BK;
    const int byte = kc * (int)sizeof(T);
    return reinterpret_cast<T*>(base + (size_t)kt * rows * 128 + lds_chunk_off(row, byte >> 4) + (byte & 15));
}
template <class T> __device__ __forceinline__ frag_t<T> opm_load_frag(const char* base, int rows, int row, int fcg) {
    constexpr int FPR = TileGeom<T>::FPR;
    return tile_load_frag<T>(base + (size_t)(fcg / FPR) * rows * 128, row, fcg % FPR);
}

// LDS byte offset of element (R0 + rowc(r), ch) of a swizzled tile, r = accumulator register index of the 32x32 MFMA C/D
// layout, GIVEN the offset `base` of element (R0, ch) with R0 = 32*w + 4*(lane>>5): rowc(r) = (r&3) + 8*(r>>2) never
// carries out of the low five row bits, so the two swizzle terms of lds_chunk_off split into a per-lane part (already in
// `base`) XOR a function of r alone:  ((row>>1)&7) -> ((r>>1)&1) | ((r>>2)&1)<<2,  ((row>>4)&7) -> (r>>3).
// One v_xor plus an immediate offset per access instead of a dozen integer ops, and ONE register instead of sixteen.
__device__ __forceinline__ int acc_elem_off(int base, int r) {
    const int k = (((r >> 1) & 1) | (((r >> 2) & 1) << 2)) ^ ((r >> 3) & 1);
    return (base ^ (k << 4)) + ((r & 3) + 8 * (r >> 2)) * 128;
}

// stage a row-major global block [rows][kcols] (leading dimension ld elements) into an operand matrix; all 256 threads
template <class T> __device__ __forceinline__ void opm_stage(char* base, const T* g, int ld, int rows, int kcols, int tid) {
    const int fpr_g = kcols / 8;
    for (int f = tid; f < rows * fpr_g; f += 256) {
        const int row = f / fpr_g, fcg = f % fpr_g;
        opm_store_frag<T>(base, rows, row, fcg, frag_load<T>(g + (size_t)row * ld + fcg * 8));
    }
}

// acc[i][j] += A[a_row0 + 32 i + .][0..ktot) . B[b_row0 + 32 j + .][0..ktot)
template <class T, int MI, int NJ>
__device__ __forceinline__ void opm_mma(f32x16 (&acc)[MI][NJ], const char* A, int a_rows, int a_row0, const char* B, int b_rows,
                                        int b_row0, int ktot, int lane) {
    constexpr int BK = TileGeom<T>::BK;
    const int li = lane & 31, half = lane >> 5;
    for (int kt = 0; kt < ktot / BK; kt++) {
        const char* At = A + (size_t)kt * a_rows * 128;
        const char* Bt = B + (size_t)kt * b_rows * 128;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ks++) {
            const int fc = ks * 2 + half;
            frag_t<T> a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; i++) a[i] = tile_load_frag<T>(At, a_row0 + i * 32 + li, fc);
#pragma unroll
            for (int j = 0; j < NJ; j++) b[j] = tile_load_frag<T>(Bt, b_row0 + j * 32 + li, fc);
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) mma32(acc[i][j], a[i], b[j]);
        }
    }
}

// one 64-row pass of a 2x2-wave accumulator (wave tile 32*MI x 32*NJ) into fp32 staging [64][ld]:
// MFMA row block i of every wave: stage row wm*32+r  <->  tile row wm*32*MI+i*32+r
template <int MI, int NJ>
__device__ __forceinline__ void stage_pass(float* stage, int ld, const f32x16 (&acc)[MI][NJ], int i, int wm, int wn, int lane) {
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 16; r++)
            stage[(wm * 32 + acc_row(r, lane)) * ld + wn * (32 * NJ) + j * 32 + (lane & 31)] = acc[i][j][r];
}
template <int MI> __device__ __forceinline__ int stage_row_to_tile_row(int srow, int pass) {
    return (srow >> 5) * (32 * MI) + pass * 32 + (srow & 31);
}
__device__ __forceinline__ void stage_read8(const float* stage, int ld, int srow, int col0, float (&v)[8]) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(stage + srow * ld + col0 + h * 4);
        v[h * 4 + 0] = t[0]; v[h * 4 + 1] = t[1]; v[h * 4 + 2] = t[2]; v[h * 4 + 3] = t[3];
    }
}

#ifndef MLP_WAVES
#define MLP_WAVES 2
#endif
template <class T> struct MlpGeom { static constexpr int JC = 64; };   // 64-column hidden chunks: ~57 KiB LDS at C=64 -> two workgroups per CU

// TM = tokens per tile (64 or 128): 4 waves as 2x2, each wave TM/2 = 32*MI token rows.
template <class T, int C, int TM> struct MlpSmem {
    static constexpr int JC = MlpGeom<T>::JC;
    static constexpr int KT_C = C / TileGeom<T>::BK, KT_J = JC / TileGeom<T>::BK;
    static constexpr int A_X = KT_C * TM * 128;                        // [TM tokens][C]    (v2 in fwd, dxout in bwd)
    static constexpr int B_1 = KT_C * JC * 128;                        // [JC][C]           (W1_j / (W2 gamma)^T_j)
    static constexpr int STG_A = 64 * (JC + 4) * 4;                    // per-chunk fp32 staging, overlays B_1
    static constexpr int STG_B = 64 * (C + 4) * 4;                     // final staging: may run on into A_H / B_2 (dead by then)
    static constexpr int R1 = B_1 > STG_A ? B_1 : STG_A;
    static constexpr int A_H = KT_J * TM * 128;                        // [TM tokens][JC]   (g in fwd, dh in bwd)
    static constexpr int B_2 = KT_J * C * 128;                         // [C][JC]           (W2[:, j] / W1^T[:, j])
    static constexpr int OFF_1 = A_X, OFF_H = OFF_1 + R1, OFF_2 = OFF_H + A_H;
    static constexpr int OFF_K = OFF_2 + B_2;                         // per-channel constants (fp32): ln_w, ln_b, gamma, b2 [C] + b1 [4C]
    static constexpr int BYTES = OFF_K + 8 * C * 4;
    static_assert(BYTES <= 160 * 1024, "fused MLP tile does not fit the LDS");
    static_assert(STG_B <= OFF_K - OFF_1, "final staging does not fit behind the input tile");
    static_assert(256 * 16 * 4 <= BYTES, "reduction scratch");
};

// (row, 8-channel chunk) thread layout of a [TM][C] tile: slot q of thread tid is frag f = tid + 256 q, row f/G, chunk f%G
template <class T, int C, int TM> struct TileSlots {
    static constexpr int G = C / 8;
    static constexpr int NFX = TM * G / 256;
    static_assert(NFX >= 1, "tile too small for 256 threads");
};

// Global-memory schedule shared by both kernels (see the GEMM epilogue notes in gemm.hpp: loads and stores retire
// through ONE in-order counter, so a load issued after a store cannot be waited for without also waiting for that
// store's acknowledgement):
//   * everything a stretch of code reads from global memory is issued BEFORE the stores of the stretch preceding it:
//     the next tile's rows at the top of the current tile, the next hidden chunk's weight panels (and gp rows) at
//     the top of the current chunk — one full tile / chunk ahead of their use;
//   * per-thread column constants (LayerNorm weights, LayerScale, b2) live in registers for the whole launch;
//   * prefetched registers are moved to LDS at the BOTTOM of the loop that consumes them, where every path has
//     issued the same loads and stores, so the compiler's wait counts exclude the stores.
template <class T, int C, int TM> struct MlpPanels {       // register image of one hidden chunk's two weight panels
    static constexpr int JC = MlpGeom<T>::JC;
    static constexpr int F1 = C / 8, F2 = JC / 8;            // 16-byte frags per row of panel 1 [JC][C] / panel 2 [C][JC]
    static constexpr int N1 = JC * F1 / 256, N2 = C * F2 / 256;
    static_assert(N1 >= 1 && N2 >= 1 && (JC * F1) % 256 == 0 && (C * F2) % 256 == 0, "weight panel / thread mapping");
    frag_t<T> r1[N1], r2[N2];
    // panel 1 = rows j0.. of P1 ([4C][C] row-major), panel 2 = columns j0.. of P2 ([C][4C] row-major)
    __device__ __forceinline__ void load(const T* P1, const T* P2, int j0, int tid) {
#pragma unroll
        for (int i = 0; i < N1; i++) {
            const int f = tid + i * 256;
            r1[i] = frag_load<T>(P1 + (size_t)(j0 + f / F1) * C + (f % F1) * 8);
        }
#pragma unroll
        for (int i = 0; i < N2; i++) {
            const int f = tid + i * 256;
            r2[i] = frag_load<T>(P2 + (size_t)(f / F2) * (4 * C) + j0 + (f % F2) * 8);
        }
    }
    __device__ __forceinline__ void store(char* B1, char* B2, int tid) const {
#pragma unroll
        for (int i = 0; i < N1; i++) {
            const int f = tid + i * 256;
            opm_store_frag<T>(B1, JC, f / F1, f % F1, r1[i]);
        }
#pragma unroll
        for (int i = 0; i < N2; i++) {
            const int f = tid + i * 256;
            opm_store_frag<T>(B2, C, f / F2, f % F2, r2[i]);
        }
    }
};

template <int N> __device__ __forceinline__ void load_cols(const float* p, int c0, float (&v)[N]) {
#pragma unroll
    for (int q = 0; q < N / 4; q++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + c0 + q * 4);
        v[q * 4 + 0] = t[0]; v[q * 4 + 1] = t[1]; v[q * 4 + 2] = t[2]; v[q * 4 + 3] = t[3];
    }
}

// ===================================================================================================== forward
template <class T, int C, int TM>
__global__ void __launch_bounds__(256, (sizeof(T) == 2 && TM == 64) ? MLP_WAVES : 1)
mlp_fwd_kernel(const T* __restrict__ xmid, T* __restrict__ xout, T* __restrict__ g_out, T* __restrict__ gp_out,
               T* __restrict__ v2_out, const float* __restrict__ ln_w, const float* __restrict__ ln_b, const T* __restrict__ W1,
               const float* __restrict__ b1, const T* __restrict__ W2, const float* __restrict__ b2,
               const float* __restrict__ gamma, int M, float eps) {
    typedef MlpSmem<T, C, TM> S;
    constexpr int MI = TM / 64;
    constexpr int JC = S::JC, HID = 4 * C;
    constexpr int G = TileSlots<T, C, TM>::G, NFX = TileSlots<T, C, TM>::NFX;
    constexpr int NJ1 = JC / 64, NJ2 = C / 64;
    constexpr int LD1 = JC + 4, LD2 = C + 4;
    constexpr int UPR1 = JC / 8;                           // 8-column units per staged row (fc1 side)
    constexpr int UPT1 = 64 * UPR1 / 256;                  // units per thread per 64-row pass
    static_assert(256 % G == 0 && 256 % UPR1 == 0 && UPT1 >= 1, "a thread keeps one column chunk");
    __shared__ __attribute__((aligned(16))) char smem[S::BYTES];
    char* const Ax = smem;
    char* const B1 = smem + S::OFF_1;
    float* const stage = reinterpret_cast<float*>(smem + S::OFF_1);
    char* const Ah = smem + S::OFF_H;
    char* const B2 = smem + S::OFF_2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_tiles = (M + TM - 1) / TM;
    const int cl = tid % G;                                // this thread's 8-channel chunk in the (row, chunk) layout
    const int cu = tid % UPR1;                             // ... and its 8-column unit of a hidden chunk
    // per-channel constants live in LDS for the whole launch: reading them is an LDS access (lgkmcnt), which never has
    // to wait behind the global stores in flight
    float* const kst = reinterpret_cast<float*>(smem + S::OFF_K);
    const float* const k_lnw = kst, * const k_lnb = kst + C, * const k_gam = kst + 2 * C, * const k_b2 = kst + 3 * C;
    const float* const k_b1 = kst + 4 * C;
    for (int i = tid; i < C; i += 256) { kst[i] = ln_w[i]; kst[C + i] = ln_b[i]; kst[2 * C + i] = gamma[i]; kst[3 * C + i] = b2[i]; }
    for (int i = tid; i < HID; i += 256) kst[4 * C + i] = b1[i];
    __syncthreads();

    auto load_rows = [&](int tile, frag_t<T> (&r)[NFX]) {
#pragma unroll
        for (int q = 0; q < NFX; q++) {
            const int row = (tid + q * 256) / G;
            const bool ok = tile * TM + row < M;
            r[q] = frag_load<T>(xmid + (size_t)(ok ? tile * TM + row : 0) * C + cl * 8);
        }
    };
    // LayerNorm (maxvit.py:241) of a tile held in registers -> A operand of fc1
    auto norm_to_lds = [&](int tile, const frag_t<T> (&r)[NFX]) {
#pragma unroll
        for (int q = 0; q < NFX; q++) {
            const int row = (tid + q * 256) / G;
            const bool ok = tile * TM + row < M;
            float v[8];
            frag_to_float<T>(r[q], v);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) s += v[e];
            const float mean = group_sum(s, G) / (float)C;
            float qq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = v[e] - mean; qq += d * d; }
            const float rstd = 1.0f / sqrtf(group_sum(qq, G) / (float)C + eps);
            float o[8], lnw[8], lnb[8];
            load_cols<8>(k_lnw, cl * 8, lnw); load_cols<8>(k_lnb, cl * 8, lnb);
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = ok ? (v[e] - mean) * rstd * lnw[e] + lnb[e] : 0.f;
            const frag_t<T> of = frag_from_float<T>(o);
            opm_store_frag<T>(Ax, TM, row, cl, of);
            if (v2_out != nullptr && ok)      // LN2(xmid): the B operand of the fc1 weight gradient (saved, not recomputed)
                frag_store<T>(v2_out + (size_t)(tile * TM + row) * C + cl * 8, of);
        }
    };

    int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    frag_t<T> raw[NFX], nxt[NFX];
    MlpPanels<T, C, TM> wp;
    load_rows(tile, raw);
    wp.load(W1, W2, 0, tid);
    norm_to_lds(tile, raw);
    wp.store(B1, B2, tid);

    for (; tile < n_tiles; tile += gridDim.x) {
        const int m0 = tile * TM;
        const int tile2 = tile + gridDim.x;
        const bool have2 = tile2 < n_tiles;
        if (have2) load_rows(tile2, nxt);

        f32x16 acc2[MI][NJ2];
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < NJ2; j++) acc_zero(acc2[i][j]);

        for (int j0 = 0; j0 < HID; j0 += JC) {
            const bool last = j0 + JC >= HID;
            lds_barrier();                                                     // this chunk's panels (and Ax) are in LDS
            if (!last) wp.load(W1, W2, j0 + JC, tid);
            else if (have2) wp.load(W1, W2, 0, tid);
            float b1v[8];
            load_cols<8>(k_b1, j0 + cu * 8, b1v);
            sched_fence();
            f32x16 acc1[MI][NJ1];
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ1; j++) acc_zero(acc1[i][j]);
            opm_mma<T, MI, NJ1>(acc1, Ax, TM, wm * 32 * MI, B1, JC, wn * (JC / 2), C, lane);
            lds_barrier();                                                     // B1 consumed -> staging may overlay it
#pragma unroll
            for (int i = 0; i < MI; i++) {
                if (i) lds_barrier();
                stage_pass<MI, NJ1>(stage, LD1, acc1, i, wm, wn, lane);
                lds_barrier();
#pragma unroll
                for (int q = 0; q < UPT1; q++) {
                    const int srow = tid / UPR1 + q * (256 / UPR1);
                    const int row = stage_row_to_tile_row<MI>(srow, i);
                    float v[8], a[8], b[8];
                    stage_read8(stage, LD1, srow, cu * 8, v);
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] += b1v[e];
                    gelu_both_8(v, a, b);
                    const frag_t<T> gf = frag_from_float<T>(a);
                    opm_store_frag<T>(Ah, TM, row, cu, gf);
                    if (g_out != nullptr && m0 + row < M) {
                        const size_t o = (size_t)(m0 + row) * HID + j0 + cu * 8;
                        if (gp_out != nullptr) {
                            frag_store<T>(g_out + o, gf);
                            frag_store<T>(gp_out + o, frag_from_float<T>(b));
                        } else {                      // pre-activation only (round 4): half the 4C-wide bytes; the backward applies GELU /
                            frag_store<T>(g_out + o, frag_from_float<T>(v));   // GELU' on load (rvt_linear_wgrad gelu_in, rvt_linear_dgrad gelu_pre)
                        }
                    }
                }
            }
            lds_barrier();
            opm_mma<T, MI, NJ2>(acc2, Ah, TM, wm * 32 * MI, B2, C, wn * (C / 2), JC, lane);
            lds_barrier();                                                     // Ah / B2 / staging free for the next chunk
            if (!last) wp.store(B1, B2, tid);
        }

        // ---- LayerScale + residual (maxvit.py:51-53,269) in the load layout: the residual is still in registers ----
#pragma unroll
        for (int i = 0; i < MI; i++) {
            if (i) lds_barrier();
            stage_pass<MI, NJ2>(stage, LD2, acc2, i, wm, wn, lane);
            lds_barrier();
#pragma unroll
            for (int q = 0; q < NFX; q++) {
                const int row = (tid + q * 256) / G;
                if (((row >> 5) % MI) != i) continue;
                if (m0 + row < M) {
                    float v[8], res[8], gam[8], b2v[8];
                    load_cols<8>(k_gam, cl * 8, gam); load_cols<8>(k_b2, cl * 8, b2v);
                    frag_to_float<T>(raw[q], res);
                    stage_read8(stage, LD2, (row / (32 * MI)) * 32 + (row & 31), cl * 8, v);
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = res[e] + gam[e] * (v[e] + b2v[e]);
                    frag_store<T>(xout + (size_t)(m0 + row) * C + cl * 8, frag_from_float<T>(v));
                }
            }
        }
        lds_barrier();                                                         // staging (over B1 .. B2) is free again
        if (have2) {
            wp.store(B1, B2, tid);
#pragma unroll
            for (int q = 0; q < NFX; q++) raw[q] = nxt[q];
            norm_to_lds(tile2, raw);
        }
    }
}

// ================================================================================ backward: input-gradient chain
// dh[m][4C]   = (dxout W2g)[m][:] * gp[m][:]              W2g^T = (W2 * gamma)^T stored [4C][C]  ("fc2_wt")
// dv2[m][C]   = dh W1                                      W1^T stored [C][4C]                    ("fc1_wt")
// dxmid       = dxout + LN2'(dv2; xmid)                    dln_w += dv2 * xhat, dln_b += dv2
template <class T, int C, int TM>
__global__ void __launch_bounds__(256, (sizeof(T) == 2 && C == 64 && TM == 64) ? MLP_WAVES : 1)
mlp_bwd_dgrad_kernel(const T* __restrict__ dxout, const T* __restrict__ gp, const T* __restrict__ xmid, T* __restrict__ dh,
                     T* __restrict__ dxmid, const float* __restrict__ ln_w, const T* __restrict__ W2gT,
                     const T* __restrict__ W1T, float* __restrict__ dln_w, float* __restrict__ dln_b, int M, float eps) {
    typedef MlpSmem<T, C, TM> S;
    constexpr int MI = TM / 64;
    constexpr int JC = S::JC, HID = 4 * C;
    constexpr int G = TileSlots<T, C, TM>::G, NFX = TileSlots<T, C, TM>::NFX;
    constexpr int NJ1 = JC / 64, NJ2 = C / 64;
    constexpr int LD1 = JC + 4, LD2 = C + 4;
    constexpr int UPR1 = JC / 8;
    constexpr int UPT1 = 64 * UPR1 / 256;
    static_assert(256 % G == 0 && 256 % UPR1 == 0 && UPT1 >= 1, "a thread keeps one column chunk");
    __shared__ __attribute__((aligned(16))) char smem[S::BYTES];
    char* const Ax = smem;                                  // dxout tile
    char* const B1 = smem + S::OFF_1;                       // W2g^T rows j0..: [JC][C]
    float* const stage = reinterpret_cast<float*>(smem + S::OFF_1);
    char* const Ah = smem + S::OFF_H;                       // dh chunk
    char* const B2 = smem + S::OFF_2;                       // W1^T[:, j0..]: [C][JC]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_tiles = (M + TM - 1) / TM;
    const int cl = tid % G;                                 // every slot of this thread has the same channel chunk
    const int cu = tid % UPR1;
    float aw[8], ab[8];
    float* const k_lnw = reinterpret_cast<float*>(smem + S::OFF_K);   // LDS-resident constants, see the forward kernel
    for (int i = tid; i < C; i += 256) k_lnw[i] = ln_w[i];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; e++) { aw[e] = 0.f; ab[e] = 0.f; }

    auto load_rows = [&](int tile, frag_t<T> (&rdx)[NFX], frag_t<T> (&rx)[NFX]) {
#pragma unroll
        for (int q = 0; q < NFX; q++) {
            const int row = (tid + q * 256) / G;
            const bool ok = tile * TM + row < M;
            const size_t o = (size_t)(ok ? tile * TM + row : 0) * C + cl * 8;
            rdx[q] = frag_load<T>(dxout + o);
            rx[q] = frag_load<T>(xmid + o);
        }
    };
    // gp rows of hidden chunk j0 this thread multiplies with: unit (pass i, slot q)
    auto load_gp = [&](int tile, int j0, frag_t<T> (&r)[MI][UPT1]) {
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int q = 0; q < UPT1; q++) {
                const int row = stage_row_to_tile_row<MI>(tid / UPR1 + q * (256 / UPR1), i);
                const bool ok = tile * TM + row < M;
                r[i][q] = frag_load<T>(gp + (size_t)(ok ? tile * TM + row : 0) * HID + j0 + cu * 8);
            }
    };
    float mean[NFX], rstd[NFX];
    // dxout tile -> A operand of the fc2 input gradient; LayerNorm statistics of the xmid tile
    auto stage_tile = [&](int tile, const frag_t<T> (&rdx)[NFX], const frag_t<T> (&rx)[NFX]) {
#pragma unroll
        for (int q = 0; q < NFX; q++) {
            const int row = (tid + q * 256) / G;
            const bool ok = tile * TM + row < M;
            const frag_t<T> z = frag_zero<T>();
            opm_store_frag<T>(Ax, TM, row, cl, ok ? rdx[q] : z);
            float v[8];
            frag_to_float<T>(rx[q], v);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) s += v[e];
            mean[q] = group_sum(s, G) / (float)C;
            float qq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = v[e] - mean[q]; qq += d * d; }
            rstd[q] = 1.0f / sqrtf(group_sum(qq, G) / (float)C + eps);
        }
    };

    int tile = blockIdx.x;
    if (tile < n_tiles) {
        frag_t<T> rawdx[NFX], rawx[NFX], ndx[NFX], nx[NFX], gpr[MI][UPT1];
        MlpPanels<T, C, TM> wp;
        load_rows(tile, rawdx, rawx);
        wp.load(W2gT, W1T, 0, tid);
        load_gp(tile, 0, gpr);
        stage_tile(tile, rawdx, rawx);
        wp.store(B1, B2, tid);

        for (; tile < n_tiles; tile += gridDim.x) {
            const int m0 = tile * TM;
            const int tile2 = tile + gridDim.x;
            const bool have2 = tile2 < n_tiles;
            if (have2) load_rows(tile2, ndx, nx);

            f32x16 acc2[MI][NJ2];
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ2; j++) acc_zero(acc2[i][j]);

            for (int j0 = 0; j0 < HID; j0 += JC) {
                const bool last = j0 + JC >= HID;
                lds_barrier();
                if (!last) wp.load(W2gT, W1T, j0 + JC, tid);
                else if (have2) wp.load(W2gT, W1T, 0, tid);
                sched_fence();
                f32x16 acc1[MI][NJ1];
#pragma unroll
                for (int i = 0; i < MI; i++)
#pragma unroll
                    for (int j = 0; j < NJ1; j++) acc_zero(acc1[i][j]);
                opm_mma<T, MI, NJ1>(acc1, Ax, TM, wm * 32 * MI, B1, JC, wn * (JC / 2), C, lane);
                lds_barrier();
#pragma unroll
                for (int i = 0; i < MI; i++) {
                    if (i) lds_barrier();
                    stage_pass<MI, NJ1>(stage, LD1, acc1, i, wm, wn, lane);
                    lds_barrier();
                    frag_t<T> df[UPT1];
#pragma unroll
                    for (int q = 0; q < UPT1; q++) {
                        const int srow = tid / UPR1 + q * (256 / UPR1);
                        const int row = stage_row_to_tile_row<MI>(srow, i);
                        const bool ok = m0 + row < M;
                        float v[8], p[8];
                        stage_read8(stage, LD1, srow, cu * 8, v);
                        frag_to_float<T>(gpr[i][q], p);
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] = ok ? v[e] * p[e] : 0.f;
                        df[q] = frag_from_float<T>(v);
                    }
                    // this pass's gp registers are consumed: refill them for the next chunk BEFORE the dh stores go out
#pragma unroll
                    for (int q = 0; q < UPT1; q++) {
                        const int row = stage_row_to_tile_row<MI>(tid / UPR1 + q * (256 / UPR1), i);
                        const int t2 = last ? tile2 : tile;
                        const int jn = last ? 0 : j0 + JC;
                        const bool ok2 = (!last || have2) && (t2 * TM + row < M);
                        gpr[i][q] = frag_load<T>(gp + (size_t)(ok2 ? t2 * TM + row : 0) * HID + jn + cu * 8);
                    }
                    sched_fence();
#pragma unroll
                    for (int q = 0; q < UPT1; q++) {
                        const int srow = tid / UPR1 + q * (256 / UPR1);
                        const int row = stage_row_to_tile_row<MI>(srow, i);
                        opm_store_frag<T>(Ah, TM, row, cu, df[q]);
                        if (m0 + row < M) frag_store<T>(dh + (size_t)(m0 + row) * HID + j0 + cu * 8, df[q]);
                    }
                }
                lds_barrier();
                opm_mma<T, MI, NJ2>(acc2, Ah, TM, wm * 32 * MI, B2, C, wn * (C / 2), JC, lane);
                lds_barrier();
                if (!last) wp.store(B1, B2, tid);
            }

            // ---- LayerNorm backward + residual, in the load layout (all lanes of a row group take part in the shuffles) ----
#pragma unroll
            for (int i = 0; i < MI; i++) {
                if (i) lds_barrier();
                stage_pass<MI, NJ2>(stage, LD2, acc2, i, wm, wn, lane);
                lds_barrier();
#pragma unroll
                for (int q = 0; q < NFX; q++) {
                    const int row = (tid + q * 256) / G;
                    const bool mine = ((row >> 5) % MI) == i;         // uniform over the G lanes of a row
                    const bool ok = mine && (m0 + row < M);
                    float d[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, xv[8], dxv[8], xh[8], lnw[8];
                    load_cols<8>(k_lnw, cl * 8, lnw);
                    if (mine) stage_read8(stage, LD2, (row / (32 * MI)) * 32 + (row & 31), cl * 8, d);
                    frag_to_float<T>(rawx[q], xv);
                    frag_to_float<T>(rawdx[q], dxv);
                    float gsum = 0.f, gxsum = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        d[e] = ok ? d[e] : 0.f;
                        xh[e] = ok ? (xv[e] - mean[q]) * rstd[q] : 0.f;
                        const float g_ = d[e] * lnw[e];
                        gsum += g_; gxsum += g_ * xh[e];
                        aw[e] += d[e] * xh[e]; ab[e] += d[e];
                    }
                    const float m1 = group_sum(gsum, G) / (float)C;
                    const float m2 = group_sum(gxsum, G) / (float)C;
                    if (ok) {
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) o[e] = dxv[e] + rstd[q] * (d[e] * lnw[e] - m1 - xh[e] * m2);
                        frag_store<T>(dxmid + (size_t)(m0 + row) * C + cl * 8, frag_from_float<T>(o));
                    }
                }
            }
            lds_barrier();
            if (have2) {
                wp.store(B1, B2, tid);
#pragma unroll
                for (int q = 0; q < NFX; q++) { rawdx[q] = ndx[q]; rawx[q] = nx[q]; }
                stage_tile(tile2, rawdx, rawx);
            }
        }
    }

    // LayerNorm parameter gradients: fold the 256/G threads that own the same channel chunk, one atomic per channel per WG
    lds_barrier();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; e++) { red[tid * 16 + e] = aw[e]; red[tid * 16 + 8 + e] = ab[e]; }
    lds_barrier();
    if (tid < G) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float sw = 0.f, sb = 0.f;
            for (int t = tid; t < 256; t += G) { sw += red[t * 16 + e]; sb += red[t * 16 + 8 + e]; }
            atomicAdd(dln_w + cl * 8 + e, sw);
            atomicAdd(dln_b + cl * 8 + e, sb);
        }
    }
}


// ====================================================================== backward, everything on chip (C = 64)
// mlp_bwd_fused_kernel: the whole backward of the MLP half from (dxout, xmid) alone.  Nothing but the block input was
// kept by the forward: LN2, fc1 and GELU / GELU' are RECOMPUTED per hidden chunk, the two input-gradient products and
// the LayerNorm backward follow as in mlp_bwd_dgrad_kernel, and the weight gradients
//     S2[c][j]  = sum_tok dxout[tok][c] g[tok][j]        (raw: LayerScale is folded in afterwards, pack.hpp)
//     dW1[j][c] = sum_tok dh[tok][j] v2[tok][c],   db1[j] = sum_tok dh[tok][j],   cs2[c] = sum_tok dxout[tok][c]
// are accumulated IN REGISTERS across all the tiles a persistent workgroup walks (the contraction index is the token:
// both operands come out of the row-major LDS tiles through the transpose read) and leave as one partial per workgroup.
// HBM traffic per token: read dxout + xmid, write dxmid — 3 rows of C instead of 5 + 11 + 5 + 5 for the separate
// input-gradient kernel and the two weight-gradient GEMMs (which re-read g, dh, v2 that the forward had to store).
// Accumulator-layout trick: fc1 (recomputed) and the fc2 input gradient use the SAME wave tiling, so h and dg of a
// (token, hidden column) meet in one lane at one accumulator index: GELU' * dg needs no staging, g and dh go to LDS
// as the next products' operands with 2-byte stores at analytic offsets (acc_elem_off).
template <class T, int C> struct MlpFusedSmem {
    static constexpr int TM = 64, JC = 64;
    static constexpr int KT_C = C / TileGeom<T>::BK, KT_J = JC / TileGeom<T>::BK;
    static constexpr int T_C = KT_C * TM * 128;           // [TM][C]
    static constexpr int T_J = KT_J * TM * 128;           // [TM][JC]
    static constexpr int P_1 = KT_C * JC * 128;           // [JC][C] weight panel
    static constexpr int P_2 = KT_J * C * 128;            // [C][JC] weight panel
    static constexpr int OFF_AX = 0, OFF_AV = T_C, OFF_B0 = 2 * T_C, OFF_B1 = OFF_B0 + P_1, OFF_B2 = OFF_B1 + P_1;
    static constexpr int OFF_AG = OFF_B2 + P_2, OFF_AH = OFF_AG + T_J, OFF_K = OFF_AH + T_J;
    static constexpr int STG = 64 * (C + 4) * 4;          // final fp32 staging, overlays the three weight panels
    static_assert(STG <= 2 * P_1 + P_2 + 2 * T_J, "staging overlay");
    static constexpr int BYTES = OFF_K + (2 * C + 4 * C) * 4;
    static_assert(BYTES <= 160 * 1024, "LDS");
};

// MODE 0: everything in one kernel (input gradient + all weight gradients: 128 accumulator registers, one workgroup per CU)
// MODE 1: input-gradient half only (dxmid, LayerNorm parameter gradients): no accumulators, several workgroups per CU —
//         the kernel on the critical path of backward
// MODE 2: weight-gradient half only, blockIdx.y = group of NCH/NG hidden chunks (64 accumulator registers per group with
//         NG = 2): re-reads (dxout, xmid) once per group, for the weight-gradient stream
template <class T, int C, int MODE>
__global__ void __launch_bounds__(256, (MODE == 0 || sizeof(T) == 4) ? 1 : 2)
mlp_bwd_fused_kernel(const T* __restrict__ dxout, const T* __restrict__ xmid, T* __restrict__ dxmid,
                     const float* __restrict__ ln_w, const float* __restrict__ ln_b, const T* __restrict__ W1,
                     const float* __restrict__ b1, const T* __restrict__ W2gT, const T* __restrict__ W1T,
                     float* __restrict__ dln_w, float* __restrict__ dln_b, float* __restrict__ ws, int M, float eps) {
    typedef MlpFusedSmem<T, C> S;
    constexpr int TM = S::TM, JC = S::JC, HID = 4 * C, NCH = HID / JC;
    constexpr bool DGRAD = MODE != 2, WGRAD = MODE != 1;
    constexpr int NG = MODE == 2 ? 2 : 1;                  // chunk groups (blockIdx.y)
    constexpr int NCG = NCH / NG;                          // hidden chunks this workgroup walks
    const int ch_base = MODE == 2 ? (int)blockIdx.y * NCG : 0;
    constexpr int G = C / 8, NFX = TM * G / 256;
    constexpr int F1 = C / 8, F2 = JC / 8, N1 = JC * F1 / 256, N2 = C * F2 / 256;
    constexpr int LD2 = C + 4;
    static_assert(C == 64 && JC == 64 && TM == 64, "wave tiling below is 2x2 waves of 32x32 for 64x64 products");
    __shared__ __attribute__((aligned(16))) char smem[S::BYTES];
    char* const Ax = smem + S::OFF_AX;      // dxout tile
    char* const Av = smem + S::OFF_AV;      // LN2(xmid) tile
    char* const B0 = smem + S::OFF_B0;      // W1 rows j0..        [JC][C]   (fc1 recompute)
    char* const B1 = smem + S::OFF_B1;      // (W2 gamma)^T rows j0.. [JC][C] (fc2 input gradient)
    char* const B2 = smem + S::OFF_B2;      // W1^T[:, j0..]       [C][JC]   (fc1 input gradient)
    char* const Ag = smem + S::OFF_AG;      // g = GELU(h) chunk
    char* const Ah = smem + S::OFF_AH;      // dh chunk
    float* const stage = reinterpret_cast<float*>(smem + S::OFF_B0);
    float* const kst = reinterpret_cast<float*>(smem + S::OFF_K);
    const float* const k_lnw = kst, * const k_lnb = kst + C, * const k_b1 = kst + 2 * C;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int n_tiles = (M + TM - 1) / TM;
    const int cl = tid % G;
    for (int i = tid; i < C; i += 256) { kst[i] = ln_w[i]; kst[C + i] = ln_b[i]; }
    for (int i = tid; i < HID; i += 256) kst[2 * C + i] = b1[i];
    __syncthreads();

    // persistent accumulators (whole launch): weight gradients per hidden chunk in the MFMA C/D layout, db1 per lane
    // column, LayerNorm parameter gradients and cs2 per (thread, 8-channel chunk)
    constexpr int NACC = WGRAD ? NCG : 1;
    f32x16 dW2acc[NACC], dW1acc[NACC];
    float db1acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; c++) { acc_zero(dW2acc[c]); acc_zero(dW1acc[c]); db1acc[c] = 0.f; }
    float aw[8], ab[8], acs[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { aw[e] = 0.f; ab[e] = 0.f; acs[e] = 0.f; }

    // accumulator-layout LDS offset of element (wm*32 + 4*half, wn*32 + li) of a [TM][JC] tile, and the transposed-fragment
    // address constants of the feature blocks this wave contracts over tokens
    const int off0 = (int)(reinterpret_cast<char*>(opm_elem_ptr<T>(Ag, TM, wm * 32 + 4 * half, wn * 32 + li)) - Ag);
    TrFeat<T> trm, trn;
    trm.init(wm * 32, TM, lane);
    trn.init(wn * 32, TM, lane);

    auto load_rows = [&](int tile, frag_t<T> (&rdx)[NFX], frag_t<T> (&rx)[NFX]) {
#pragma unroll
        for (int q = 0; q < NFX; q++) {
            const int row = (tid + q * 256) / G;
            const bool ok = tile * TM + row < M;
            const size_t o = (size_t)(ok ? tile * TM + row : 0) * C + cl * 8;
            rdx[q] = frag_load<T>(dxout + o);
            rx[q] = frag_load<T>(xmid + o);
        }
    };
    float mean[NFX], rstd[NFX];
    // dxout tile -> A operand; LayerNorm of the xmid tile -> A operand of the fc1 recompute; column sums of dxout
    auto stage_tile = [&](int tile, const frag_t<T> (&rdx)[NFX], const frag_t<T> (&rx)[NFX]) {
#pragma unroll
        for (int q = 0; q < NFX; q++) {
            const int row = (tid + q * 256) / G;
            const bool ok = tile * TM + row < M;
            const frag_t<T> z = frag_zero<T>();
            opm_store_frag<T>(Ax, TM, row, cl, ok ? rdx[q] : z);
            float v[8], d[8];
            frag_to_float<T>(rx[q], v);
            frag_to_float<T>(rdx[q], d);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) { s += v[e]; if (WGRAD) acs[e] += ok ? d[e] : 0.f; }
            mean[q] = group_sum(s, G) / (float)C;
            float qq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) { const float dd = v[e] - mean[q]; qq += dd * dd; }
            rstd[q] = 1.0f / sqrtf(group_sum(qq, G) / (float)C + eps);
            float o[8], lnw[8], lnb[8];
            load_cols<8>(k_lnw, cl * 8, lnw); load_cols<8>(k_lnb, cl * 8, lnb);
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = ok ? (v[e] - mean[q]) * rstd[q] * lnw[e] + lnb[e] : 0.f;
            opm_store_frag<T>(Av, TM, row, cl, frag_from_float<T>(o));
        }
    };
    struct Panels {                       // register image of one hidden chunk's three weight panels
        frag_t<T> r0[N1], r1[N1], r2[N2];
    };
    auto load_panels = [&](Panels& p, int j0) {
#pragma unroll
        for (int i = 0; i < N1; i++) {
            const int f = tid + i * 256;
            p.r0[i] = frag_load<T>(W1 + (size_t)(j0 + f / F1) * C + (f % F1) * 8);
            p.r1[i] = frag_load<T>(W2gT + (size_t)(j0 + f / F1) * C + (f % F1) * 8);
        }
        if (DGRAD) {
#pragma unroll
            for (int i = 0; i < N2; i++) {
                const int f = tid + i * 256;
                p.r2[i] = frag_load<T>(W1T + (size_t)(f / F2) * HID + j0 + (f % F2) * 8);
            }
        }
    };
    auto store_panels = [&](const Panels& p) {
#pragma unroll
        for (int i = 0; i < N1; i++) {
            const int f = tid + i * 256;
            opm_store_frag<T>(B0, JC, f / F1, f % F1, p.r0[i]);
            opm_store_frag<T>(B1, JC, f / F1, f % F1, p.r1[i]);
        }
        if (DGRAD) {
#pragma unroll
            for (int i = 0; i < N2; i++) {
                const int f = tid + i * 256;
                opm_store_frag<T>(B2, C, f / F2, f % F2, p.r2[i]);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < n_tiles) {
        frag_t<T> rawdx[NFX], rawx[NFX], ndx[NFX], nx[NFX];
        Panels wp;
        load_rows(tile, rawdx, rawx);
        load_panels(wp, ch_base * JC);
        stage_tile(tile, rawdx, rawx);
        store_panels(wp);

        for (; tile < n_tiles; tile += gridDim.x) {
            const int m0 = tile * TM;
            const int tile2 = tile + gridDim.x;
            const bool have2 = tile2 < n_tiles;
            constexpr bool PREFETCH_ROWS = sizeof(T) == 2;             // (f32 parity mode: registers)
            if (PREFETCH_ROWS && have2) load_rows(tile2, ndx, nx);

            f32x16 acc2[1][1];
            acc_zero(acc2[0][0]);
#pragma unroll
            for (int ch = 0; ch < NCG; ch++) {
                const int j0 = (ch_base + ch) * JC;
                const bool last = ch == NCG - 1;
                lds_barrier();                                         // this chunk's panels and the tiles are in LDS
                if (!last) load_panels(wp, j0 + JC);
                else if (have2) load_panels(wp, ch_base * JC);
                const float b1c = k_b1[j0 + wn * 32 + li];
                sched_fence();
                // h = v2 W1_j^T (recomputed) and dg = dxout (W2 gamma)_j: same tiling -> same lane, same register
                f32x16 acch[1][1], acc1[1][1];
                acc_zero(acch[0][0]); acc_zero(acc1[0][0]);
                opm_mma<T, 1, 1>(acch, Av, TM, wm * 32, B0, JC, wn * 32, C, lane);
                opm_mma<T, 1, 1>(acc1, Ax, TM, wm * 32, B1, JC, wn * 32, C, lane);
                float colsum = 0.f;
#pragma unroll
                for (int hb = 0; hb < 2; hb++) {
                    float hv[8], gv[8], gpv[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) hv[e] = acch[0][0][hb * 8 + e] + b1c;
                    gelu_both_8(hv, gv, gpv);
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int r = hb * 8 + e;
                        const float dh = acc1[0][0][r] * gpv[e];
                        colsum += dh;
                        const int o = acc_elem_off(off0, r);
                        if (WGRAD) *reinterpret_cast<T*>(Ag + o) = (T)gv[e];
                        *reinterpret_cast<T*>(Ah + o) = (T)dh;
                    }
                }
                if (WGRAD) db1acc[ch] += colsum;
                lds_barrier();                                         // g / dh chunk complete; B0 / B1 consumed
                if (DGRAD) opm_mma<T, 1, 1>(acc2, Ah, TM, wm * 32, B2, C, wn * 32, JC, lane);
                // weight gradients: contraction over the TM tokens of the tile, operands transposed on the way out of LDS
                if (WGRAD) {
#pragma unroll
                    for (int k0 = 0; k0 < TM; k0 += 16) {
                        mma32(dW2acc[ch], trm.load(Ax, k0, lane), trn.load(Ag, k0, lane));       // rows c, columns j
                        mma32(dW1acc[ch], trm.load(Ah, k0, lane), trn.load(Av, k0, lane));       // rows j, columns c
                    }
                }
                lds_barrier();                                         // Ag / Ah / B2 free for the next chunk
                if (!last) store_panels(wp);
                sched_fence();            // keep the unrolled chunks apart: interleaving them only multiplies live registers
            }

            if (DGRAD) {
                // ---- LayerNorm backward + residual, in the load layout (all lanes of a row group take part in the shuffles) ----
                stage_pass<1, 1>(stage, LD2, acc2, 0, wm, wn, lane);
                lds_barrier();
    #pragma unroll
                for (int q = 0; q < NFX; q++) {
                    const int row = (tid + q * 256) / G;
                    const bool ok = m0 + row < M;
                    float d[8], xv[8], dxv[8], xh[8], lnw[8];
                    load_cols<8>(k_lnw, cl * 8, lnw);
                    stage_read8(stage, LD2, row, cl * 8, d);
                    frag_to_float<T>(rawx[q], xv);
                    frag_to_float<T>(rawdx[q], dxv);
                    float gsum = 0.f, gxsum = 0.f;
    #pragma unroll
                    for (int e = 0; e < 8; e++) {
                        d[e] = ok ? d[e] : 0.f;
                        xh[e] = ok ? (xv[e] - mean[q]) * rstd[q] : 0.f;
                        const float g_ = d[e] * lnw[e];
                        gsum += g_; gxsum += g_ * xh[e];
                        aw[e] += d[e] * xh[e]; ab[e] += d[e];
                    }
                    const float m1 = group_sum(gsum, G) / (float)C;
                    const float m2 = group_sum(gxsum, G) / (float)C;
                    if (ok) {
                        float o[8];
    #pragma unroll
                        for (int e = 0; e < 8; e++) o[e] = dxv[e] + rstd[q] * (d[e] * lnw[e] - m1 - xh[e] * m2);
                        frag_store<T>(dxmid + (size_t)(m0 + row) * C + cl * 8, frag_from_float<T>(o));
                    }
                }
            }
            lds_barrier();
            if (have2) {
                store_panels(wp);
                if (PREFETCH_ROWS) {
#pragma unroll
                    for (int q = 0; q < NFX; q++) { rawdx[q] = ndx[q]; rawx[q] = nx[q]; }
                } else {
                    load_rows(tile2, rawdx, rawx);
                }
                stage_tile(tile2, rawdx, rawx);
            }
        }
    }

    // ---- partial results of this workgroup: ws = [dW1: grid x 4C x C][S2: grid x C x 4C][db1: 2 grid x 4C][cs2: grid x C] ----
    // (MODE 2: every chunk group of a column of workgroups writes its own rows of the same per-workgroup record)
    if (WGRAD) {
        const size_t nwg = gridDim.x, wg = blockIdx.x;
        float* const p_dw1 = ws + wg * (size_t)(HID * C);
        float* const p_s2 = ws + nwg * (size_t)(HID * C) + wg * (size_t)(C * HID);
        float* const p_db1 = ws + 2 * nwg * (size_t)(HID * C) + (wg * 2 + wm) * (size_t)HID;
#pragma unroll
        for (int ch = 0; ch < NACC; ch++) {
            const int jb = (ch_base + ch) * JC;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int rr = wm * 32 + acc_row(r, lane), cc = wn * 32 + li;
                p_dw1[(size_t)(jb + rr) * C + cc] = dW1acc[ch][r];            // rows j, columns c
                p_s2[(size_t)rr * HID + jb + cc] = dW2acc[ch][r];            // rows c, columns j
            }
            const float v = db1acc[ch] + __shfl_xor(db1acc[ch], 32);
            if (half == 0) p_db1[jb + wn * 32 + li] = v;
        }
    }
    // LayerNorm parameter gradients and cs2: fold the 256/G threads that own the same channel chunk
    lds_barrier();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; e++) { red[tid * 24 + e] = aw[e]; red[tid * 24 + 8 + e] = ab[e]; red[tid * 24 + 16 + e] = acs[e]; }
    lds_barrier();
    if (tid < G) {
        float* const p_cs2 = ws + 2 * (size_t)gridDim.x * (HID * C) + 2 * (size_t)gridDim.x * HID + (size_t)blockIdx.x * C;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float sw = 0.f, sb = 0.f, sc = 0.f;
            for (int t = tid; t < 256; t += G) { sw += red[t * 24 + e]; sb += red[t * 24 + 8 + e]; sc += red[t * 24 + 16 + e]; }
            if (DGRAD) {
                atomicAdd(dln_w + cl * 8 + e, sw);
                atomicAdd(dln_b + cl * 8 + e, sb);
            }
            if (WGRAD && ch_base == 0) p_cs2[cl * 8 + e] = sc;
        }
    }
}

}  // namespace rvt
