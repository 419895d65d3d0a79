// MLP half of a block (reference maxvit.py:269 with the MLP of :100-118) with the data flow kept in REGISTERS:
//
//     xout = xmid + gamma2 * ( GELU( LN2(xmid) W1^T + b1 ) W2^T + b2 )
//
// Same accumulator-to-operand chaining as the attention half (attn_block.hpp): ONE WAVE owns 32 token rows, read from HBM
// directly in MFMA-operand form (lane = row, 16 bytes per k-step; LayerNorm statistics = in-lane sums + one lane^32
// exchange).  Per 32-column hidden chunk
//     h^T = W1_j v2^T                     (A = weight rows, B = token rows)  -> accumulator: col = token, registers = hidden j
//     GELU / GELU' in the accumulator registers
//     out^T += W2[:, j] g^T               (B = g straight from the registers; W2 columns staged in accumulator order)
// so nothing but the weight panels ever touches LDS and the waves of a workgroup never synchronise.  The fused kernels of
// mlp.hpp move every hidden chunk through LDS twice (fp32 staging + operand tile) behind three workgroup barriers.
//
//   mlpc_fwd_kernel        reads xmid, writes xout                                       (nothing saved: backward recomputes)
//   mlpc_bwd_dgrad_kernel  reads dxout, xmid; writes dxmid = dxout + LN2'(dh W1), dh = (dxout (W2 gamma)) * GELU'(h)
// (The weight gradients of the MLP half stay with mlp_bwd_fused_kernel<MODE 2> of mlp.hpp.  A chained version — wave w of
// a workgroup owns hidden chunk w, products in N-form so that the hidden index sits in the lanes and the tokens in the
// registers, v2 / dxout transposed by identity MFMAs — was built and measured in round 2: 4.85 ms, 4.16 with a lock-step
// barrier, 3.05 with the weights in LDS, against 3.05 ms for the LDS-tile kernel: its eight waves re-read the same rows
// (7 GB fetched for 2 GB of input) and two waves per SIMD do not hide the load latency.  Removed)
#pragma once
#include "common.hpp"
#include "attn_block.hpp"
#include "gelu_lut.hpp"

namespace rvt {

// stage a row-major [rows][K] weight matrix into an LDS operand matrix; PERM: 32-column blocks in accumulator order
template <class T, int K, bool PERM>
__device__ __forceinline__ void chain_stage_weights(char* dst, const T* __restrict__ src, int rows, int tid, int nthreads) {
    constexpr int FPR = K / 8;
    for (int f = tid; f < rows * FPR; f += nthreads) {
        const int row = f / FPR, fcg = f % FPR;
        frag_t<T> v;
        if (!PERM) {
            v = frag_load<T>(src + (size_t)row * K + fcg * 8);
        } else {
            const int blk = fcg >> 2, q = (fcg >> 1) & 1, half = fcg & 1;
            const T* p = src + (size_t)row * K + blk * 32 + 16 * q + 4 * half;
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] = p[e]; v[4 + e] = p[8 + e]; }
        }
        opm_store_frag<T>(dst, rows, row, fcg, v);
    }
}

template <class T, int C> struct McSmem {
    static constexpr int KT = C / TileGeom<T>::BK, HID = 4 * C;
    static constexpr int W_1 = KT * HID * 128;                     // [4C rows][C]
    static constexpr int W_2 = (HID / TileGeom<T>::BK) * C * 128;  // [C rows][4C]   (forward: W2, columns in accumulator order)
    static constexpr int K_LNW = 0, K_LNB = C, K_B2 = 2 * C, K_GAM = 3 * C, K_B1 = 4 * C, NCONST = 8 * C;
};

// one token row per lane (row = tile * 32 + lane & 31) in operand form; LayerNorm in place
template <class T, int C>
__device__ __forceinline__ void mc_load_row(frag_t<T> (&f)[C / 16], const T* __restrict__ src, int row, bool valid, int half) {
#pragma unroll
    for (int ks = 0; ks < C / 16; ks++) {
        const frag_t<T> v = frag_load<T>(src + (size_t)(valid ? row : 0) * C + (2 * ks + half) * 8);
        const frag_t<T> z = frag_zero<T>();
        f[ks] = valid ? v : z;
    }
}
template <class T, int C>
__device__ __forceinline__ void mc_layernorm(const frag_t<T> (&xf)[C / 16], frag_t<T> (&uf)[C / 16], const float* k_lnw,
                                              const float* k_lnb, bool valid, int half, float eps, float& mean, float& rstd) {
    constexpr int KS = C / 16;
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int e = 0; e < 8; e++) s += (float)xf[ks][e];
    s += __shfl_xor(s, 32);
    mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ks++)
#pragma unroll
        for (int e = 0; e < 8; e++) { const float d = (float)xf[ks][e] - mean; q += d * d; }
    q += __shfl_xor(q, 32);
    rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        float w[8], bb[8];
        load_cols<8>(k_lnw, 16 * ks + 8 * half, w);
        load_cols<8>(k_lnb, 16 * ks + 8 * half, bb);
        frag_t<T> o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (T)(valid ? ((float)xf[ks][e] - mean) * rstd * w[e] + bb[e] : 0.f);
        uf[ks] = o;
    }
}

// ===================================================================================================== forward
template <class T, int C, int WPB, int MINW>
__global__ void __launch_bounds__(64 * WPB, MINW)
mlpc_fwd_kernel(const T* __restrict__ xmid, T* __restrict__ xout, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                const T* __restrict__ W1, const float* __restrict__ b1, const T* __restrict__ W2, const float* __restrict__ b2,
                const float* __restrict__ gamma, int M, float eps) {
    typedef McSmem<T, C> S;
    constexpr int KS = C / 16, NCB = C / 32, HID = 4 * C, NJC = HID / 32;
    __shared__ __attribute__((aligned(16))) char smem[S::W_1 + S::W_2 + S::NCONST * 4 + GeluTab<T>::BYTES];
    char* const W1_l = smem;
    char* const W2_l = smem + S::W_1;
    float* const kst = reinterpret_cast<float*>(smem + S::W_1 + S::W_2);
    float* const lut = kst + S::NCONST;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, half = lane >> 5, wave = tid >> 6;
    GeluTab<T>::template fill<false>(lut, tid, 64 * WPB);
    chain_stage_weights<T, C, false>(W1_l, W1, HID, tid, 64 * WPB);
    chain_stage_weights<T, HID, true>(W2_l, W2, C, tid, 64 * WPB);
    for (int i = tid; i < C; i += 64 * WPB) {
        kst[S::K_LNW + i] = ln_w[i]; kst[S::K_LNB + i] = ln_b[i]; kst[S::K_B2 + i] = b2[i]; kst[S::K_GAM + i] = gamma[i];
    }
    for (int i = tid; i < HID; i += 64 * WPB) kst[S::K_B1 + i] = b1[i];
    __syncthreads();

    const int n_tiles = (M + 31) / 32;
    // the next tile's rows are fetched while this one is computed: with two waves per SIMD nothing else covers the HBM
    // round trip at the top of a tile (sixteen registers; stores of this tile retire behind the prefetch, in order)
    frag_t<T> xn[KS];
    {
        const int t0 = blockIdx.x * WPB + wave;
        mc_load_row<T, C>(xn, xmid, t0 * 32 + li, t0 < n_tiles && t0 * 32 + li < M, half);
    }
    for (int tile = blockIdx.x * WPB + wave; tile < n_tiles; tile += gridDim.x * WPB) {
        const int row = tile * 32 + li;
        const bool valid = row < M;
        frag_t<T> xf[KS], uf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) xf[ks] = xn[ks];
        {
            const int tn = tile + gridDim.x * WPB;
            mc_load_row<T, C>(xn, xmid, tn * 32 + li, tn < n_tiles && tn * 32 + li < M, half);
        }
        float mean, rstd;
        // (interior tiles - wave-uniform - take the LayerNorm without the per-lane "row exists" selects: the kernel is bound by VALU issue,
        //  profiles/r6/experiments.txt item 9, and a tile spent 35 v_cndmask on rows that all exist)
        if (tile * 32 + 32 <= M) mc_layernorm<T, C>(xf, uf, kst + S::K_LNW, kst + S::K_LNB, true, half, eps, mean, rstd);
        else mc_layernorm<T, C>(xf, uf, kst + S::K_LNW, kst + S::K_LNB, valid, half, eps, mean, rstd);
        f32x16 oacc[NCB];
        // the first hidden chunk is peeled: its fc2 products take C = 0 from the instruction (no zero fill of the 16 NCB accumulators)
        auto chunk = [&](int jc, auto first) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first)::value;
            f32x16 h;
            acc_load_rows(h, kst + S::K_B1 + 32 * jc, half);           // fc1 bias = initial value of the accumulator
#pragma unroll
            for (int ks = 0; ks < KS; ks++) mma32(h, opm_load_frag<T>(W1_l, HID, 32 * jc + li, 2 * ks + half), uf[ks]);
            float g[16];
            GeluTab<T>::eval16(lut, h, g);
#pragma unroll
            for (int r = 0; r < 16; r++) g[r] = mul_nopack(g[r], h[r], r);
            frag_t<T> gf[2];
            gf[0] = arr_slot_frag<T>(g, 0);
            gf[1] = arr_slot_frag<T>(g, 1);
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const frag_t<T> wf = opm_load_frag<T>(W2_l, C, cb * 32 + li, 4 * jc + 2 * q + half);
                    if (FIRST && q == 0) mma32_zero(oacc[cb], wf, gf[q]);
                    else mma32(oacc[cb], wf, gf[q]);
                }
        };
        chunk(0, std::true_type());
#pragma unroll 2
        for (int jc = 1; jc < NJC; jc++) chunk(jc, std::false_type());
        // LayerScale + residual (maxvit.py:51-53,269); the raw row pieces are still in registers
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            float r8[2][8];
            acc_to_rows(oacc[cb], r8);
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int ks = 2 * cb + m;
                float res[8], gam[8], b2v[8], o[8];
                frag_to_float<T>(xf[ks], res);
                load_cols<8>(kst + S::K_GAM, 16 * ks + 8 * half, gam);
                load_cols<8>(kst + S::K_B2, 16 * ks + 8 * half, b2v);
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = res[e] + gam[e] * (r8[m][e] + b2v[e]);
                if (valid) frag_store<T>(xout + (size_t)row * C + (2 * ks + half) * 8, frag_from_float<T>(o));
            }
        }
    }
}

// ============================================================================ backward: input-gradient chain
// dh[m][j] = (dxout (W2 gamma))[m][j] * GELU'(h[m][j]);  dv2 = dh W1;  dxmid = dxout + LN2'(dv2; xmid);  dln_w / dln_b += .
// W2gT = (W2 * gamma[:, None])^T stored [4C][C].  dv2^T's A operand (rows c, contraction over j) comes out of the ONE LDS
// image of W1 through the transposing LDS read, in accumulator order.
template <class T, int C, int WPB>
__global__ void __launch_bounds__(64 * WPB)
mlpc_bwd_dgrad_kernel(const T* __restrict__ dxout, const T* __restrict__ xmid, T* __restrict__ dxmid,
                      const float* __restrict__ ln_w, const float* __restrict__ ln_b, const T* __restrict__ W1,
                      const float* __restrict__ b1, const T* __restrict__ W2gT, float* __restrict__ dln_w,
                      float* __restrict__ dln_b, int M, float eps) {
    typedef McSmem<T, C> S;
    constexpr int KS = C / 16, NCB = C / 32, HID = 4 * C, NJC = HID / 32;
    __shared__ __attribute__((aligned(16))) char smem[2 * S::W_1 + S::NCONST * 4 + GeluTab<T>::BYTES];
    char* const W1_l = smem;
    char* const W2_l = smem + S::W_1;                                // (W2 gamma)^T: [4C rows][C]
    float* const kst = reinterpret_cast<float*>(smem + 2 * S::W_1);
    float* const lut = kst + S::NCONST;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, half = lane >> 5, wave = tid >> 6;
    GeluTab<T>::template fill<true>(lut, tid, 64 * WPB);
    chain_stage_weights<T, C, false>(W1_l, W1, HID, tid, 64 * WPB);
    chain_stage_weights<T, C, false>(W2_l, W2gT, HID, tid, 64 * WPB);
    for (int i = tid; i < C; i += 64 * WPB) { kst[S::K_LNW + i] = ln_w[i]; kst[S::K_LNB + i] = ln_b[i]; }
    for (int i = tid; i < HID; i += 64 * WPB) kst[S::K_B1 + i] = b1[i];
    __syncthreads();

    // LayerNorm parameter gradients = column sums over tokens of dv2 * xhat and dv2.  The row pieces hold tokens in the LANES;
    // an MFMA against an identity operand turns a piece block into "col = channel, registers = tokens" (exact), where the
    // column sum is an in-lane sum: one accumulator per 32-channel block instead of 2 x 32 per-lane partial sums.
    frag_t<T> idf[2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int e = 0; e < 8; e++) idf[m][e] = (T)((16 * m + 8 * half + e == li) ? 1.0f : 0.0f);
    float aw[NCB], ab[NCB];                                          // dln_w / dln_b of channel 32 cb + (lane & 31), this half's tokens
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { aw[cb] = 0.f; ab[cb] = 0.f; }

    const int n_tiles = (M + 31) / 32;
    for (int tile = blockIdx.x * WPB + wave; tile < n_tiles; tile += gridDim.x * WPB) {
        const int row = tile * 32 + li;
        const bool valid = row < M;
        frag_t<T> xf[KS], uf[KS], df[KS];
        mc_load_row<T, C>(xf, xmid, row, valid, half);
        mc_load_row<T, C>(df, dxout, row, valid, half);
        float mean, rstd;
        mc_layernorm<T, C>(xf, uf, kst + S::K_LNW, kst + S::K_LNB, valid, half, eps, mean, rstd);
        f32x16 dacc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc_zero(dacc[cb]);
#pragma unroll 2
        for (int jc = 0; jc < NJC; jc++) {
            f32x16 h, dg;
            acc_load_rows(h, kst + S::K_B1 + 32 * jc, half);
            acc_zero(dg);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                mma32(h, opm_load_frag<T>(W1_l, HID, 32 * jc + li, 2 * ks + half), uf[ks]);
                mma32(dg, opm_load_frag<T>(W2_l, HID, 32 * jc + li, 2 * ks + half), df[ks]);
            }
            float dh[16];
            GeluTab<T>::eval16(lut, h, dh);
#pragma unroll
            for (int r = 0; r < 16; r++) dh[r] = mul_nopack(dh[r], dg[r], r);
            frag_t<T> dhf[2];
            dhf[0] = arr_slot_frag<T>(dh, 0);
            dhf[1] = arr_slot_frag<T>(dh, 1);
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                for (int q = 0; q < 2; q++) mma32(dacc[cb], ab_tr_frag<T>(W1_l, HID, 32 * jc + 16 * q, cb * 32, lane), dhf[q]);
        }
        // LayerNorm backward + residual in operand-piece form
        float d8[KS][8], xh[KS][8];
        float gsum = 0.f, gxsum = 0.f;
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            float r8[2][8];
            acc_to_rows(dacc[cb], r8);
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int ks = 2 * cb + m;
                float w[8];
                load_cols<8>(kst + S::K_LNW, 16 * ks + 8 * half, w);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    d8[ks][e] = valid ? r8[m][e] : 0.f;
                    xh[ks][e] = valid ? ((float)xf[ks][e] - mean) * rstd : 0.f;
                    const float gw = d8[ks][e] * w[e];
                    gsum += gw;
                    gxsum += gw * xh[ks][e];
                }
            }
            f32x16 tw, tb;
            acc_zero(tw);
            acc_zero(tb);
#pragma unroll
            for (int m = 0; m < 2; m++) {
                float pw[8];
#pragma unroll
                for (int e = 0; e < 8; e++) pw[e] = d8[2 * cb + m][e] * xh[2 * cb + m][e];
                mma32(tw, frag_from_float<T>(pw), idf[m]);
                mma32(tb, frag_from_float<T>(d8[2 * cb + m]), idf[m]);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) { aw[cb] += tw[r]; ab[cb] += tb[r]; }
        }
        gsum += __shfl_xor(gsum, 32);
        gxsum += __shfl_xor(gxsum, 32);
        const float m1 = gsum / (float)C, m2 = gxsum / (float)C;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            float w[8], o[8];
            load_cols<8>(kst + S::K_LNW, 16 * ks + 8 * half, w);
#pragma unroll
            for (int e = 0; e < 8; e++) o[e] = (float)df[ks][e] + rstd * (d8[ks][e] * w[e] - m1 - xh[ks][e] * m2);
            if (valid) frag_store<T>(dxmid + (size_t)row * C + (2 * ks + half) * 8, frag_from_float<T>(o));
        }
    }
    // fold the two halves and the waves: one atomic per channel per workgroup
    __syncthreads();                                                 // weights are dead: the LDS becomes reduction scratch
    float* const red = reinterpret_cast<float*>(smem);               // [WPB][dln_w C | dln_b C]
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        const float a = aw[cb] + __shfl_xor(aw[cb], 32), b = ab[cb] + __shfl_xor(ab[cb], 32);
        if (half == 0) {
            red[wave * 2 * C + 32 * cb + li] = a;
            red[wave * 2 * C + C + 32 * cb + li] = b;
        }
    }
    __syncthreads();
    for (int v = tid; v < 2 * C; v += 64 * WPB) {
        float sum = 0.f;
        for (int w = 0; w < WPB; w++) sum += red[w * 2 * C + v];
        atomicAdd((v < C ? dln_w : dln_b) + (v % C), sum);
    }
}

// =========================================================================== backward: weight gradients (bf16, C = 64)
//   dW1[j][c] = sum_t dh[t][j] v2[t][c],  db1[j] = sum_t dh[t][j],  S2[c][j] = sum_t dxout[t][c] g[t][j],  cs2[c] = sum_t dxout[t][c]
//   with v2 = LN2(xmid), h = v2 W1^T + b1, g = GELU(h), dh = (dxout (gamma W2)) GELU'(h)      (everything recomputed)
// A workgroup of eight waves walks 32-token tiles; wave w owns the hidden columns j = 32 w .. 32 w + 31 and keeps its
// rows of W1 and (gamma W2)^T in registers (no weights in LDS).  Per tile the v2 rows (LayerNorm done once, sixteen lanes
// per row) and the dxout rows go into two small LDS tiles; from there every operand is a plain or a transposing read:
//   h, dg    A = token rows (16-byte row pieces), B = the wave's weight rows -> accumulator column = hidden j (this lane),
//            registers = the tile's tokens in accumulator order;
//   dW1, S2  contract over the TOKENS: one operand is g / dh straight from those accumulator registers, the other is
//            v2^T / dxout^T = ds_read_b64_tr_b16 on the row-major tiles, addressed in the same (accumulator) token order.
// No identity-MFMA transposes, no copies of the hidden activations anywhere; GELU and GELU' come from one 8-byte nearest-entry table gather.
// Partial results per workgroup in `ws`, laid out as mlp_fold_partials expects:
// [dW1: grid x 4C x C][S2: grid x C x 4C][db1: 2 grid x 4C][cs2: grid x C].
constexpr int MCW_ROWB = 144;              // LDS row pitch of the [32 tokens][64 channels] bf16 tiles (bank spread for the transposing reads)
// DGRAD = true (round 4): the same launch ALSO produces the input gradient dxmid = dxout + LN2'(dh W1) and dln_w / dln_b — what
// mlpc_bwd_dgrad_kernel computes from a second recompute of h and a second read of xmid / dxout.  dh is already here, split over the
// waves by hidden column; it takes three more stations, each one phase (= one workgroup barrier) behind the previous:
//   compute(i)   every wave also writes its dh columns to the tile DH^T[j][t]                     (4 eight-byte LDS stores per lane)
//   duty(i-1)    all eight waves multiply  dv2[t][c] = sum_j DH[t][j] W1[j][c]  as 16 x 16 blocks (16 tokens x 16 channels per wave,
//                eight v_mfma_f32_16x16x32_bf16; operands = rows of W1^T staged in LDS once and rows of DH, plain 16-byte reads) and
//                leave it in PB[t][c] (bf16).  (Round 4: two waves, 32 x 32 blocks, sixteen chained MFMAs each - six waves waiting.)
//   lnbwd(i-2)   all 512 threads in the staging role (16 per token, 4 channels each): LayerNorm backward + residual, 8-byte
//                stores of dxmid; the rows of xmid / dxout come back from L2 (fetched one phase ahead), mean / rstd from the
//                tile's own stash through a 1-KB LDS ring.
// Measured (M = 7.74 M tokens): 2.41 ms against 1.48 + 1.48 ms for the two launches; per station (compile-time masks): loop
// structure +0.14, dh store +0.12, duty +0.33 (two of eight waves: the others wait at the barrier), LayerNorm backward +0.25 ms.
constexpr int MCW_HP = 528;                // LDS row pitch of the [.][256 hidden] bf16 tiles (DH, W1^T)
constexpr int MCW_PP = 144;                // LDS row pitch of the [32 tokens][64 channels] bf16 tile PB
template <bool DGRAD>
__global__ void __launch_bounds__(512)
mlpc_bwd_wgrad_kernel(const bf16* __restrict__ dxout, const bf16* __restrict__ xmid, const float* __restrict__ ln_w,
                      const float* __restrict__ ln_b, const bf16* __restrict__ W1, const float* __restrict__ b1,
                      const bf16* __restrict__ W2gT, float* __restrict__ ws, int M, float eps,
                      const bf16* __restrict__ W1T, bf16* __restrict__ dxmid, float* __restrict__ dln_w, float* __restrict__ dln_b) {
    typedef bf16 T;
    constexpr int C = 64, KS = C / 16, NCB = C / 32, HID = 4 * C, TILE = 32 * MCW_ROWB;
    constexpr int DHT_BYTES = HID * 64;                                          // one DH^T tile: [256 hidden][32 tokens] bf16
    constexpr int EXTRA = DGRAD ? (C * MCW_HP + 2 * DHT_BYTES + 2 * 32 * MCW_PP + 4 * 32 * 8) : 0;
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE + GELU_NLUT2_BYTES + 16 + EXTRA];
    char* const V2 = smem;
    char* const DX = smem + 2 * TILE;
    float* const lut = reinterpret_cast<float*>(smem + 4 * TILE);
    char* const WT = smem + ((4 * TILE + GELU_NLUT2_BYTES + 15) & ~15);          // W1^T: [64 channels][256 hidden]
    char* const DH = WT + C * MCW_HP;                                            // 2 x DH^T [256 hidden][32 tokens], 8-byte chunk ^= (row >> 2) & 7
    char* const PB = DH + 2 * DHT_BYTES;                                         // 2 x [32 tokens][64 channels]
    float* const MR = reinterpret_cast<float*>(PB + 2 * 32 * MCW_PP);            // ring of 4 tiles x [32 tokens] x {mean, rstd} (stash -> lnbwd)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, half = lane >> 5, wave = tid >> 6;
    gelu_nlut2_fill(lut, tid, 512);
    if (DGRAD)
        for (int f = tid; f < C * (HID / 8); f += 512)
            *reinterpret_cast<frag_t<T>*>(WT + (f / (HID / 8)) * MCW_HP + (f % (HID / 8)) * 16) = frag_load<T>(W1T + (size_t)f * 8);
    // this wave's weight rows j = 32 wave + li, as B operands (k-step ks: channels 16 ks + 8 half ..)
    frag_t<T> w1f[KS], w2f[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        w1f[ks] = frag_load<T>(W1 + (size_t)(32 * wave + li) * C + 16 * ks + 8 * half);
        w2f[ks] = frag_load<T>(W2gT + (size_t)(32 * wave + li) * C + 16 * ks + 8 * half);
    }
    const float b1v = b1[32 * wave + li];
    // staging role: 16 threads per token row, 4 channels each
    const int srow = tid >> 4, spc = tid & 15;
    float lw[4], lb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { lw[i] = ln_w[4 * spc + i]; lb[i] = ln_b[4 * spc + i]; }
    // transposing reads in accumulator token order: slots 0..3 = tokens 16 q + 4 half + 0..3, slots 4..7 = those + 8
    const int tr_off = (4 * half + ((lane & 15) >> 2)) * MCW_ROWB + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    f32x16 dw1[NCB], s2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { acc_zero(dw1[cb]); acc_zero(s2[cb]); }
    float db1 = 0.f, cs[4] = {0.f, 0.f, 0.f, 0.f};
    float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};          // dln_w / dln_b of channels 4 spc .. (this row slot)

    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    struct Stage { u32x2 x, d; };
    const int n_tiles = (M + 31) / 32;
    auto fetch = [&](Stage& st, int tile) {
        const int row = tile * 32 + srow;
        const bool ok = tile >= 0 && tile < n_tiles && row < M;
        const size_t off = (size_t)(ok ? row : 0) * C + 4 * spc;
        const u32x2 z = {0u, 0u};
        const u32x2 vx = *reinterpret_cast<const u32x2*>(xmid + off), vd = *reinterpret_cast<const u32x2*>(dxout + off);
        st.x = ok ? vx : z;
        st.d = ok ? vd : z;
    };
    auto stash = [&](const Stage& st, int tile, int buf, int slot) {
        if (tile >= n_tiles) return;
        const bf16x4 xb = __builtin_bit_cast(bf16x4, st.x), db = __builtin_bit_cast(bf16x4, st.d);
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; i++) { x[i] = (float)xb[i]; cs[i] += (float)db[i]; }
        const float mean = row16_sum((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / C);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) { x[i] -= mean; ss += x[i] * x[i]; }
        const float rstd = 1.0f / sqrtf(row16_sum(ss) * (1.0f / C) + eps);
        bf16x4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = (T)fmaf(x[i] * rstd, lw[i], lb[i]);
        *reinterpret_cast<bf16x4*>(V2 + buf * TILE + srow * MCW_ROWB + spc * 8) = v;
        *reinterpret_cast<u32x2*>(DX + buf * TILE + srow * MCW_ROWB + spc * 8) = st.d;
        if (DGRAD && spc == 0) { MR[((slot & 3) * 32 + srow) * 2] = mean; MR[((slot & 3) * 32 + srow) * 2 + 1] = rstd; }
    };
    auto compute = [&](int buf) {
        const char* const v2t = V2 + buf * TILE;
        const char* const dxt = DX + buf * TILE;
        // accumulator column = hidden j (this lane), registers = the tile's tokens
        f32x16 h, dg;
#pragma unroll
        for (int r = 0; r < 16; r++) h[r] = b1v;
        acc_zero(dg);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            mma32(h, *reinterpret_cast<const frag_t<T>*>(v2t + li * MCW_ROWB + (2 * ks + half) * 16), w1f[ks]);
            mma32(dg, *reinterpret_cast<const frag_t<T>*>(dxt + li * MCW_ROWB + (2 * ks + half) * 16), w2f[ks]);
        }
        frag_t<T> gf[2], dhf[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {               // slot q of the operands = accumulator registers 8q .. 8q+7
            float x8[8], g8[8], p8[8];
#pragma unroll
            for (int e = 0; e < 8; e++) x8[e] = h[8 * q + e];
            gelu_both_nlut2_8(lut, x8, g8, p8);
            float d8[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                d8[e] = mul_nopack(dg[8 * q + e], p8[e], e);
                db1 += d8[e];
            }
            gf[q] = frag_from_float<T>(g8);             // (converted as whole fragments: pairs per v_cvt_pk_bf16_f32)
            dhf[q] = frag_from_float<T>(d8);
        }
        if (DGRAD) {                  // dh^T[j][t]: this lane's hidden column is a ROW of the tile; accumulator registers 4 g .. 4 g + 3 are the
                                      // four consecutive tokens 8 g + 4 half ..: four 8-byte stores (the [t][j] form took sixteen 2-byte stores)
            char* const dhb = DH + buf * DHT_BYTES + (32 * wave + li) * 64;
            const int sw = (li >> 2) & 7;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                bf16x4 v;
#pragma unroll
                for (int w = 0; w < 4; w++) v[w] = dhf[g >> 1][4 * (g & 1) + w];
                *reinterpret_cast<bf16x4*>(dhb + (((2 * g + half) ^ sw) << 3)) = v;
            }
        }
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int o = tr_off + 16 * q * MCW_ROWB + 64 * cb;
                const frag_t<T> vT = frag_from_tr<T>(reinterpret_cast<const bf16*>(v2t + o), reinterpret_cast<const bf16*>(v2t + o + 8 * MCW_ROWB));
                const frag_t<T> dT = frag_from_tr<T>(reinterpret_cast<const bf16*>(dxt + o), reinterpret_cast<const bf16*>(dxt + o + 8 * MCW_ROWB));
                mma32(dw1[cb], dhf[q], vT);         // rows j, columns c
                mma32(s2[cb], dT, gf[q]);           // rows c, columns j
            }
    };
    // ---- DGRAD stations (see the head comment).  `k` = how many tiles this workgroup has started before the one in question.
    auto duty = [&](int buf, int tile, int k) {
        if (!DGRAD || tile < 0 || tile >= n_tiles) return;
        (void)k;
        // dv2^T[c][t] = sum_j W1^T[c][j] DH[t][j] cut into EIGHT 16 x 16 blocks, one per wave (v_mfma_f32_16x16x32_bf16): wave w owns
        // tokens 16 (w & 1) .., channels 16 (w >> 1) ..  (Round 4 gave the product to two waves as 32 x 32 blocks, sixteen chained
        // MFMAs each, while the other six waited at the barrier.)  A = rows of W1^T, B = rows of DH: a lane ends up with FOUR
        // CONSECUTIVE CHANNELS of one token = one 8-byte store into PB.
        const int l15 = lane & 15, kg = lane >> 4, tb = wave & 1, cq = wave >> 1;
        const char* const a = WT + (16 * cq + l15) * MCW_HP + kg * 16;
        // B = rows of DH ("row token, slots = hidden 32 ks + 8 kg + e") out of the DH^T tile through the transposing read: lane L of a
        // 16-lane group addresses the four tokens 16 tb + 4 (L & 3) .. of hidden row 32 ks + 8 kg + (L >> 2) (+ 4 for slots 4..7)
        const int jr = 8 * kg + (l15 >> 2), ct = 4 * tb + (l15 & 3);
        const char* const b_lo = DH + buf * DHT_BYTES + jr * 64 + ((ct ^ ((2 * kg) & 7)) << 3);
        const char* const b_hi = DH + buf * DHT_BYTES + (jr + 4) * 64 + ((ct ^ ((2 * kg + 1) & 7)) << 3);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};          // two chains: a dependent MFMA waits for its predecessor
#pragma unroll
        for (int ks = 0; ks < HID / 32; ks += 2) {
            mma16(acc, *reinterpret_cast<const frag_t<T>*>(a + ks * 64),
                  frag_from_tr<T>(reinterpret_cast<const bf16*>(b_lo + ks * 2048), reinterpret_cast<const bf16*>(b_hi + ks * 2048)));
            mma16(acc2, *reinterpret_cast<const frag_t<T>*>(a + (ks + 1) * 64),
                  frag_from_tr<T>(reinterpret_cast<const bf16*>(b_lo + (ks + 1) * 2048), reinterpret_cast<const bf16*>(b_hi + (ks + 1) * 2048)));
        }
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = (T)(acc[r] + acc2[r]);
        *reinterpret_cast<bf16x4*>(PB + buf * (32 * MCW_PP) + (16 * tb + l15) * MCW_PP + (16 * cq + 4 * kg) * 2) = o;
    };
    auto lnbwd = [&](const Stage& st, int buf, int tile, int slot) {
        if (!DGRAD || tile < 0 || tile >= n_tiles) return;
        const int row = tile * 32 + srow;
        const bf16x4 xb = __builtin_bit_cast(bf16x4, st.x), db = __builtin_bit_cast(bf16x4, st.d);
        const bf16x4 gb = *reinterpret_cast<const bf16x4*>(PB + buf * (32 * MCW_PP) + srow * MCW_PP + spc * 8);
        const float mean = MR[((slot & 3) * 32 + srow) * 2], rstd = MR[((slot & 3) * 32 + srow) * 2 + 1];   // (from this tile's stash)
        float x[4];
#pragma unroll
        for (int i = 0; i < 4; i++) x[i] = (float)xb[i] - mean;
        float gw[4], s1 = 0.f, s2x = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float g = (float)gb[i];
            x[i] *= rstd;                                            // xhat
            gw[i] = g * lw[i];
            s1 += gw[i];
            s2x += gw[i] * x[i];
            aw[i] += g * x[i];
            ab[i] += g;
        }
        const float m1 = row16_sum(s1) * (1.0f / C), m2 = row16_sum(s2x) * (1.0f / C);
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = (T)((float)db[i] + rstd * (gw[i] - m1 - x[i] * m2));
        if (row < M) *reinterpret_cast<bf16x4*>(dxmid + (size_t)row * C + 4 * spc) = o;
    };
    // (same pipeline as the stem weight gradient, csrc/stem.hpp: two tiles in flight, the waves sharing a SIMD staggered)
    const bool mfma_first = wave < 4;
    const int t0 = blockIdx.x, G = gridDim.x;
    Stage sa, sb, sla, slb;                         // sla / slb: the rows phase A / B needs for its LayerNorm backward, fetched one phase ahead
    fetch(sa, t0);
    fetch(sb, t0 + G);
    __syncthreads();                                // table (and W1^T)
    stash(sa, t0, 0, 0);
    fetch(sa, t0 + 2 * G);
    if (DGRAD) fetch(sla, -1);
    lds_barrier();
    int k = 0;                                      // tile slots started by this workgroup
    int tile = t0;
    for (; tile < n_tiles; tile += 2 * G, k += 2) {
        if (mfma_first) compute(0);
        stash(sb, tile + G, 1, k + 1);
        fetch(sb, tile + 3 * G);
        if (DGRAD) fetch(slb, tile - G);            // (younger than everything this phase waits for)
        if (!mfma_first) compute(0);
        duty(1, tile - G, k - 1);
        lnbwd(sla, 0, tile - 2 * G, k - 2);
        lds_barrier();
        if (mfma_first && tile + G < n_tiles) compute(1);
        stash(sa, tile + 2 * G, 0, k + 2);
        fetch(sa, tile + 4 * G);
        if (DGRAD) fetch(sla, tile);
        if (!mfma_first && tile + G < n_tiles) compute(1);
        duty(0, tile, k);
        lnbwd(slb, 1, tile - G, k - 1);
        lds_barrier();
    }
    if (DGRAD) {                                    // drain: the stations of the last two tile slots
        fetch(slb, tile - G);
        duty(1, tile - G, k - 1);
        lnbwd(sla, 0, tile - 2 * G, k - 2);
        lds_barrier();
        lnbwd(slb, 1, tile - G, k - 1);
    }
    const size_t nwg = gridDim.x, wg = blockIdx.x;
    float* const p_dw1 = ws + wg * (size_t)(HID * C);
    float* const p_s2 = ws + nwg * (size_t)(HID * C) + wg * (size_t)(C * HID);
    float* const p_db1 = ws + 2 * nwg * (size_t)(HID * C) + (wg * 2) * (size_t)HID;
    float* const p_cs2 = ws + 2 * nwg * (size_t)(HID * C) + 2 * nwg * (size_t)HID + wg * (size_t)C;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            p_dw1[(size_t)(32 * wave + acc_row(r, lane)) * C + 32 * cb + li] = dw1[cb][r];
            p_s2[(size_t)(32 * cb + acc_row(r, lane)) * HID + 32 * wave + li] = s2[cb][r];
        }
    db1 += __shfl_xor(db1, 32);
    if (half == 0) {
        p_db1[32 * wave + li] = db1;
        p_db1[HID + 32 * wave + li] = 0.f;
    }
    // column sums of dxout (and the LayerNorm parameter gradients): fold the 32 row slots through LDS (the tiles are done with)
    float* const red = reinterpret_cast<float*>(smem);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) red[srow * C + 4 * spc + i] = cs[i];
    __syncthreads();
    if (tid < C) {
        float s = 0.f;
        for (int r = 0; r < 32; r++) s += red[r * C + tid];
        p_cs2[tid] = s;
    }
    if (DGRAD) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) { red[srow * C + 4 * spc + i] = aw[i]; red[32 * C + srow * C + 4 * spc + i] = ab[i]; }
        __syncthreads();
        if (tid < 2 * C) {                          // one atomic per channel per workgroup
            float s = 0.f;
            for (int r = 0; r < 32; r++) s += red[(tid / C) * 32 * C + r * C + (tid % C)];
            atomicAdd((tid < C ? dln_w : dln_b) + (tid % C), s);
        }
    }
}

}  // namespace rvt
