// Fused attention half of a PartitionAttentionCl block (reference maxvit.py:268 with SelfAttentionCl :343-354 on the
// partitions of :273-304):
//
//     xmid = x + gamma1 * ( softmax( q k^T / sqrt(dh) ) v  Wp^T + bp ),    [q|k|v] = LN1(x) Wqkv^T + bqkv   per partition
//
// Op-by-op this half moves ~12 (forward) + ~25 (backward) activation rows of C elements per token through HBM (LayerNorm
// output, the 3C-wide qkv and dqkv, the attention output and its gradient, ...); at C <= 128 every one of those kernels is
// HBM-bound.  Here ONE WAVE owns ONE PARTITION (L <= 96 tokens) from the x rows to the xmid rows:
//
//   attn_block_fwd_kernel   reads x, writes xmid (+ a = attention output, kept for the proj weight gradient)     1 + 2 rows
//   attn_block_bwd_kernel   reads dxmid, x; recomputes LN1 / q / k / v / P; writes dx and dqkv (+ LN1(x)) for the
//                           qkv weight-gradient GEMM                                                             2 + 4(5) rows
//
// Everything between those rows lives in registers.  The chain of products is arranged so that the accumulator of one
// MFMA *is* the operand of the next (C/D layout of v_mfma_f32_32x32x*: col = lane & 31, row = (r & 3) + 8 (r >> 2) +
// 4 (lane >> 5); an A/B operand wants row = lane & 31 and eight contraction slots per lane: an accumulator block Out[m][n]
// read as an operand is "row n, contraction over m", with the m's in accumulator order — A and B slots pair one-to-one, so
// any enumeration of the contraction index is fine as long as both operands use the same one):
//
//   Q^T = Wq u^T, K^T = Wk u^T   (A = weight rows, B = token rows)   -> operands "row token, contract d"
//   S^T = K Q^T                  lane owns a query column: softmax max / sum are in-lane + one lane^32 exchange
//   V   = u Wv^T                 (A = token rows, B = weight rows)   -> operand "row d, contract key"
//   O^T = V^T P^T                P^T straight from the S^T accumulators
//   out^T = Wp O^T               Wp columns pre-permuted into accumulator order when the weights are staged into LDS
//
// Token rows are read from HBM directly in MFMA-operand form (lane = row, 16 bytes of the row per k-step), so LayerNorm
// statistics are an in-lane sum plus one exchange with lane^32, and the epilogue turns accumulator columns back into
// 16-byte row pieces with v_permlane32_swap (the two halves of a wave hold the two 4-channel halves of every 8-channel
// piece).  Window / grid partitioning is index arithmetic on the token rows (attn.hpp), weights stay in LDS for the
// whole launch (persistent workgroups), and the waves of a workgroup never synchronise with each other.
//
// The backward additionally needs the transposes P, dS (contraction over queries for dV, dK): one trip through a
// wave-private LDS tile, written as 8-byte row pieces from the accumulators and read back with the transposing LDS read.
#pragma once
#include "common.hpp"
#include "attn.hpp"
#include "mlp.hpp"

namespace rvt {

// per-row (= per accumulator register) additive constants of a T-form block: p points at the 32 values of the block
__device__ __forceinline__ void acc_add_rows(f32x16& c, const float* p, int half) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + 8 * g + 4 * half);
#pragma unroll
        for (int w = 0; w < 4; w++) c[4 * g + w] += t[w];
    }
}


// S^T column softmax for one query block.  In: raw scores s[bj][r] (keys down the registers, this lane's query).  Out:
// p[bj][r] = exp(scale (s - max)) (UN-normalised) and the column's 1/sum.  Padded keys (j >= L) only exist in the LAST
// 32-key block: they are selected away by comparing the (compile-time) accumulator row of register r with one per-lane
// limit; exp2 with scale * log2(e) folded into one multiply.
template <int NB>
__device__ __forceinline__ float ab_softmax_cols(const f32x16 (&s)[NB], float (&p)[NB][16], int klim, float scale_log2e) {
    float mx = -3.0e38f;
#pragma unroll
    for (int bj = 0; bj < NB; bj++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float v = (bj == NB - 1 && ((r & 3) + 8 * (r >> 2)) >= klim) ? -1.0e30f : s[bj][r];
            p[bj][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int bj = 0; bj < NB; bj++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float e = fast_exp2((p[bj][r] - mx) * scale_log2e);
            p[bj][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32);
    return 1.0f / sum;
}

// ... and the same constants as the INITIAL value of the accumulator (bias folded into the first MFMA: no zero fill, no adds)
__device__ __forceinline__ void acc_load_rows(f32x16& c, const float* p, int half) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + 8 * g + 4 * half);
#pragma unroll
        for (int w = 0; w < 4; w++) c[4 * g + w] = t[w];
    }
}

template <class T, int C> struct AbSmem {
    static constexpr int KT = C / TileGeom<T>::BK;
    static constexpr int W_QKV = KT * 3 * C * 128;        // [3C rows][C]   (Wqkv; rows = [head][q|k|v][dh])
    static constexpr int W_P = KT * C * 128;              // [C rows][C]    (fwd: Wp, columns in accumulator order; bwd: (gamma Wp)^T)
    static constexpr int OFF_P = W_QKV, OFF_K = W_QKV + W_P;
    static constexpr int K_LNW = 0, K_LNB = C, K_BQKV = 2 * C, K_BP = 5 * C, K_GAM = 6 * C, NCONST = 7 * C;
    static constexpr int OFF_S = OFF_K + NCONST * 4;      // per-wave scratch (backward only) starts here
};

// stage a row-major [rows][C] weight matrix into an LDS operand matrix; PERM: the 32-column blocks are stored in
// accumulator order (position 16 q + 8 half + e holds column (e & 3) + 8 (2 q + (e >> 2)) + 4 half)
template <class T, int C, bool PERM>
__device__ __forceinline__ void ab_stage_weights(char* dst, const T* __restrict__ src, int rows, int tid, int nthreads) {
    constexpr int FPR = C / 8;
    for (int f = tid; f < rows * FPR; f += nthreads) {
        const int row = f / FPR, fcg = f % FPR;
        frag_t<T> v;
        if (!PERM) {
            v = frag_load<T>(src + (size_t)row * C + fcg * 8);
        } else {
            const int blk = fcg >> 2, q = (fcg >> 1) & 1, half = fcg & 1;
            const T* p = src + (size_t)row * C + blk * 32 + 16 * q + 4 * half;
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] = p[e]; v[4 + e] = p[8 + e]; }
        }
        opm_store_frag<T>(dst, rows, row, fcg, v);
    }
}

// token rows of this lane (one per 32-token block of the partition) in operand form: piece (ks, half) = 16 bytes
template <class T, int C, int NB>
__device__ __forceinline__ void ab_load_rows(frag_t<T> (&f)[NB][C / 16], const T* __restrict__ src, const int (&tok)[NB],
                                              const bool (&valid)[NB], int half) {
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int ks = 0; ks < C / 16; ks++) {
            const frag_t<T> v = frag_load<T>(src + (size_t)tok[b] * C + (2 * ks + half) * 8);     // padded rows read token 0
            const frag_t<T> z = frag_zero<T>();
            f[b][ks] = valid[b] ? v : z;
        }
}

// LayerNorm (maxvit.py:229) of the rows held in operand form; keeps mean / rstd for the backward
template <class T, int C, int NB>
__device__ __forceinline__ void ab_layernorm(const frag_t<T> (&xf)[NB][C / 16], frag_t<T> (&uf)[NB][C / 16], const float* k_lnw,
                                              const float* k_lnb, const bool (&valid)[NB], int half, float eps,
                                              float (&mean)[NB], float (&rstd)[NB]) {
    constexpr int KS = C / 16;
#pragma unroll
    for (int b = 0; b < NB; b++) {
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int e = 0; e < 8; e++) s += (float)xf[b][ks][e];
        s += __shfl_xor(s, 32);
        mean[b] = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = (float)xf[b][ks][e] - mean[b]; q += d * d; }
        q += __shfl_xor(q, 32);
        rstd[b] = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            float w[8], bb[8];
            load_cols<8>(k_lnw, 16 * ks + 8 * half, w);
            load_cols<8>(k_lnb, 16 * ks + 8 * half, bb);
#pragma unroll
            for (int e = 0; e < 8; e++)
                uf[b][ks][e] = (T)(valid[b] ? ((float)xf[b][ks][e] - mean[b]) * rstd[b] * w[e] + bb[e] : 0.f);
        }
    }
}

// T-form projection block: acc[feature][token] += W[row0 + .][:] . u[token][:]   (A = weight rows, B = token rows)
template <class T, int C>
__device__ __forceinline__ void ab_proj_t(f32x16& acc, const char* Wl, int wrows, int row0, const frag_t<T> (&uf)[C / 16], int li, int half) {
#pragma unroll
    for (int ks = 0; ks < C / 16; ks++) mma32(acc, opm_load_frag<T>(Wl, wrows, row0 + li, 2 * ks + half), uf[ks]);
}
// N-form projection block: acc[token][feature]   (A = token rows, B = weight rows)
template <class T, int C>
__device__ __forceinline__ void ab_proj_n(f32x16& acc, const char* Wl, int wrows, int row0, const frag_t<T> (&uf)[C / 16], int li, int half) {
#pragma unroll
    for (int ks = 0; ks < C / 16; ks++) mma32(acc, uf[ks], opm_load_frag<T>(Wl, wrows, row0 + li, 2 * ks + half));
}

template <class T> __device__ __forceinline__ void acc_to_frags(const f32x16& a, frag_t<T> (&f)[2]) {
    f[0] = acc_slot_frag<T>(a, 0);
    f[1] = acc_slot_frag<T>(a, 1);
}

template <int NB> struct AbPart {      // which partition, which token rows
    int tok[NB];
    bool valid[NB];
};
template <int NB>
__device__ __forceinline__ void ab_partition(AbPart<NB>& pt, const AttnGeom& g, int pidx, int li) {
    uint32_t f, p;
    g.dP.divmod((uint32_t)pidx, f, p);
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int l = 32 * b + li;
        pt.valid[b] = b < NB - 1 ? true : l < g.L;        // (NB = ceil(L / 32): only the LAST block has rows beyond L - a compile-time `true` drops the selects of the others)
        pt.tok[b] = attn_token(g, (int)f, (int)p, pt.valid[b] ? l : 0);
    }
}

// ===================================================================================================== forward
// WPB waves per workgroup, each walking partitions pidx = blockIdx.x * WPB + wave, += gridDim.x * WPB.
template <class T, int C, int NB, bool LN, int WPB>
__global__ void __launch_bounds__(64 * WPB, (sizeof(T) == 2 && NB == 2 && C == 64 && WPB == 4) ? 2 : 1)
attn_block_fwd_kernel(const T* __restrict__ x, T* __restrict__ xmid, T* __restrict__ a_out, const float* __restrict__ ln_w,
                      const float* __restrict__ ln_b, const T* __restrict__ Wqkv, const float* __restrict__ bqkv,
                      const T* __restrict__ Wp, const float* __restrict__ bp, const float* __restrict__ gamma, AttnGeom g,
                      float eps) {
    typedef AbSmem<T, C> S;
    constexpr int KS = C / 16, HEADS = C / 32, NCB = C / 32;
    // LN: the raw rows are needed again for the residual.  They used to be re-read from L2 in the epilogue - but vmcnt retires loads
    // and stores in issue order, so that load waited for the acknowledgement of every attention-output row stored before it
    // (1.13 -> 1.00 ms per block with the stash).  bf16: the rows are stashed in lane-private LDS slots
    // (piece (b, ks) of lane l at ((b KS + ks) 64 + l) 16: no conflicts, no synchronisation) and come back over lgkmcnt.
    constexpr bool STASH = LN && sizeof(T) == 2;
    constexpr int STASH_B = STASH ? NB * KS * 64 * 16 : 0;
    __shared__ __attribute__((aligned(16))) char smem[S::OFF_S + WPB * STASH_B];
    char* const Wq_l = smem;
    char* const Wp_l = smem + S::OFF_P;
    float* const kst = reinterpret_cast<float*>(smem + S::OFF_K);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, half = lane >> 5, wave = tid >> 6;
    char* const stash = smem + S::OFF_S + wave * STASH_B + lane * 16;

    ab_stage_weights<T, C, false>(Wq_l, Wqkv, 3 * C, tid, 64 * WPB);
    ab_stage_weights<T, C, true>(Wp_l, Wp, C, tid, 64 * WPB);
    for (int i = tid; i < C; i += 64 * WPB) {
        kst[S::K_LNW + i] = LN ? ln_w[i] : 1.f;
        kst[S::K_LNB + i] = LN ? ln_b[i] : 0.f;
        kst[S::K_BP + i] = bp[i];
        kst[S::K_GAM + i] = gamma[i];
    }
    for (int i = tid; i < 3 * C; i += 64 * WPB) kst[S::K_BQKV + i] = bqkv[i];
    __syncthreads();

    const int klim = g.L - 32 * (NB - 1) - 4 * half;        // accumulator rows (r & 3) + 8 (r >> 2) >= klim of the last key block are padding
    const float scale_log2e = g.scale * 1.4426950408889634f;
    const int n_part = g.F * g.P;

    for (int pidx = blockIdx.x * WPB + wave; pidx < n_part; pidx += gridDim.x * WPB) {
        AbPart<NB> pt;
        ab_partition<NB>(pt, g, pidx, li);
        frag_t<T> uf[NB][KS];
        ab_load_rows<T, C, NB>(uf, x, pt.tok, pt.valid, half);
        if (STASH) {
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int ks = 0; ks < KS; ks++) *reinterpret_cast<frag_t<T>*>(stash + (b * KS + ks) * 1024) = uf[b][ks];
        }
        if (LN) {
            float mean[NB], rstd[NB];
            ab_layernorm<T, C, NB>(uf, uf, kst + S::K_LNW, kst + S::K_LNB, pt.valid, half, eps, mean, rstd);
        }
        // One query block at a time through all heads (only ONE block of output accumulators is live); K^T and V of a head
        // are recomputed per query block (16 more MFMAs per head and block) rather than held: the kernel is nowhere near
        // MFMA-bound, and the registers buy a second wave per SIMD at NB = 2.
#pragma unroll
        for (int bi = 0; bi < NB; bi++) {
            f32x16 oacc[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) acc_zero(oacc[cb]);
#pragma unroll 1
            for (int h = 0; h < HEADS; h++) {
                const int r0 = h * 96;
                frag_t<T> qf[2], kf[NB][2], vf[NB][2];
                // biases: q's is the initial value of its accumulator; k's is dropped (it shifts every score of a query by the
                // same amount: softmax is invariant to it); v's is added to the attention output instead (rows of P sum to 1)
                {
                    f32x16 acc;
                    acc_load_rows(acc, kst + S::K_BQKV + r0, half);
                    ab_proj_t<T, C>(acc, Wq_l, 3 * C, r0, uf[bi], li, half);
                    acc_to_frags<T>(acc, qf);
                }
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    f32x16 acc;
                    acc_zero(acc);
                    ab_proj_t<T, C>(acc, Wq_l, 3 * C, r0 + 32, uf[b], li, half);
                    acc_to_frags<T>(acc, kf[b]);
                    acc_zero(acc);
                    ab_proj_n<T, C>(acc, Wq_l, 3 * C, r0 + 64, uf[b], li, half);
                    acc_to_frags<T>(acc, vf[b]);
                }
                f32x16 s[NB];
#pragma unroll
                for (int bj = 0; bj < NB; bj++) {
                    acc_zero(s[bj]);
#pragma unroll
                    for (int q = 0; q < 2; q++) mma32(s[bj], kf[bj][q], qf[q]);
                }
                float pr[NB][16];
                const float inv = ab_softmax_cols<NB>(s, pr, klim, scale_log2e);
                f32x16 o;
                acc_zero(o);
#pragma unroll
                for (int bj = 0; bj < NB; bj++)
#pragma unroll
                    for (int q = 0; q < 2; q++) mma32(o, vf[bj][q], arr_slot_frag<T>(pr[bj], q));
                {
                    f32x16 bvr;                          // v bias of this lane's 16 accumulator rows (features d)
                    acc_load_rows(bvr, kst + S::K_BQKV + r0 + 64, half);
#pragma unroll
                    for (int r = 0; r < 16; r++) o[r] = fmaf(o[r], inv, bvr[r]);
                }
                frag_t<T> of[2];
                acc_to_frags<T>(o, of);
                if (a_out != nullptr) {       // attention output rows (the B operand of the proj weight gradient)
                    float r8[2][8];
                    acc_to_rows(o, r8);
                    if (pt.valid[bi]) {
#pragma unroll
                        for (int m = 0; m < 2; m++)
                            frag_store<T>(a_out + (size_t)pt.tok[bi] * C + h * 32 + 16 * m + 8 * half, frag_from_float<T>(r8[m]));
                    }
                }
#pragma unroll
                for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                    for (int q = 0; q < 2; q++)
                        mma32(oacc[cb], opm_load_frag<T>(Wp_l, C, cb * 32 + li, h * 4 + 2 * q + half), of[q]);
            }
            // ---- LayerScale + residual (maxvit.py:51-53,268) for this query block; the residual is the raw x row piece ----
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                float r8[2][8];
                acc_to_rows(oacc[cb], r8);
#pragma unroll
                for (int m = 0; m < 2; m++) {
                    const int ks = 2 * cb + m;
                    float res[8], gam[8], bpv[8], o[8];
                    const size_t off = (size_t)pt.tok[bi] * C + (2 * ks + half) * 8;
                    if (LN) {
                        const frag_t<T> xr = STASH ? *reinterpret_cast<const frag_t<T>*>(stash + (bi * KS + ks) * 1024)
                                                   : frag_load<T>(x + off);          // (fp32: L2 - this wave read the row a moment ago)
                        frag_to_float<T>(xr, res);
                    } else {
                        frag_to_float<T>(uf[bi][ks], res);
                    }
                    load_cols<8>(kst + S::K_GAM, 16 * ks + 8 * half, gam);
                    load_cols<8>(kst + S::K_BP, 16 * ks + 8 * half, bpv);
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = res[e] + gam[e] * (r8[m][e] + bpv[e]);
                    if (pt.valid[bi]) frag_store<T>(xmid + off, frag_from_float<T>(o));
                }
            }
        }
    }
}

// ==================================================================================================== backward
// Transposed operand piece in ACCUMULATOR order from a row-major LDS tile X[row][col] (opm layout, `rows` rows):
//   f[e] = X[row0 + (e & 3) + 8 (e >> 2) + 4 (lane >> 5)][col0 + (lane & 31)],  e = 0..7
// i.e. the operand "row = column col0 + lane&31, contraction over the 16 rows row0..row0+15" whose slot order matches
// accumulator registers 8 q .. 8 q + 7 of the other operand (row0 = 16 q within a 32-row block).
template <class T> __device__ __forceinline__ frag_t<T> ab_tr_frag(char* base, int rows, int row0, int col0, int lane) {
    if constexpr (sizeof(T) == 2) {
        const int rl = 4 * (lane >> 5) + ((lane & 15) >> 2), cl = col0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        return frag_from_tr<T>(reinterpret_cast<const bf16*>(opm_elem_ptr<T>(base, rows, row0 + rl, cl)),
                               reinterpret_cast<const bf16*>(opm_elem_ptr<T>(base, rows, row0 + rl + 8, cl)));
    } else {
        frag_t<T> f;
#pragma unroll
        for (int e = 0; e < 8; e++)
            f[e] = *opm_elem_ptr<T>(base, rows, row0 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col0 + (lane & 31));
        return f;
    }
}

// write a T-form accumulator block (col = lane & 31 = LDS row, registers = 32 columns col0..col0+31) as 4-element row pieces
template <class T> __device__ __forceinline__ void ab_acc_to_lds(char* base, int rows, int row, int col0, const float (&v)[16], int half) {
    typedef __attribute__((ext_vector_type(4))) T vec4;
#pragma unroll
    for (int gq = 0; gq < 4; gq++) {
        vec4 t;
#pragma unroll
        for (int w = 0; w < 4; w++) t[w] = (T)v[4 * gq + w];
        *reinterpret_cast<vec4*>(opm_elem_ptr<T>(base, rows, row, col0 + 8 * gq + 4 * half)) = t;
    }
}

template <class T, int NB> struct AbBwdScratch {          // per wave: P and dS blocks [32 queries][32 NB keys]
    static constexpr int KTJ = 32 * NB * (int)sizeof(T) / 128 + ((32 * NB * (int)sizeof(T)) % 128 ? 1 : 0);
    static constexpr int ONE = KTJ * 32 * 128;
    static constexpr int BYTES = 2 * ONE;
};

// dx = dxmid + LN1'( dqkv Wqkv ; x ),  dqkv = attention backward of da = dxmid (gamma Wp)   — dqkv (and u = LN1(x)) go to
// HBM for the qkv weight-gradient GEMM; LayerNorm parameter gradients leave as one atomic per channel per workgroup.
// (Measured and dropped in round 2: the qkv weight gradient accumulated inside this kernel — dqkv blocks transposed by
// identity MFMAs, multiplied with u in the same form, 32x32 results added into an LDS-resident fp32 image of dW with
// ds_add_f32: correct, but an LDS float atomic costs ~180 cycles per wave instruction on gfx950 — 16.7 ms against 2.7 ms for
// this kernel plus the weight-gradient GEMM.)
// PRE (with LN = false; round 6): the block input x is itself the output of a LayerNorm, x = LN_pre(y0) - the down-sampling
// norm in front of a stage's first block (maxvit.py:177, which is why that block has no norm1: `skip_first_norm`) - and the launch
// carries the gradient through it: `dx` receives dy0 = LN_pre'(dxmid + du ; y0) instead of dxmid + du, `ln_w` is THAT norm's weight,
// dln_w / dln_b its parameter gradients.  The attention part is the LN = false kernel unchanged (same dqkv bits); only the last
// step differs, where the y0 rows are read (one more row per token) - the stand-alone LayerNorm backward launch (three rows per
// token: 0.63 ms at stage 1 of RVT-Base) disappears.
template <class T, int C, int NB, bool LN, int WPB, bool PRE = false>
__global__ void __launch_bounds__(64 * WPB, 1)
attn_block_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dxmid, T* __restrict__ dx, T* __restrict__ dqkv,
                      T* __restrict__ u_out, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                      const T* __restrict__ Wqkv, const float* __restrict__ bqkv, const T* __restrict__ WpgT,
                      float* __restrict__ dln_w, float* __restrict__ dln_b, AttnGeom g, float eps, const T* __restrict__ y0) {
    static_assert(!(PRE && LN), "PRE carries the gradient through the norm in FRONT of a block without norm1");
    typedef AbSmem<T, C> S;
    typedef AbBwdScratch<T, NB> SC;
    constexpr int KS = C / 16, HEADS = C / 32, NCB = C / 32, LP = 32 * NB;
    // per wave: P / dS scratch
    constexpr int DLN = 0;
    constexpr bool STASH = (LN || PRE) && sizeof(T) == 2;              // raw rows for the LayerNorm backward: lane-private LDS slots (see the forward)
    constexpr int STASH_B = STASH ? NB * KS * 64 * 16 : 0;
    constexpr int OFF_STASH = S::OFF_S + WPB * (SC::BYTES + DLN) + ((LN || PRE) ? WPB * 2 * C * 4 : 0);
    __shared__ __attribute__((aligned(16))) char smem[OFF_STASH + WPB * STASH_B];
    char* const Wq_l = smem;
    char* const Wp_l = smem + S::OFF_P;
    float* const kst = reinterpret_cast<float*>(smem + S::OFF_K);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, half = lane >> 5, wave = tid >> 6;
    char* const Pl = smem + S::OFF_S + wave * (SC::BYTES + DLN);
    char* const stash = smem + OFF_STASH + wave * STASH_B + lane * 16;
    char* const dSl = Pl + SC::ONE;
    // LayerNorm parameter gradients = column sums over tokens of du * xhat and du: the row pieces hold tokens in the LANES; an
    // MFMA against an identity operand turns a piece block into "col = channel, registers = tokens" (exact), where the column
    // sum is an in-lane sum — one running value per 32-channel block instead of per-lane partial sums for every channel
    frag_t<T> idf[2];
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int e = 0; e < 8; e++) idf[m][e] = (T)((16 * m + 8 * half + e == li) ? 1.0f : 0.0f);
    float aw[NCB], ab[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { aw[cb] = 0.f; ab[cb] = 0.f; }

    ab_stage_weights<T, C, false>(Wq_l, Wqkv, 3 * C, tid, 64 * WPB);
    ab_stage_weights<T, C, false>(Wp_l, WpgT, C, tid, 64 * WPB);
    for (int i = tid; i < C; i += 64 * WPB) {
        kst[S::K_LNW + i] = (LN || PRE) ? ln_w[i] : 1.f;
        kst[S::K_LNB + i] = LN ? ln_b[i] : 0.f;
    }
    for (int i = tid; i < 3 * C; i += 64 * WPB) kst[S::K_BQKV + i] = bqkv[i];
    __syncthreads();

    const int klim = g.L - 32 * (NB - 1) - 4 * half;        // accumulator rows (r & 3) + 8 (r >> 2) >= klim of the last key block are padding
    const float scale_log2e = g.scale * 1.4426950408889634f;
    const int n_part = g.F * g.P;

    for (int pidx = blockIdx.x * WPB + wave; pidx < n_part; pidx += gridDim.x * WPB) {
        AbPart<NB> pt;
        ab_partition<NB>(pt, g, pidx, li);
        frag_t<T> uf[NB][KS], df[NB][KS];
        float mean[NB], rstd[NB];
        ab_load_rows<T, C, NB>(uf, x, pt.tok, pt.valid, half);
        ab_load_rows<T, C, NB>(df, dxmid, pt.tok, pt.valid, half);
        if (STASH && !PRE) {
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int ks = 0; ks < KS; ks++) *reinterpret_cast<frag_t<T>*>(stash + (b * KS + ks) * 1024) = uf[b][ks];
        }
        if (PRE) {
            // the rows of the norm IN FRONT of the block: requested here, beside the x / dxmid rows (at the end of the partition the
            // wave has nothing else in flight to hide an HBM round trip behind: +0.47 ms measured), statistics now (lane = row:
            // in-lane sums + one exchange), the rows themselves parked in the lane-private LDS stash until the last step
            frag_t<T> yf[NB][KS];
            ab_load_rows<T, C, NB>(yf, y0, pt.tok, pt.valid, half);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                float sm = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ks++)
#pragma unroll
                    for (int e = 0; e < 8; e++) sm += (float)yf[b][ks][e];
                sm += __shfl_xor(sm, 32);
                mean[b] = sm / (float)C;
                float q = 0.f;
#pragma unroll
                for (int ks = 0; ks < KS; ks++)
#pragma unroll
                    for (int e = 0; e < 8; e++) { const float dd = (float)yf[b][ks][e] - mean[b]; q += dd * dd; }
                q += __shfl_xor(q, 32);
                rstd[b] = 1.0f / sqrtf(q / (float)C + eps);
                if (STASH) {
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) *reinterpret_cast<frag_t<T>*>(stash + (b * KS + ks) * 1024) = yf[b][ks];
                }
            }
        }
        if (LN) {       // the raw rows come back from the LDS stash (fp32: L2) for the LayerNorm backward at the end: 32 registers less across the heads
            ab_layernorm<T, C, NB>(uf, uf, kst + S::K_LNW, kst + S::K_LNB, pt.valid, half, eps, mean, rstd);
            if (u_out != nullptr) {
#pragma unroll
                for (int b = 0; b < NB; b++)
#pragma unroll
                    for (int ks = 0; ks < KS; ks++)
                        if (pt.valid[b]) frag_store<T>(u_out + (size_t)pt.tok[b] * C + (2 * ks + half) * 8, uf[b][ks]);
            }
        }
        f32x16 du[NCB][NB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++)
#pragma unroll
            for (int b = 0; b < NB; b++) acc_zero(du[cb][b]);

        // du^T[c][tok] += sum_n Wqkv[n0 + n][c] dz[tok][n]  for one 32-feature block n0 of dqkv held T-form in `z` (token block b);
        // also writes the block's rows to HBM.  A = W^T pieces through the transposing LDS read, in accumulator order.
        auto emit_dqkv = [&](const f32x16& z, int b, int n0) {
            float r8[2][8];
            acc_to_rows(z, r8);
            if (pt.valid[b]) {
#pragma unroll
                for (int m = 0; m < 2; m++)
                    frag_store<T>(dqkv + (size_t)pt.tok[b] * (3 * C) + n0 + 16 * m + 8 * half, frag_from_float<T>(r8[m]));
            }
            frag_t<T> zf[2];
            acc_to_frags<T>(z, zf);
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                for (int q = 0; q < 2; q++) mma32(du[cb][b], ab_tr_frag<T>(Wq_l, 3 * C, n0 + 16 * q, cb * 32, lane), zf[q]);
        };

#pragma unroll 1
        for (int h = 0; h < HEADS; h++) {
            const int r0 = h * 96;
            // key side, all blocks: K, V T-form (operand "row token, contract d"), K N-form (operand "row d, contract token")
            frag_t<T> kf[NB][2], vtf[NB][2], kn[NB][2];
#pragma unroll
            for (int b = 0; b < NB; b++) {
                // (k bias dropped: S, P, and with them every gradient but its own — which is identically zero — are invariant to it)
                f32x16 acc;
                acc_zero(acc);
                ab_proj_t<T, C>(acc, Wq_l, 3 * C, r0 + 32, uf[b], li, half);
                acc_to_frags<T>(acc, kf[b]);
                acc_load_rows(acc, kst + S::K_BQKV + r0 + 64, half);
                ab_proj_t<T, C>(acc, Wq_l, 3 * C, r0 + 64, uf[b], li, half);
                acc_to_frags<T>(acc, vtf[b]);
                acc_zero(acc);
                ab_proj_n<T, C>(acc, Wq_l, 3 * C, r0 + 32, uf[b], li, half);
                acc_to_frags<T>(acc, kn[b]);
            }
            f32x16 dk[NB], dv[NB];
#pragma unroll
            for (int b = 0; b < NB; b++) { acc_zero(dk[b]); acc_zero(dv[b]); }

#pragma unroll
            for (int bi = 0; bi < NB; bi++) {
                // query side, this block: Q, dO in both forms
                frag_t<T> qf[2], dotf[2], qn[2], don[2];
                {
                    f32x16 acc;
                    acc_load_rows(acc, kst + S::K_BQKV + r0, half);
                    ab_proj_t<T, C>(acc, Wq_l, 3 * C, r0, uf[bi], li, half);
                    acc_to_frags<T>(acc, qf);
                    acc_zero(acc);
                    ab_proj_t<T, C>(acc, Wp_l, C, h * 32, df[bi], li, half);
                    acc_to_frags<T>(acc, dotf);
                    const float bv = kst[S::K_BQKV + r0 + li];
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[r] = bv;
                    ab_proj_n<T, C>(acc, Wq_l, 3 * C, r0, uf[bi], li, half);
                    acc_to_frags<T>(acc, qn);
                    acc_zero(acc);
                    ab_proj_n<T, C>(acc, Wp_l, C, h * 32, df[bi], li, half);
                    acc_to_frags<T>(acc, don);
                }
                f32x16 s[NB], dp[NB];
#pragma unroll
                for (int bj = 0; bj < NB; bj++) {
                    acc_zero(s[bj]);
                    acc_zero(dp[bj]);
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        mma32(s[bj], kf[bj][q], qf[q]);                  // S^T[j][i]
                        mma32(dp[bj], vtf[bj][q], dotf[q]);              // dP^T[j][i] = sum_d V[j][d] dO[i][d]
                    }
                }
                float pr[NB][16];
                const float inv = ab_softmax_cols<NB>(s, pr, klim, scale_log2e);
                float delta = 0.f;
#pragma unroll
                for (int bj = 0; bj < NB; bj++)
#pragma unroll
                    for (int r = 0; r < 16; r++) { pr[bj][r] *= inv; delta += pr[bj][r] * dp[bj][r]; }
                delta += __shfl_xor(delta, 32);
                float ds[NB][16];
#pragma unroll
                for (int bj = 0; bj < NB; bj++)
#pragma unroll
                    for (int r = 0; r < 16; r++) ds[bj][r] = pr[bj][r] * (dp[bj][r] - delta) * g.scale;
                // P, dS of this query block -> LDS, row = query (this lane), 4 consecutive keys per write
#pragma unroll
                for (int bj = 0; bj < NB; bj++) {
                    ab_acc_to_lds<T>(Pl, 32, li, 32 * bj, pr[bj], half);
                    ab_acc_to_lds<T>(dSl, 32, li, 32 * bj, ds[bj], half);
                }
                // dQ^T[d][i] = sum_j K[j][d] dS[i][j]   (A = K N-form, B = dS^T from the registers)
                f32x16 dq;
                acc_zero(dq);
#pragma unroll
                for (int bj = 0; bj < NB; bj++)
#pragma unroll
                    for (int q = 0; q < 2; q++) mma32(dq, kn[bj][q], arr_slot_frag<T>(ds[bj], q));
                emit_dqkv(dq, bi, r0);
                wave_lds_sync();
                // dV^T[d][j] += sum_i dO[i][d] P[i][j];  dK^T[d][j] += sum_i Q[i][d] dS[i][j]   (contraction over this query block)
#pragma unroll
                for (int bj = 0; bj < NB; bj++)
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        mma32(dv[bj], don[q], ab_tr_frag<T>(Pl, 32, 16 * q, 32 * bj, lane));
                        mma32(dk[bj], qn[q], ab_tr_frag<T>(dSl, 32, 16 * q, 32 * bj, lane));
                    }
                wave_lds_sync();          // the next query block overwrites P / dS
            }
#pragma unroll
            for (int bj = 0; bj < NB; bj++) {
                emit_dqkv(dk[bj], bj, r0 + 32);
                emit_dqkv(dv[bj], bj, r0 + 64);
            }
        }

        // ---- dx = dxmid + LN1'(du)  (maxvit.py:229,268) in operand-piece form ----
#pragma unroll
        for (int b = 0; b < NB; b++) {
            float d8[KS][8];
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                float r8[2][8];
                acc_to_rows(du[cb][b], r8);
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int e = 0; e < 8; e++) d8[2 * cb + m][e] = pt.valid[b] ? r8[m][e] : 0.f;
            }
            if (LN || PRE) {
                float gsum = 0.f, gxsum = 0.f;
                float xh[KS][8];
                if (PRE) {             // the cotangent that enters the norm: g = dxmid + du
#pragma unroll
                    for (int ks = 0; ks < KS; ks++)
#pragma unroll
                        for (int e = 0; e < 8; e++) d8[ks][e] += (float)df[b][ks][e];
                }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    float w[8];
                    load_cols<8>(kst + S::K_LNW, 16 * ks + 8 * half, w);
                    const frag_t<T> xr = STASH ? *reinterpret_cast<const frag_t<T>*>(stash + (b * KS + ks) * 1024)
                                               : frag_load<T>((PRE ? y0 : x) + (size_t)pt.tok[b] * C + (2 * ks + half) * 8);
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        xh[ks][e] = pt.valid[b] ? ((float)xr[e] - mean[b]) * rstd[b] : 0.f;
                        const float gw = d8[ks][e] * w[e];
                        gsum += gw;
                        gxsum += gw * xh[ks][e];
                    }
                }
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
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
                    for (int e = 0; e < 8; e++)
                        o[e] = (PRE ? 0.f : (float)df[b][ks][e]) + rstd[b] * (d8[ks][e] * w[e] - m1 - xh[ks][e] * m2);
                    if (pt.valid[b]) frag_store<T>(dx + (size_t)pt.tok[b] * C + (2 * ks + half) * 8, frag_from_float<T>(o));
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) o[e] = (float)df[b][ks][e] + d8[ks][e];
                    if (pt.valid[b]) frag_store<T>(dx + (size_t)pt.tok[b] * C + (2 * ks + half) * 8, frag_from_float<T>(o));
                }
            }
        }
    }
    if (LN || PRE) {
        // fold the two halves and the waves: one atomic per channel per workgroup
        float* const red = reinterpret_cast<float*>(smem + S::OFF_S + WPB * (SC::BYTES + DLN));      // [WPB][dln_w C | dln_b C]
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
}

}  // namespace rvt
