// Partitioned multi-head self-attention core (reference maxvit.py:343-354 minus the two linears, on the partitions of
// :273-304) for the stages whose attention half is not fused (attn_block.hpp): same contract as the kernels of attn.hpp
// (qkv [F*H*W][3C] in image token order, per-head [q|k|v]; one wave per (frame, partition, head)), rebuilt on the
// accumulator-to-operand chaining of attn_block.hpp:
//   * Q, K, V, dO rows come from HBM in operand form = the "row token, contract d" operands of S^T = K Q^T and dP^T = V dO^T;
//   * the "row d, contract token" operands (V for O^T = V^T P^T; K, Q, dO for dQ^T, dK^T, dV^T) are transposes of those rows:
//     an MFMA against an identity operand (exact) instead of a transposed LDS copy built with 2-byte stores and read back with
//     2-byte loads — the forward touches no LDS at all;
//   * P and dS (contraction over queries for dV, dK) take one trip through a wave-private LDS tile as 8-byte row pieces and
//     come back through the transposing LDS read;
//   * dQ^T, dK^T, dV^T leave as 16-byte row pieces (v_permlane32_swap) instead of being staged through LDS.
#pragma once
#include "attn_block.hpp"
#include "line_bounce.hpp"

namespace rvt {

// identity operand pieces of a 32-column block: row n = lane & 31, k-step cc: 1 at slot k = 16 cc + 8 half + e == n
template <class T> __device__ __forceinline__ void make_identity_frags(frag_t<T> (&idf)[2], int li, int half) {
#pragma unroll
    for (int cc = 0; cc < 2; cc++)
#pragma unroll
        for (int e = 0; e < 8; e++) idf[cc][e] = (T)((16 * cc + 8 * half + e == li) ? 1.0f : 0.0f);
}
// rows held in operand form (lane = row, pieces cc = 0, 1) -> the transposed operand "row = column, contract over the 32 rows"
template <class T> __device__ __forceinline__ void transpose_rows(const frag_t<T> (&rows)[2], const frag_t<T> (&idf)[2], frag_t<T> (&out)[2]) {
    f32x16 acc;
    acc_zero(acc);
#pragma unroll
    for (int cc = 0; cc < 2; cc++) mma32(acc, rows[cc], idf[cc]);
    out[0] = acc_slot_frag<T>(acc, 0);
    out[1] = acc_slot_frag<T>(acc, 1);
}

struct AcPart {
    int f, p, head, qoff;
};
__device__ __forceinline__ AcPart ac_partition(const AttnGeom& g, int HG, int wv) {
    uint32_t fp, grp, f, p;
    g.dGroups.divmod(blockIdx.x, fp, grp);
    g.dP.divmod(fp, f, p);
    AcPart a;
    a.f = (int)f; a.p = (int)p; a.head = (int)grp * HG + wv; a.qoff = a.head * 3 * g.dh;
    return a;
}

template <class T, int NB, int HG>
// (four waves per SIMD — 117 registers instead of 162 — measured: no gain, 502 vs 478 us on the stage-2 shape)
__global__ void __launch_bounds__(64 * HG)
attn_core_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, AttnGeom g) {
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
    const AcPart a = ac_partition(g, HG, wv);
    const int C3 = 3 * g.C, dh = g.dh;
    int tok[NB]; bool valid[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int l = 32 * b + li;
        valid[b] = l < g.L;
        tok[b] = attn_token(g, a.f, a.p, valid[b] ? l : 0);
    }
    const int klim = g.L - 32 * (NB - 1) - 4 * half;
    const float scale_log2e = g.scale * 1.4426950408889634f;
    frag_t<T> qf[NB][2], kf[NB][2], vr[NB][2];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const int chunk = half + 2 * cc;
            const T* row = qkv + (size_t)tok[b] * C3 + a.qoff;
            qf[b][cc] = load_chunk<T>(row, chunk, dh, valid[b]);
            kf[b][cc] = load_chunk<T>(row + dh, chunk, dh, valid[b]);
            vr[b][cc] = load_chunk<T>(row + 2 * dh, chunk, dh, valid[b]);
        }
    frag_t<T> idf[2], vf[NB][2];
    make_identity_frags<T>(idf, li, half);
#pragma unroll
    for (int b = 0; b < NB; b++) transpose_rows<T>(vr[b], idf, vf[b]);      // V^T: "row d, contract keys"

#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
        f32x16 s[NB];
#pragma unroll
        for (int bj = 0; bj < NB; bj++) {
            acc_zero(s[bj]);
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
                if (ks * 16 < dh) mma32(s[bj], kf[bj][ks], qf[bi][ks]);
        }
        float pr[NB][16];
        const float inv = ab_softmax_cols<NB>(s, pr, klim, scale_log2e);
        f32x16 o;
        acc_zero(o);
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int q = 0; q < 2; q++) mma32(o, vf[bj][q], arr_slot_frag<T>(pr[bj], q));
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] *= inv;
        float r8[2][8];
        acc_to_rows(o, r8);
        if (valid[bi]) {
#pragma unroll
            for (int m = 0; m < 2; m++)
                if (16 * m + 8 * half < dh)
                    frag_store<T>(out + (size_t)tok[bi] * g.C + a.head * dh + 16 * m + 8 * half, frag_from_float<T>(r8[m]));
        }
    }
}

// The same forward with the rows STAGED through LDS (bf16, partitions of 33..96 tokens, dim_head 32).  In the kernel above a lane owns a
// token and fetches its q / k / v as 16-byte chunks: 32 bytes of 32 different lines per load instruction, and the output rows leave
// the same way - four times the memory requests of whole lines.  With every load and store instruction of the kernel above replaced by a
// contiguous 1-KiB access (wrong results, same bytes) the stage-2 launch takes 0.37 instead of 0.48 ms.  Here the 192 contiguous bytes
// [q | k | v] of a (token, head) arrive by LDS-DMA into a wave-private tile [32 NB tokens][192 B] (lane-linear destination = the tile
// itself; piece p = 64 i + lane is token p / 12, 16-byte piece p % 12 - the token's row offset comes from the lane that computed it by
// a wave shuffle; tokens beyond L get an out-of-range offset = zeros), the operand fragments are plain 16-byte LDS reads, and the
// output rows bounce through the same tile and leave as 64-byte pieces of 16 tokens per store instruction.
template <int NB, int HG>
__global__ void __launch_bounds__(64 * HG)
attn_core_fwd_staged_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, AttnGeom g, unsigned qkv_bytes, unsigned out_bytes) {
    typedef bf16 T;
    static_assert(NB == 2 || NB == 3, "partitions of 33 .. 96 tokens");
    constexpr int TILE = 32 * NB * 192, NDMA = 32 * NB * 12 / 64;
    __shared__ __attribute__((aligned(16))) char smem[HG * TILE];
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
    const int wv = wave_uniform((int)threadIdx.x >> 6);
    char* const tile = smem + wv * TILE;
    const AcPart a = ac_partition(g, HG, wv);
    // byte offsets in qkv / out of the partition's token slots: slot l = lane in (rowq, rowo), slot 64 + lane (lanes 0 .. 31, NB = 3) in (rowq2, rowo2)
    const bool lv = lane < g.L, lv2 = NB == 3 && 64 + lane < g.L;
    const int tk = attn_token(g, a.f, a.p, lv ? lane : 0), tk2 = NB == 3 ? attn_token(g, a.f, a.p, lv2 ? 64 + lane : 0) : 0;
    const int rowq = lv ? tk * (3 * g.C * 2) + a.qoff * 2 : 0x7ffff000, rowq2 = lv2 ? tk2 * (3 * g.C * 2) + a.qoff * 2 : 0x7ffff000;
    const int rowo = lv ? tk * (g.C * 2) + a.head * 64 : 0x7ffff000, rowo2 = lv2 ? tk2 * (g.C * 2) + a.head * 64 : 0x7ffff000;
    auto slot = [&](int v, int v2, int t) __attribute__((always_inline)) {     // the offset of token slot t, from the lane that holds it
        const int lo = __shfl(v, t & 63);
        if (NB == 2) return lo;
        const int hi = __shfl(v2, t & 31);
        return t < 64 ? lo : hi;
    };
    {
        const pp_rsrc rq = pp_make_rsrc(qkv, qkv_bytes);
#pragma unroll
        for (int i = 0; i < NDMA; i++) {
            const int p = 64 * i + lane, t = (p * 5462) >> 16, w = p - 12 * t;       // p / 12 for p < 1152
            pp_glds16(rq, smem, wv * TILE + i * 1024, slot(rowq, rowq2, t) + w * 16, 0);
        }
    }
    const int klim = g.L - 32 * (NB - 1) - 4 * half;
    const float scale_log2e = g.scale * 1.4426950408889634f;
    frag_t<T> idf[2];
    make_identity_frags<T>(idf, li, half);
    pp_wait_vm<0>();
    wave_rendezvous();
    frag_t<T> qf[NB][2], kf[NB][2], vr[NB][2];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const char* const base = tile + (32 * b + li) * 192 + (half + 2 * cc) * 16;
            qf[b][cc] = *reinterpret_cast<const frag_t<T>*>(base);
            kf[b][cc] = *reinterpret_cast<const frag_t<T>*>(base + 64);
            vr[b][cc] = *reinterpret_cast<const frag_t<T>*>(base + 128);
        }
    frag_t<T> vf[NB][2];
#pragma unroll
    for (int b = 0; b < NB; b++) transpose_rows<T>(vr[b], idf, vf[b]);      // V^T: "row d, contract keys"
    const pp_rsrc ro = pp_make_rsrc(out, out_bytes);
#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
        f32x16 s[NB];
#pragma unroll
        for (int bj = 0; bj < NB; bj++) {
            acc_zero(s[bj]);
#pragma unroll
            for (int ks = 0; ks < 2; ks++) mma32(s[bj], kf[bj][ks], qf[bi][ks]);
        }
        float pr[NB][16];
        const float inv = ab_softmax_cols<NB>(s, pr, klim, scale_log2e);
        f32x16 o;
        acc_zero(o);
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int q = 0; q < 2; q++) mma32(o, vf[bj][q], arr_slot_frag<T>(pr[bj], q));
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] *= inv;
        float r8[2][8];
        acc_to_rows(o, r8);
        // rows of this query block: [32 tokens][64 B] through the tile (its q / k / v are in registers by now), back as
        // lane = (token lane / 4, piece lane % 4): 16 tokens x 64 bytes per store instruction
        wave_rendezvous();
#pragma unroll
        for (int m = 0; m < 2; m++) *reinterpret_cast<frag_t<T>*>(tile + li * 64 + (((2 * m + half) ^ (li & 3)) << 4)) = frag_from_float<T>(r8[m]);
        wave_rendezvous();
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int t = 16 * it + (lane >> 2), q = lane & 3;
            const u32x4 v = *reinterpret_cast<const u32x4*>(tile + t * 64 + ((q ^ (t & 3)) << 4));
            pp_store16(ro, slot(rowo, rowo2, 32 * bi + t) + q * 16, v);
        }
    }
}

// Backward: recomputes S / P from the saved qkv.  dP^T = V dO^T; delta_i = sum_j P dP; dS^T = P^T (dP^T - delta) scale;
// dQ^T = K^T dS^T; dV^T = dO^T P; dK^T = Q^T dS (the last two contract over the queries: P, dS through the LDS tile).
template <class T, int NB, int HG>
__global__ void __launch_bounds__(64 * HG, NB == 3 ? 1 : 2)
attn_core_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv, AttnGeom g) {
    typedef AbBwdScratch<T, NB> SC;
    __shared__ __attribute__((aligned(16))) char smem[HG * SC::BYTES];
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
    char* const Pl = smem + wv * SC::BYTES;
    char* const dSl = Pl + SC::ONE;
    const AcPart a = ac_partition(g, HG, wv);
    const int C3 = 3 * g.C, dh = g.dh;
    int tok[NB]; bool valid[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        const int l = 32 * b + li;
        valid[b] = l < g.L;
        tok[b] = attn_token(g, a.f, a.p, valid[b] ? l : 0);
    }
    const int klim = g.L - 32 * (NB - 1) - 4 * half;
    const float scale_log2e = g.scale * 1.4426950408889634f;
    // every global read of the kernel in one batch: rows in operand form = the T-form operands
    frag_t<T> qf[NB][2], kf[NB][2], vf[NB][2], df[NB][2];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const int chunk = half + 2 * cc;
            const T* row = qkv + (size_t)tok[b] * C3 + a.qoff;
            qf[b][cc] = load_chunk<T>(row, chunk, dh, valid[b]);
            kf[b][cc] = load_chunk<T>(row + dh, chunk, dh, valid[b]);
            vf[b][cc] = load_chunk<T>(row + 2 * dh, chunk, dh, valid[b]);
            df[b][cc] = load_chunk<T>(dout + (size_t)tok[b] * g.C + a.head * dh, chunk, dh, valid[b]);
        }
    frag_t<T> idf[2], kn[NB][2];
    make_identity_frags<T>(idf, li, half);
#pragma unroll
    for (int b = 0; b < NB; b++) transpose_rows<T>(kf[b], idf, kn[b]);
    auto store_rows = [&](const f32x16& z, int b, int off) {       // T-form block (lane = token) -> 16-byte row pieces
        float r8[2][8];
        acc_to_rows(z, r8);
        if (valid[b]) {
#pragma unroll
            for (int m = 0; m < 2; m++)
                if (16 * m + 8 * half < dh)
                    frag_store<T>(dqkv + (size_t)tok[b] * C3 + a.qoff + off + 16 * m + 8 * half, frag_from_float<T>(r8[m]));
        }
    };
    f32x16 dk[NB], dv[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) { acc_zero(dk[b]); acc_zero(dv[b]); }

#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
        f32x16 s[NB], dp[NB];
#pragma unroll
        for (int bj = 0; bj < NB; bj++) {
            acc_zero(s[bj]);
            acc_zero(dp[bj]);
#pragma unroll
            for (int ks = 0; ks < 2; ks++)
                if (ks * 16 < dh) {
                    mma32(s[bj], kf[bj][ks], qf[bi][ks]);
                    mma32(dp[bj], vf[bj][ks], df[bi][ks]);
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
#pragma unroll
        for (int bj = 0; bj < NB; bj++) {
            ab_acc_to_lds<T>(Pl, 32, li, 32 * bj, pr[bj], half);
            ab_acc_to_lds<T>(dSl, 32, li, 32 * bj, ds[bj], half);
        }
        f32x16 dq;
        acc_zero(dq);
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int q = 0; q < 2; q++) mma32(dq, kn[bj][q], arr_slot_frag<T>(ds[bj], q));
        store_rows(dq, bi, 0);
        frag_t<T> qn[2], don[2];                         // this query block's Q and dO as "row d, contract queries" operands
        transpose_rows<T>(qf[bi], idf, qn);
        transpose_rows<T>(df[bi], idf, don);
        wave_lds_sync();
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                mma32(dv[bj], don[q], ab_tr_frag<T>(Pl, 32, 16 * q, 32 * bj, lane));
                mma32(dk[bj], qn[q], ab_tr_frag<T>(dSl, 32, 16 * q, 32 * bj, lane));
            }
        wave_lds_sync();
    }
#pragma unroll
    for (int bj = 0; bj < NB; bj++) {
        store_rows(dk[bj], bj, dh);
        store_rows(dv[bj], bj, 2 * dh);
    }
}

// The backward with its rows staged the same way: [q | k | v] (192 B per token and head) and dO (64 B) arrive by LDS-DMA into a
// wave-private 16-KiB region whose head is reused for the P / dS tiles once the fragments are in registers; dQ, dK, dV leave through a
// 2-KiB bounce as 64-byte pieces of 16 tokens per store instruction.  The arithmetic is that of attn_core_bwd_kernel, value for value.
template <int NB, int HG>
__global__ void __launch_bounds__(64 * HG, NB == 3 ? 1 : 2)
attn_core_bwd_staged_kernel(const bf16* __restrict__ qkv, const bf16* __restrict__ dout, bf16* __restrict__ dqkv, AttnGeom g,
                            unsigned qkv_bytes, unsigned out_bytes) {
    typedef bf16 T;
    static_assert(NB == 2 || NB == 3, "partitions of 33 .. 96 tokens");
    typedef AbBwdScratch<T, NB> SC;
    constexpr int QKV_T = 32 * NB * 192, DO_T = 32 * NB * 64, REGION = QKV_T + DO_T;       // 16 (24) KiB per wave
    static_assert(SC::BYTES + 32 * 64 <= REGION, "P / dS tiles + the store bounce reuse the staging region");
    __shared__ __attribute__((aligned(16))) char smem[HG * REGION];
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
    const int wv = wave_uniform((int)threadIdx.x >> 6);
    char* const reg = smem + wv * REGION;
    char* const Pl = reg;
    char* const dSl = reg + SC::ONE;
    char* const bnc = reg + SC::BYTES;                    // [32 tokens][64 B]
    const AcPart a = ac_partition(g, HG, wv);
    const int C3 = 3 * g.C;
    // byte offsets of the partition's token slots (see the forward)
    const bool lv = lane < g.L, lv2 = NB == 3 && 64 + lane < g.L;
    const int tk = attn_token(g, a.f, a.p, lv ? lane : 0), tk2 = NB == 3 ? attn_token(g, a.f, a.p, lv2 ? 64 + lane : 0) : 0;
    const int rowq = lv ? tk * (C3 * 2) + a.qoff * 2 : 0x7ffff000, rowq2 = lv2 ? tk2 * (C3 * 2) + a.qoff * 2 : 0x7ffff000;
    const int rowo = lv ? tk * (g.C * 2) + a.head * 64 : 0x7ffff000, rowo2 = lv2 ? tk2 * (g.C * 2) + a.head * 64 : 0x7ffff000;
    auto slot = [&](int v, int v2, int t) __attribute__((always_inline)) {
        const int lo = __shfl(v, t & 63);
        if (NB == 2) return lo;
        const int hi = __shfl(v2, t & 31);
        return t < 64 ? lo : hi;
    };
    {
        const pp_rsrc rq = pp_make_rsrc(qkv, qkv_bytes), rd = pp_make_rsrc(dout, out_bytes);
#pragma unroll
        for (int i = 0; i < QKV_T / 1024; i++) {
            const int p = 64 * i + lane, t = (p * 5462) >> 16, w = p - 12 * t;       // p / 12 for p < 1152
            pp_glds16(rq, smem, wv * REGION + i * 1024, slot(rowq, rowq2, t) + w * 16, 0);
        }
#pragma unroll
        for (int i = 0; i < DO_T / 1024; i++) {
            const int p = 64 * i + lane;
            pp_glds16(rd, smem, wv * REGION + QKV_T + i * 1024, slot(rowo, rowo2, p >> 2) + (p & 3) * 16, 0);
        }
    }
    const int klim = g.L - 32 * (NB - 1) - 4 * half;
    const float scale_log2e = g.scale * 1.4426950408889634f;
    frag_t<T> idf[2];
    make_identity_frags<T>(idf, li, half);
    pp_wait_vm<0>();
    wave_rendezvous();
    frag_t<T> qf[NB][2], kf[NB][2], vf[NB][2], df[NB][2];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const char* const base = reg + (32 * b + li) * 192 + (half + 2 * cc) * 16;
            qf[b][cc] = *reinterpret_cast<const frag_t<T>*>(base);
            kf[b][cc] = *reinterpret_cast<const frag_t<T>*>(base + 64);
            vf[b][cc] = *reinterpret_cast<const frag_t<T>*>(base + 128);
            df[b][cc] = *reinterpret_cast<const frag_t<T>*>(reg + QKV_T + (32 * b + li) * 64 + (half + 2 * cc) * 16);
        }
    wave_rendezvous();                                     // every lane has its fragments: the region is free for P / dS / the bounce
    frag_t<T> kn[NB][2];
#pragma unroll
    for (int b = 0; b < NB; b++) transpose_rows<T>(kf[b], idf, kn[b]);
    const pp_rsrc ro = pp_make_rsrc(dqkv, qkv_bytes);
    // T-form block (lane = token of block b) -> [32 tokens][64 B] bounce -> 16 tokens x 64 bytes per store instruction, at byte `off` of the head's [dq | dk | dv]
    auto store_rows = [&](const f32x16& z, int b, int off) __attribute__((always_inline)) {
        float r8[2][8];
        acc_to_rows(z, r8);
        wave_rendezvous();
#pragma unroll
        for (int m = 0; m < 2; m++) *reinterpret_cast<frag_t<T>*>(bnc + li * 64 + (((2 * m + half) ^ (li & 3)) << 4)) = frag_from_float<T>(r8[m]);
        wave_rendezvous();
#pragma unroll
        for (int it = 0; it < 2; it++) {
            const int t = 16 * it + (lane >> 2), q = lane & 3;
            const u32x4 v = *reinterpret_cast<const u32x4*>(bnc + t * 64 + ((q ^ (t & 3)) << 4));
            pp_store16(ro, slot(rowq, rowq2, 32 * b + t) + off + q * 16, v);
        }
    };
    f32x16 dk[NB], dv[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) { acc_zero(dk[b]); acc_zero(dv[b]); }
#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
        f32x16 s[NB], dp[NB];
#pragma unroll
        for (int bj = 0; bj < NB; bj++) {
            acc_zero(s[bj]);
            acc_zero(dp[bj]);
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                mma32(s[bj], kf[bj][ks], qf[bi][ks]);
                mma32(dp[bj], vf[bj][ks], df[bi][ks]);
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
#pragma unroll
        for (int bj = 0; bj < NB; bj++) {
            ab_acc_to_lds<T>(Pl, 32, li, 32 * bj, pr[bj], half);
            ab_acc_to_lds<T>(dSl, 32, li, 32 * bj, ds[bj], half);
        }
        f32x16 dq;
        acc_zero(dq);
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int q = 0; q < 2; q++) mma32(dq, kn[bj][q], arr_slot_frag<T>(ds[bj], q));
        store_rows(dq, bi, 0);
        frag_t<T> qn[2], don[2];                         // this query block's Q and dO as "row d, contract queries" operands
        transpose_rows<T>(qf[bi], idf, qn);
        transpose_rows<T>(df[bi], idf, don);
        wave_lds_sync();
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                mma32(dv[bj], don[q], ab_tr_frag<T>(Pl, 32, 16 * q, 32 * bj, lane));
                mma32(dk[bj], qn[q], ab_tr_frag<T>(dSl, 32, 16 * q, 32 * bj, lane));
            }
        wave_lds_sync();
    }
#pragma unroll
    for (int bj = 0; bj < NB; bj++) {
        store_rows(dk[bj], bj, 64);
        store_rows(dv[bj], bj, 128);
    }
}

}  // namespace rvt
