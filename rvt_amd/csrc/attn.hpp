// Partitioned multi-head self-attention core (reference maxvit.py:343-354 on the partitions of
// maxvit.py:273-304), forward and backward, one 64-lane wave per (frame, partition, head); a workgroup = HG waves = HG
// consecutive heads of ONE partition, so the 128-byte lines of a token row that two heads share ([q|k|v] of dh = 32
// channels is 192 bytes per head) are fetched from HBM by one CU, once, instead of by workgroups on different XCDs.
// The waves of a workgroup are independent (private LDS slices): they only ever synchronise with themselves.
//
// The qkv activations stay in IMAGE token order [F*H*W][3C] (per-head channel layout [q|k|v], dh each —
// reference maxvit.py:347); window / grid partitioning is pure index arithmetic on the token rows that a
// partition gathers, so the reference's four permute().contiguous() copies per block never exist.
//
// MFMA formulation (L <= 96 tokens per partition, dh <= 32):
//   S^T = K Q^T        A = K rows (keys j), B = Q rows (queries i)  -> lane owns ONE query column i,
//                      its 16*NB accumulator registers run over keys  => softmax max/sum are in-lane
//                      reductions plus one exchange with lane^32.
//   O^T = V^T P^T      B = P^T straight from the accumulator registers: the MFMA pairs A's and B's
//                      k-slots (half,e) one-to-one, so we are free to *choose* which key each slot
//                      means; we pick the keys the lane already holds ( j = 32bj+16q+8(e>>2)+4*half+(e&3) )
//                      and read V^T from LDS in that same order.  No register shuffles, no P round trip.
// Backward recomputes S/P from the saved qkv (nothing but qkv and the output is ever stored):
//   dP^T = V dO^T;  delta_i = sum_j P dP;  dS^T = P^T (dP^T - delta) * scale
//   dQ^T = K^T dS^T (same slot trick);  dV = P^T-rows x dO,  dK = dS^T-rows x Q  (contraction over queries:
//   P^T / dS^T take one trip through LDS to become row-major A operands).
#pragma once
#include "common.hpp"

namespace rvt {

// LDS hand-off between the lanes of ONE wave (private slice): DS operations of a wave execute in issue order, so draining
// lgkmcnt (plus a compiler barrier) is all the synchronisation there is to do — no s_barrier across the workgroup.
__device__ __forceinline__ void wave_lds_sync() {
#ifdef RVT_EMU
    emu::wave_barrier();          // emulator: lanes are fibers; the LDS hand-off is between the lanes of this wave only
#else
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

struct AttnGeom {
    int F, H, W, C, dh, heads, ph, pw, L, window;   // L = ph*pw
    int nPw;        // partitions along x
    int P;          // partitions per frame
    float scale;
    FastDiv dGroups, dP, dnPw, dpw;   // dGroups: head groups (of HG heads) per partition
};

// image-order token row of slot l of partition p of frame f
__device__ __forceinline__ int attn_token(const AttnGeom& g, int f, int p, int l) {
    uint32_t py, px, ly, lx;
    g.dnPw.divmod((uint32_t)p, py, px);
    g.dpw.divmod((uint32_t)l, ly, lx);
    int y, x;
    if (g.window) { y = py * g.ph + ly; x = px * g.pw + lx; }                  // maxvit.py:273-279
    else { y = ly * (g.H / g.ph) + py; x = lx * (g.W / g.pw) + px; }           // maxvit.py:290-296 (dilated grid)
    return (f * g.H + y) * g.W + x;
}

template <class T> __device__ __forceinline__ frag_t<T> load_chunk(const T* row, int chunk, int dh, bool valid) {
    // branch-free: padded rows point at token 0 (always mapped); the select zeroes what is not real
    const bool ok = valid & (chunk * 8 < dh);
    const frag_t<T> v = frag_load<T>(row + (ok ? chunk * 8 : 0));
    const frag_t<T> z = frag_zero<T>();
    return ok ? v : z;
}

// store 4 consecutive channels of one token row as ONE 8-byte (bf16) / 16-byte (f32) write
template <class T> __device__ __forceinline__ void store4(T* dst, float a, float b, float c, float d) {
    typedef __attribute__((ext_vector_type(4))) T vec4;
    vec4 v;
    v[0] = (T)a; v[1] = (T)b; v[2] = (T)c; v[3] = (T)d;
    *reinterpret_cast<vec4*>(dst) = v;
}

// write an 8-element chunk transposed: dst[(chunk*8+e)*pitch + col] = v[e]
template <class T> __device__ __forceinline__ void store_transposed(T* dst, int pitch, int chunk, int col, const frag_t<T>& v) {
#pragma unroll
    for (int e = 0; e < 8; e++) dst[(chunk * 8 + e) * pitch + col] = v[e];
}

// A operand of the "slot trick": rows d = lane&31 of a transposed [32][pitch] LDS matrix, keys in slot order
template <class T> __device__ __forceinline__ frag_t<T> load_slot_frag(const T* mt, int pitch, int li, int half, int bj, int q) {
    const T* p = mt + li * pitch + 32 * bj + 16 * q + 4 * half;
    frag_t<T> f;
#pragma unroll
    for (int e = 0; e < 4; e++) { f[e] = p[e]; f[4 + e] = p[8 + e]; }
    return f;
}

template <class T> __device__ __forceinline__ frag_t<T> acc_slot_frag(const f32x16& a, int q) {
    frag_t<T> f;
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = (T)a[8 * q + e];
    return f;
}

// S^T column softmax for one query block.  In: raw scores s[bj][r] (keys down the registers, this lane's query).
// Out: p[bj][r] = exp(scale*(s - max)) (UN-normalised, plain VGPR array so the accumulator registers are only read)
// and the column's 1/sum.  Padded keys (j >= L) only exist in the last 32-key block: they get an additive -1e30
// (kmask[r], built once per kernel) instead of per-element selects; exp2 with scale*log2(e) folded into one multiply.
template <int NB>
__device__ __forceinline__ float softmax_cols(const f32x16 (&s)[NB], float (&p)[NB][16], const float (&kmask)[16], float scale_log2e) {
    float mx = -3.0e38f;
#pragma unroll
    for (int bj = 0; bj < NB; bj++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float v = bj == NB - 1 ? s[bj][r] + kmask[r] : s[bj][r];
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

template <class T> __device__ __forceinline__ frag_t<T> arr_slot_frag(const float (&a)[16], int q) {
    frag_t<T> f;
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = (T)a[8 * q + e];
    return f;
}

// additive key mask of the LAST key block for this lane's accumulator rows
template <int NB> __device__ __forceinline__ void make_kmask(float (&kmask)[16], int lane, int L) {
#pragma unroll
    for (int r = 0; r < 16; r++) kmask[r] = (32 * (NB - 1) + acc_row(r, lane) < L) ? 0.f : -1.0e30f;
}

template <class T, int NB, int HG>
__global__ void __launch_bounds__(64 * HG)
attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ out, AttnGeom g) {
    constexpr int LP = 32 * NB, PITCH = LP + 8;
    __shared__ __attribute__((aligned(16))) T Vt_all[HG][32 * PITCH];
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
    T* const Vt = Vt_all[wv];
    uint32_t fp, grp, f, p;
    g.dGroups.divmod(blockIdx.x, fp, grp);
    g.dP.divmod(fp, f, p);
    const int head = (int)grp * HG + wv;
    const int C3 = 3 * g.C, dh = g.dh;
    const int qoff = head * 3 * dh, koff = qoff + dh, voff = qoff + 2 * dh;

    int tok[NB]; bool valid[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        int l = 32 * b + li;
        valid[b] = l < g.L;
        tok[b] = attn_token(g, (int)f, (int)p, valid[b] ? l : 0);
    }
    float kmask[16];
    make_kmask<NB>(kmask, lane, g.L);
    const float scale_log2e = g.scale * 1.4426950408889634f;
    // all global reads in one batch (see the backward kernel); V^T -> LDS (zeros for padded keys so that 0 * garbage can
    // never appear)
    frag_t<T> qf[NB][2], kf[NB][2];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const int chunk = half + 2 * cc;
            const T* row = qkv + (size_t)tok[b] * C3;
            qf[b][cc] = load_chunk<T>(row + qoff, chunk, dh, valid[b]);
            kf[b][cc] = load_chunk<T>(row + koff, chunk, dh, valid[b]);
            store_transposed<T>(Vt, PITCH, chunk, 32 * b + li, load_chunk<T>(row + voff, chunk, dh, valid[b]));
        }
    wave_lds_sync();

#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
        f32x16 s[NB];
#pragma unroll
        for (int bj = 0; bj < NB; bj++) acc_zero(s[bj]);
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            if (ks * 16 < dh) {
#pragma unroll
                for (int bj = 0; bj < NB; bj++) mma32(s[bj], kf[bj][ks], qf[bi][ks]);
            }
        }
        float pr[NB][16];
        const float inv = softmax_cols<NB>(s, pr, kmask, scale_log2e);
        f32x16 o; acc_zero(o);
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int q = 0; q < 2; q++)
                mma32(o, load_slot_frag<T>(Vt, PITCH, li, half, bj, q), arr_slot_frag<T>(pr[bj], q));
        if (valid[bi]) {       // normalise the 16 outputs instead of the 32*NB probabilities
            T* orow = out + (size_t)tok[bi] * g.C + head * dh;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                int d0 = 8 * gq + 4 * half;
                if (d0 < dh) store4<T>(orow + d0, o[4 * gq] * inv, o[4 * gq + 1] * inv, o[4 * gq + 2] * inv, o[4 * gq + 3] * inv);
            }
        }
    }
}

template <class T, int NB> struct AttnBwdLds {       // per wave
    static constexpr int LP = 32 * NB, PITCH = LP + 8, PSP = 40;
    static constexpr int BYTES = (3 * 32 * PITCH + LP * PSP) * (int)sizeof(T);
};

template <class T, int NB, int HG>
__global__ void __launch_bounds__(64 * HG, NB == 3 ? 1 : 2)      // <= 256 VGPR+AGPR: two waves per SIMD (three key blocks need more)
attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dout, T* __restrict__ dqkv, AttnGeom g) {
    constexpr int LP = 32 * NB, PITCH = LP + 8, PSP = 40;
    __shared__ __attribute__((aligned(16))) T Qt_all[HG][32 * PITCH];
    __shared__ __attribute__((aligned(16))) T Kt_all[HG][32 * PITCH];
    __shared__ __attribute__((aligned(16))) T dOt_all[HG][32 * PITCH];
    __shared__ __attribute__((aligned(16))) T PS_all[HG][LP * PSP];
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5, wv = threadIdx.x >> 6;
    T* const Qt = Qt_all[wv]; T* const Kt = Kt_all[wv]; T* const dOt = dOt_all[wv]; T* const PS = PS_all[wv];
    uint32_t fp, grp, f, p;
    g.dGroups.divmod(blockIdx.x, fp, grp);
    g.dP.divmod(fp, f, p);
    const int head = (int)grp * HG + wv;
    const int C3 = 3 * g.C, dh = g.dh;
    const int qoff = head * 3 * dh, koff = qoff + dh, voff = qoff + 2 * dh, ooff = head * dh;

    int tok[NB]; bool valid[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
        int l = 32 * b + li;
        valid[b] = l < g.L;
        tok[b] = attn_token(g, (int)f, (int)p, valid[b] ? l : 0);
    }
    float kmask[16];
    make_kmask<NB>(kmask, lane, g.L);
    const float scale_log2e = g.scale * 1.4426950408889634f;
    // Every global read of the kernel is issued here, in one batch (the wave then never waits for memory again until its
    // stores): lane (li, half) holds chunks {half, half+2} of rows 32b+li of Q, K, V and dO — exactly the MFMA operand
    // fragments of K-step ks = cc, and the source of the transposed LDS copies.
    frag_t<T> qf[NB][2], kf[NB][2], vf[NB][2], df[NB][2];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const int chunk = half + 2 * cc;
            const T* row = qkv + (size_t)tok[b] * C3;
            qf[b][cc] = load_chunk<T>(row + qoff, chunk, dh, valid[b]);
            kf[b][cc] = load_chunk<T>(row + koff, chunk, dh, valid[b]);
            vf[b][cc] = load_chunk<T>(row + voff, chunk, dh, valid[b]);
            df[b][cc] = load_chunk<T>(dout + (size_t)tok[b] * g.C + ooff, chunk, dh, valid[b]);
        }
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const int chunk = half + 2 * cc;
            store_transposed<T>(Qt, PITCH, chunk, 32 * b + li, qf[b][cc]);
            store_transposed<T>(Kt, PITCH, chunk, 32 * b + li, kf[b][cc]);
            store_transposed<T>(dOt, PITCH, chunk, 32 * b + li, df[b][cc]);
        }
    wave_lds_sync();

    f32x16 dk[NB], dv[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) { acc_zero(dk[b]); acc_zero(dv[b]); }

#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
        f32x16 s[NB], dp[NB];
#pragma unroll
        for (int bj = 0; bj < NB; bj++) { acc_zero(s[bj]); acc_zero(dp[bj]); }
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            if (ks * 16 < dh) {
#pragma unroll
                for (int bj = 0; bj < NB; bj++) {
                    mma32(s[bj], kf[bj][ks], qf[bi][ks]);
                    mma32(dp[bj], vf[bj][ks], df[bi][ks]);
                }
            }
        }
        float pr[NB][16];
        const float inv = softmax_cols<NB>(s, pr, kmask, scale_log2e);
        float delta = 0.f;
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int r = 0; r < 16; r++) { pr[bj][r] *= inv; delta += pr[bj][r] * dp[bj][r]; }
        delta += __shfl_xor(delta, 32);
        // P^T -> LDS [key][query-in-block] for dV
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int r = 0; r < 16; r++) PS[(32 * bj + acc_row(r, lane)) * PSP + li] = (T)pr[bj][r];
        // dS^T (keeps the softmax scale so that dQ and dK need no further factor)
        float ds[NB][16];
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int r = 0; r < 16; r++) ds[bj][r] = pr[bj][r] * (dp[bj][r] - delta) * g.scale;
        wave_lds_sync();
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                frag_t<T> a = frag_load<T>(PS + (32 * bj + li) * PSP + ks * 16 + half * 8);
                frag_t<T> b = frag_load<T>(dOt + li * PITCH + 32 * bi + ks * 16 + half * 8);
                mma32(dv[bj], a, b);
            }
        wave_lds_sync();
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int r = 0; r < 16; r++) PS[(32 * bj + acc_row(r, lane)) * PSP + li] = (T)ds[bj][r];
        wave_lds_sync();
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int ks = 0; ks < 2; ks++) {
                frag_t<T> a = frag_load<T>(PS + (32 * bj + li) * PSP + ks * 16 + half * 8);
                frag_t<T> b = frag_load<T>(Qt + li * PITCH + 32 * bi + ks * 16 + half * 8);
                mma32(dk[bj], a, b);
            }
        // dQ^T[d][i] = sum_j K^T[d][j] dS^T[j][i]   (slot trick, dS^T from registers)
        f32x16 dq; acc_zero(dq);
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int q = 0; q < 2; q++)
                mma32(dq, load_slot_frag<T>(Kt, PITCH, li, half, bj, q), arr_slot_frag<T>(ds[bj], q));
        if (valid[bi]) {
            T* qrow = dqkv + (size_t)tok[bi] * C3 + qoff;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                int d0 = 8 * gq + 4 * half;
                if (d0 < dh) store4<T>(qrow + d0, dq[4 * gq], dq[4 * gq + 1], dq[4 * gq + 2], dq[4 * gq + 3]);
            }
        }
        wave_lds_sync();   // PS is rewritten by the next query block
    }
    // dK, dV: accumulator rows = keys, col = d = lane&31.  Stage each through LDS (the P^T buffer, [key][40]) so that the
    // global writes are 16-byte row segments instead of 2-byte scalars.
    constexpr int CPR = 32 / 8;                       // 8-channel chunks per key row (dh <= 32)
#pragma unroll
    for (int which = 0; which < 2; which++) {
        wave_lds_sync();
#pragma unroll
        for (int bj = 0; bj < NB; bj++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                PS[(32 * bj + acc_row(r, lane)) * PSP + li] = (T)(which == 0 ? dk[bj][r] : dv[bj][r]);
        wave_lds_sync();
        const int off = which == 0 ? koff : voff;
        for (int u = lane; u < LP * CPR; u += 64) {
            const int j = u / CPR, c = u % CPR;
            if (j < g.L && c * 8 < dh) {
                const int t = attn_token(g, (int)f, (int)p, j);
                frag_store<T>(dqkv + (size_t)t * C3 + off + c * 8, frag_load<T>(PS + j * PSP + c * 8));
            }
        }
    }
}

}  // namespace rvt
