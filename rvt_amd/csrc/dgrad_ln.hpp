// Input gradient of a linear layer whose input is a LayerNorm output, with that LayerNorm's backward in the epilogue:
//     dx = add + LN'( dy W ; x ),      dln_w += sum_tok (dy W) * xhat,   dln_b += sum_tok (dy W)
// (reference maxvit.py:229,347 - norm1 -> qkv - and :241,100-118 - norm2 -> fc1 - under autograd.)  dy [M][K] with K = 3C / 4C,
// W [K][C] = the forward weight as it is, x [M][C] = the LayerNorm input, add [M][C] = the cotangent arriving over the residual.
//
// Stage 2 of RVT-Base (C = 128, 1.9 M tokens) ran this as two launches - a 128-column GEMM that writes du (0.5 GB) and the
// LayerNorm backward that reads it back with x and add: 0.90 + 0.39 ms (K = 512), 0.45 + 0.39 ms (K = 384).  The GEMM's output
// rows are complete rows of the LayerNorm, so the chained form of csrc/mlp_chain.hpp applies with nothing recomputed: one wave
// owns 32 token rows, reads its dy rows from HBM directly in MFMA-operand form (lane = row, 16 bytes per k-step), multiplies them
// with W^T fragments that come out of ONE LDS image of W through the transposing read, and keeps du^T in accumulators
// (lane = token, registers = channels).  From there the LayerNorm backward is in-lane arithmetic plus one lane^32 exchange per row
// statistic; the parameter gradients are column sums over tokens = an MFMA against an identity operand (exact), as in
// mlpc_bwd_dgrad_kernel.  HBM traffic: dy + x + add in, dx out - du never exists.  No VALU-heavy element function anywhere: the
// kernel is bound by its dy stream.
#pragma once
#include "mlp_chain.hpp"

namespace rvt {

template <class T, int C> struct DglSmem {
    static constexpr int KT = C / TileGeom<T>::BK;
    static constexpr int MAXK = 4 * C;
    static constexpr int W_BYTES = KT * MAXK * 128;                 // [K rows][C] operand image
    static constexpr int NCONST = C;                                // ln_w
    static constexpr int SCR_PITCH = 80, SCR_WAVE = 32 * SCR_PITCH;  // per wave: [32 rows][64 B + pad], the store bounce of pass 2
    static constexpr int OFF_SCR = W_BYTES + NCONST * 4;
    static constexpr int BYTES_FOR(int wpb) { return OFF_SCR + wpb * SCR_WAVE; }
};

// INSIDE (round 6): the added cotangent enters the norm instead of by-passing it,  dx = LN'( dy W + add ; x )  - the first block of a
// stage (no norm1; `skip_first_norm`, maxvit_rnn.py:153): its qkv input gradient plus the block's residual cotangent is the cotangent
// of the down-sampling norm's output (maxvit.py:177), x = that norm's input y0.  Replaces the GEMM and the LayerNorm backward launch.
template <class T, int C, int WPB, int AHEAD, bool INSIDE = false>
__global__ void __launch_bounds__(64 * WPB, 2)
dgrad_ln_kernel(const T* __restrict__ dy, const T* __restrict__ W, const T* __restrict__ x, const T* __restrict__ add,
                T* __restrict__ dx, const float* __restrict__ ln_w, float* __restrict__ dln_w, float* __restrict__ dln_b,
                int M, int K, float eps) {
    typedef DglSmem<T, C> S;
    constexpr int KS = C / 16, NCB = C / 32;
    __shared__ __attribute__((aligned(16))) char smem[S::BYTES_FOR(WPB)];
    char* const W_l = smem;
    float* const kst = reinterpret_cast<float*>(smem + S::W_BYTES);
    char* const scr = smem + S::OFF_SCR + (threadIdx.x >> 6) * S::SCR_WAVE;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, half = lane >> 5, wave = tid >> 6;
    chain_stage_weights<T, C, false>(W_l, W, K, tid, 64 * WPB);
    for (int i = tid; i < C; i += 64 * WPB) kst[i] = ln_w[i];
    __syncthreads();

    static_assert(sizeof(T) == 2, "bf16 kernel (the transposing LDS read)");
    // W^T fragments "row = channel cb*32 + lane&31, slots = rows j0 + 8 half + e" out of the [K][C] image through the transposing
    // read: two per-lane byte offsets serve every (row block, channel block) - channel block cb adds a sub-tile (cb >> 1) and flips
    // chunk bit 2 (cb & 1), a row block of 16 adds 2 KB and flips the (row >> 4) & 7 swizzle term (common.hpp lds_chunk_off)
    int pre_lo, pre_hi;
    {
        const int tl = 8 * half + ((lane & 15) >> 2), byte0 = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
        pre_lo = tl * 128 + (byte0 & 15) + (((byte0 >> 4) ^ ((tl >> 1) & 7)) << 4);
        pre_hi = (tl + 4) * 128 + (byte0 & 15) + (((byte0 >> 4) ^ (((tl + 4) >> 1) & 7)) << 4);
    }
    const int sub1 = K * 128;                                        // byte offset of the second 64-column sub-tile
    float aw[NCB], ab[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { aw[cb] = 0.f; ab[cb] = 0.f; }

    const int n_tiles = (M + 31) / 32, NJC = K / 32;
    // the first AHEAD chunks of a tile's dy rows are requested while the PREVIOUS tile runs its LayerNorm epilogue (1.5 k VALU
    // instructions with nothing else in flight)
    frag_t<T> df[AHEAD][2];
    constexpr int EARLY = 1;                                         // (more of the ring alive across the epilogue spills at C = 128)
    auto prime = [&](int t) __attribute__((always_inline)) {
        const int r = t * 32 + li;
        const T* const p = dy + (size_t)(r < M ? r : M - 1) * K + 8 * half;
#pragma unroll
        for (int a = 0; a < EARLY; a++)
#pragma unroll
            for (int q = 0; q < 2; q++) df[a][q] = frag_load<T>(p + 32 * a + 16 * q);
    };
    if ((int)(blockIdx.x * WPB + wave) < n_tiles) prime(blockIdx.x * WPB + wave);
    for (int tile = blockIdx.x * WPB + wave; tile < n_tiles; tile += gridDim.x * WPB) {
        const int row = tile * 32 + li;
        const bool valid = row < M;
        const bool interior = tile * 32 + 32 <= M;                   // (wave-uniform)
        const int rowc = valid ? row : M - 1;                        // tail rows read a real row; their du is zeroed below
        const T* const dyr = dy + (size_t)rowc * K + 8 * half;
        int ko = 0;                                                  // (LayerNorm weights re-read per tile: hoisted, they are 64 registers)
        opaque_vgpr(ko);
        const float* const kw = kst + ko;
        frag_t<T> xf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) xf[ks] = frag_load<T>(x + (size_t)rowc * C + (2 * ks + half) * 8);   // (in flight during the product)
#pragma unroll
        for (int a = EARLY; a < AHEAD; a++)
#pragma unroll
            for (int q = 0; q < 2; q++) df[a][q] = frag_load<T>(dyr + 32 * a + 16 * q);
        f32x16 dacc[NCB];
        // du^T[c][tok] = sum_j W[j][c] dy[tok][j]: hidden chunks of 32, AHEAD chunks of dy fragments in flight (K / 32 % AHEAD = 0).
        // The first group is peeled: its first products take C = 0 from the instruction (no zero fill of 16 NCB registers), and the
        // accumulators are defined before the loop, so that its exit needs no copies between register sets.
        auto group = [&](int jc0, auto first) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first)::value;
            // the image is entered at an OPAQUE byte offset per group of AHEAD chunks, so that the fragment addresses are formed
            // here (one xor each) instead of being hoisted out of the loop as a table (it spills at C = 128)
            int jb = jc0 * 32 * 128;
            opaque_vgpr(jb);
            const int t_lo = pre_lo + jb, t_hi = pre_hi + jb;
            const int jx = AHEAD == 4 ? 0 : ((2 * jc0) & 7) << 4;    // (AHEAD = 2: jc0 contributes bit 2 of the swizzle term)
#pragma unroll
            for (int a = 0; a < AHEAD; a++) {
                const int jc = jc0 + a;
                frag_t<T> cur[2] = {df[a][0], df[a][1]};
                if (jc + AHEAD < NJC) {
#pragma unroll
                    for (int q = 0; q < 2; q++) df[a][q] = frag_load<T>(dyr + 32 * (jc + AHEAD) + 16 * q);
                }
#pragma unroll
                for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int xk = ((((cb & 1) << 2) ^ ((2 * a + q) & 7)) << 4) ^ jx;
                        const int off = (cb >> 1) * sub1 + (32 * a + 16 * q) * 128;
                        const frag_t<T> wf = frag_from_tr<T>(reinterpret_cast<const bf16*>(W_l + ((t_lo ^ xk) + off)),
                                                             reinterpret_cast<const bf16*>(W_l + ((t_hi ^ xk) + off)));
                        if (FIRST && a == 0 && q == 0) mma32_zero(dacc[cb], wf, cur[q]);
                        else mma32(dacc[cb], wf, cur[q]);
                    }
            }
        };
        group(0, std::true_type());
        for (int jc0 = AHEAD; jc0 < NJC; jc0 += AHEAD) group(jc0, std::false_type());
        if (tile + (int)(gridDim.x * WPB) < n_tiles) prime(tile + gridDim.x * WPB);
        // INSIDE: the added rows are needed in pass 1 already; the pieces of channel block cb + 1 are requested while block cb is folded
        const T* const addr_in = INSIDE ? add + (size_t)rowc * C + half * 8 : nullptr;
        frag_t<T> ai[2] = {frag_zero<T>(), frag_zero<T>()}, ain[2] = {frag_zero<T>(), frag_zero<T>()};
        if (INSIDE) { ai[0] = frag_load<T>(addr_in); ai[1] = frag_load<T>(addr_in + 16); }
        sched_fence();
        // row statistics of x (lane = row: in-lane sums + one exchange)
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int e = 0; e < 8; e++) s += (float)xf[ks][e];
        s += __shfl_xor(s, 32);
        const float mean = s / (float)C;
        float qv = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int e = 0; e < 8; e++) { const float d = (float)xf[ks][e] - mean; qv += d * d; }
        qv += __shfl_xor(qv, 32);
        const float rstd = 1.0f / sqrtf(qv / (float)C + eps);
        // pass 1: accumulators -> row pieces (rounded to T as the two-launch chain stores them); S1 = sum g w, S2 = sum g w x
        // (xhat = (x - mean) rstd is expanded: the row sums only need the raw x)
        frag_t<T> rf[KS];
        float s1 = 0.f, s2 = 0.f;
        const float mr = mean * rstd;
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            float r8[2][8];
            acc_to_rows(dacc[cb], r8);
            if (INSIDE) {
                if (cb + 1 < NCB) { ain[0] = frag_load<T>(addr_in + 16 * (2 * cb + 2)); ain[1] = frag_load<T>(addr_in + 16 * (2 * cb + 3)); }
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int e = 0; e < 8; e++) r8[m][e] += (float)ai[m][e];
                ai[0] = ain[0]; ai[1] = ain[1];
            }
            if (!interior) {
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int e = 0; e < 8; e++) r8[m][e] = valid ? r8[m][e] : 0.f;
            }
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int ks = 2 * cb + m;
                float w[8];
                load_cols<8>(kw, 16 * ks + 8 * half, w);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float gw = r8[m][e] * w[e];
                    s1 += gw;
                    s2 = fmaf(gw, (float)xf[ks][e], s2);
                }
                rf[ks] = frag_from_float<T>(r8[m]);
            }
            sched_fence();
        }
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        // m1 = mean_c(g w), m2 = mean_c(g w xhat) = rstd (S2 - mean S1) / C;  dx = add + rstd (g w - m1 - xhat m2) = add + (rstd w) g + A + B x
        const float m1 = s1 / (float)C, m2 = rstd * (s2 - mean * s1) / (float)C;
        const float Bc = -rstd * rstd * m2, Ac = -rstd * m1 - mean * Bc;
        // pass 2.  The rows leave through a wave-private LDS bounce, two 16-channel pieces (64 bytes of every row) at a time: in
        // operand form a store instruction would write 32 bytes of each of 32 rows (0.95 ms; 0.85 with the bounce, 0.82 with the
        // residual pieces requested ahead of the stores); from the bounce a store instruction writes 64 contiguous bytes of 16 rows.  (DS operations of one wave execute in order: no barrier.)
        const T* const addr = (!INSIDE && add != nullptr) ? add + (size_t)rowc * C + half * 8 : nullptr;
        // (vmcnt retires loads and stores in issue order: a residual piece requested AFTER the stores of the previous pair would
        // wait for their acknowledgement - the next pair's pieces are requested before this pair's rows are stored)
        frag_t<T> af[2] = {frag_zero<T>(), frag_zero<T>()}, afn[2] = {frag_zero<T>(), frag_zero<T>()};
        if (addr != nullptr) { af[0] = frag_load<T>(addr); af[1] = frag_load<T>(addr + 16); }
#pragma unroll
        for (int kp = 0; kp < KS / 2; kp++) {
            if (addr != nullptr && kp + 1 < KS / 2) { afn[0] = frag_load<T>(addr + 16 * (2 * kp + 2)); afn[1] = frag_load<T>(addr + 16 * (2 * kp + 3)); }
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int ks = 2 * kp + m;
                float w[8], o[8], av[8], dv[8];
                load_cols<8>(kw, 16 * ks + 8 * half, w);
                frag_to_float<T>(af[m], av);
                frag_to_float<T>(rf[ks], dv);
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = av[e] + fmaf((float)xf[ks][e], Bc, fmaf(dv[e], rstd * w[e], Ac));
                *reinterpret_cast<frag_t<T>*>(scr + li * S::SCR_PITCH + (2 * m + half) * 16) = frag_from_float<T>(o);
            }
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int rr = 16 * j + (lane >> 2), orow = tile * 32 + rr;
                const frag_t<T> v = *reinterpret_cast<const frag_t<T>*>(scr + rr * S::SCR_PITCH + (lane & 3) * 16);
                if (orow < M) frag_store<T>(dx + (size_t)orow * C + kp * 32 + (lane & 3) * 8, v);
            }
            wave_lds_sync();
            af[0] = afn[0]; af[1] = afn[1];
            sched_fence();
        }
        // pass 3: LayerNorm parameter gradients = column sums over the 32 tokens of du xhat and du.  The row pieces hold tokens in
        // the LANES; an MFMA against an identity operand turns a piece into "col = channel, registers = tokens" (exact), where the
        // column sum is an in-lane sum.  (After the stores: with these accumulators alive beside pass 1 the kernel spilled.)
        frag_t<T> idf[2];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int e = 0; e < 8; e++) idf[m][e] = (T)((16 * m + 8 * half + e == li) ? 1.0f : 0.0f);
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            f32x16 tw, tb;
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int ks = 2 * cb + m;
                float dv[8], pw[8];
                frag_to_float<T>(rf[ks], dv);
#pragma unroll
                for (int e = 0; e < 8; e++) pw[e] = dv[e] * fmaf((float)xf[ks][e], rstd, -mr);
                if (m == 0) { mma32_zero(tw, frag_from_float<T>(pw), idf[m]); mma32_zero(tb, rf[ks], idf[m]); }
                else { mma32(tw, frag_from_float<T>(pw), idf[m]); mma32(tb, rf[ks], idf[m]); }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) { aw[cb] += tw[r]; ab[cb] += tb[r]; }
            sched_fence();
        }
    }
    // fold the two halves and the waves: one atomic per channel per workgroup
    __syncthreads();                                                 // W is dead: the LDS becomes reduction scratch
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

}  // namespace rvt
