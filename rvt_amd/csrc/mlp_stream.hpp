// MLP half of a block (reference maxvit.py:269 with the MLP of :100-118 and its autograd) for C = 128 — the chained kernels of
// mlp_chain.hpp with the weights STREAMED through LDS instead of resident in it:
//
//     xout = xmid + gamma2 * ( GELU( LN2(xmid) W1^T + b1 ) W2^T + b2 )
//
// At C = 64 the two weight matrices (64 KiB bf16) stay in LDS for the whole persistent launch and the waves of a workgroup
// never synchronise.  At C = 128 they are 256 KiB.  Here a workgroup of 8 waves (256 token rows, 32 per wave, data flow of a
// wave exactly as in mlp_chain.hpp: rows in MFMA-operand form, accumulator -> operand chaining, nothing but weights in LDS)
// walks the hidden axis in CHUNKS of one 128-byte LDS sub-tile (64 hidden columns in bf16): chunk g+1 of (W1 rows, W2 columns)
// arrives by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction, swizzle on the SOURCE address) into the other half
// of a two-stage ring while the waves multiply chunk g; one `s_waitcnt vmcnt(0)` + one workgroup barrier per chunk.  256 KiB of
// weights per 256 tokens come from L2 (2 GB per launch at 1.9 M tokens: a tenth of what one 32-token wave streaming on its
// own would pull) and every weight fragment read from LDS feeds one MFMA.
//
//   mlps_fwd_kernel         reads xmid, writes xout                                         (nothing saved: the backward recomputes)
//   mlps_bwd_dgrad_kernel   reads dxout, xmid; writes dxmid = dxout + LN2'(dh W1), dln_w / dln_b +=
//   mlps_bwd_wgrad_kernel   reads dxout, xmid; weight-stationary: dW1, db1, S2 = dxout^T g, cs2 (partials per workgroup + fold)
//
// Per token the MLP half then moves 2 + 3 + 2 rows of C through HBM instead of the 11 + 26 rows of the op-by-op route (GELU and
// GELU' stored 4C wide by the forward, read back by two input-gradient and two weight-gradient GEMMs).
#pragma once
#include "common.hpp"
#include "mlp_chain.hpp"
#include "ppgemm.hpp"

namespace rvt {

template <class T, int C, int HC_ = TileGeom<T>::BK> struct MsGeom {
    static constexpr int BK = TileGeom<T>::BK;            // elements of a 128-byte sub-tile row
    static constexpr int HC = HC_;                        // hidden columns per streamed chunk (forward: BK, the W2 chunk = ONE sub-tile)
    static constexpr int KT = C / BK;
    static constexpr int HID = 4 * C, NCH = HID / HC, JPC = HC / 32;
    static constexpr int W1C = KT * HC * 128;             // chunk of W1 (or (gamma W2)^T): [HC rows][C]
    static constexpr int W2C = C * 128;                   // chunk of W2: [C rows][HC]
    static constexpr int STAGE = W1C + W2C;
    static constexpr int K_LNW = 0, K_LNB = C, K_B2 = 2 * C, K_GAM = 3 * C, K_B1 = 4 * C, NCONST = 8 * C;
};

// LDS image of a streamed chunk: sub-tiles of [rows][128 B] with the 16-byte chunk index XOR (row >> 1) & 7 — the first term of
// lds_chunk_off only.  (Its second term spreads the WRITES of the transposing loader; an LDS-DMA image has no such writer, and
// with it the swizzle of row 32 cb + li would depend on cb: one address register per (cb, chunk) instead of one per chunk.)
// A fragment address is (row * 128 + swz * 16) ^ (chunk * 16): a per-lane base XOR a compile-time constant.
__device__ __forceinline__ int ms_swz(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int ms_rowbase(int row) { return row * 128 + (ms_swz(row) << 4); }
// ... and the lane's half (k-slots 8 half .. of a k-step = fragment 2 ks + half) folded into the per-lane base as well: every
// fragment address of a lane is then ONE of a few registers (base ^ compile-time chunk) plus an immediate offset
template <class T> __device__ __forceinline__ int ms_rowbase_h(int row, int half) { return ms_rowbase(row) ^ ((half * TileGeom<T>::CPF) << 4); }
template <int I, int N, class F> __device__ __forceinline__ void ms_static_for_impl(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>()); ms_static_for_impl<I + 1, N>(f); }
}
template <int N, class F> __device__ __forceinline__ void ms_static_for(F&& f) { ms_static_for_impl<0, N>(f); }
// fragment fcg (8 elements; the EVEN index 2 ks when `rb` comes from ms_rowbase_h) of the row whose base is `rb`, in a sub-tile
// array with `rows` rows per sub-tile
template <class T> __device__ __forceinline__ frag_t<T> ms_load_frag(const char* base, int rows, int rb, int fcg) {
    constexpr int FPR = TileGeom<T>::FPR, CPF = TileGeom<T>::CPF;
    const char* const t = base + (fcg / FPR) * rows * 128;
    frag_t<T> v;
    u32x4* dst = reinterpret_cast<u32x4*>(&v);
#pragma unroll
    for (int c = 0; c < CPF; c++) dst[c] = *reinterpret_cast<const u32x4*>(t + (rb ^ ((((fcg % FPR) * CPF + c)) << 4)));
    return v;
}
template <class T> __device__ __forceinline__ void ms_store_frag(char* base, int rows, int rb, int fcg, const frag_t<T>& v) {
    constexpr int FPR = TileGeom<T>::FPR, CPF = TileGeom<T>::CPF;
    char* const t = base + (fcg / FPR) * rows * 128;
    const u32x4* src = reinterpret_cast<const u32x4*>(&v);
#pragma unroll
    for (int c = 0; c < CPF; c++) *reinterpret_cast<u32x4*>(t + (rb ^ ((((fcg % FPR) * CPF + c)) << 4))) = src[c];
}
// rows of a 32-row block with bits 2 and 3 of the index exchanged: lane li supplies row ms_rowperm(li) as the A operand of
// the fc1 product, so that accumulator register 8 q + e (row (e & 3) + 8 (2 q + (e >> 2)) + 4 half) holds hidden unit
// 16 q + 8 half + e — the k-slot order of a PLAIN 16-byte fragment of W2.  (mlp_chain.hpp permutes the columns of W2 instead
// when it stages them; an LDS-DMA cannot permute inside its 16-byte pieces.)
__device__ __forceinline__ int ms_rowperm(int li) { return (li & ~12) | ((li & 4) << 1) | ((li & 8) >> 1); }

// per-lane source offsets (bytes) of this wave's LDS-DMA pieces: piece p = wave + i * WPB of a [rows][ld] matrix chunk whose
// LDS image is `subtiles` sub-tiles of [rows_per_subtile][128 B] (lane -> row 8 p' + lane / 8, chunk position lane % 8)
template <class T, int NPW, int WPB>
__device__ __forceinline__ void ms_piece_offsets(int (&v)[NPW], int wave, int lane, int rows_per_subtile, int ld) {
    constexpr int EPC = 16 / (int)sizeof(T);              // elements per 16-byte chunk
    const int rgs = rows_per_subtile / 8;                 // 1-KiB pieces per sub-tile
#pragma unroll
    for (int i = 0; i < NPW; i++) {
        const int p = wave + i * WPB, kt = p / rgs, r = (p % rgs) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ms_swz(r);
        v[i] = (r * ld + kt * TileGeom<T>::BK + c * EPC) * (int)sizeof(T);
    }
}

// ===================================================================================================== forward
// ABL: ablation bits for profiles/probes/mlps_probe.hip (0 in the library): 1 no weight stream after the first two chunks (no LDS-DMA,
// no vmcnt wait), 2 no workgroup barrier, 4 GELU replaced by a multiply (no table gather), 8 no row loads / stores inside the tile loop
template <class T, int C, int WPB, int MINW, int ABL = 0>
__global__ void __launch_bounds__(64 * WPB, MINW)
mlps_fwd_kernel(const T* __restrict__ xmid, T* __restrict__ xout, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                const T* __restrict__ W1, const float* __restrict__ b1, const T* __restrict__ W2, const float* __restrict__ b2,
                const float* __restrict__ gamma, int M, float eps) {
    typedef MsGeom<T, C> G;
    constexpr int KS = C / 16, NCB = C / 32, HID = 4 * C, NCH = G::NCH, JPC = G::JPC;
    constexpr int NPW = (G::W1C / 1024) / WPB;            // 1-KiB pieces of each matrix chunk per wave
    static_assert(NPW * WPB * 1024 == G::W1C && G::W1C == G::W2C, "chunk pieces must divide over the waves");
    __shared__ __attribute__((aligned(16))) char smem[2 * G::STAGE + G::NCONST * 4 + GeluTab<T>::BYTES + WPB * 1024];
    float* const kst = reinterpret_cast<float*>(smem + 2 * G::STAGE);
    float* const lut = kst + G::NCONST;
    const int tid = threadIdx.x, lane_ = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    GeluTab<T>::template fill<false>(lut, tid, 64 * WPB);
    for (int i = tid; i < C; i += 64 * WPB) {
        kst[G::K_LNW + i] = ln_w[i]; kst[G::K_LNB + i] = ln_b[i]; kst[G::K_B2 + i] = b2[i]; kst[G::K_GAM + i] = gamma[i];
    }
    for (int i = tid; i < HID; i += 64 * WPB) kst[G::K_B1 + i] = b1[(i & ~31) | ms_rowperm(i & 31)];   // accumulator-init order
    __syncthreads();

    // ---- weight stream ----
    const pp_rsrc r1 = pp_make_rsrc(W1, (unsigned)(HID * C * sizeof(T))), r2 = pp_make_rsrc(W2, (unsigned)(C * HID * sizeof(T)));
    // (everything lane-derived is recomputed from an OPAQUE copy of the lane id where it is used: hoisted out of the tile loop
    //  these per-lane address tables cost registers the chunk loop does not have - 292 bytes of scratch in the first version)
    auto issue = [&](int ch, int stage) __attribute__((always_inline)) {
        int lane = lane_;
        opaque_vgpr(lane);
        int v1[NPW], v2[NPW];
        ms_piece_offsets<T, NPW, WPB>(v1, wave, lane, G::HC, C);        // W1 chunk: rows = hidden units of the chunk
        ms_piece_offsets<T, NPW, WPB>(v2, wave, lane, C, HID);          // W2 chunk: rows = channels (one sub-tile)
#pragma unroll
        for (int i = 0; i < NPW; i++) {
            const int p = wave + i * WPB;
            pp_glds16(r1, smem, stage * G::STAGE + p * 1024, v1[i], ch * (G::HC * C * (int)sizeof(T)));
            pp_glds16(r2, smem, stage * G::STAGE + G::W1C + p * 1024, v2[i], ch * (G::HC * (int)sizeof(T)));
        }
    };

    const int n_tiles = (M + 31) / 32;
    const int n_wg = (n_tiles + WPB - 1) / WPB;           // tiles of the workgroup: every wave walks the same count (barriers)
    char* const pf_dummy = smem + 2 * G::STAGE + G::NCONST * 4 + GeluTab<T>::BYTES;      // 1 KiB per wave: landing zone of the L2 warm-up
    issue(0, 0);
    if (ABL & 1) { issue(1, 1); pp_wait_vm<0>(); __syncthreads(); }

    // Register plan (two waves per SIMD = 256 registers): a tile's raw rows are dead after the LayerNorm (the residual is read
    // again, from L2, during the LAST chunk); the next tile's rows are NOT prefetched into registers but pulled into L2 by
    // LDS-DMA pieces that land in a dummy buffer (no destination registers), so the loads at the top of a tile are L2 hits;
    // the finished rows of a tile wait in registers until the first barrier of the next tile has been passed (their stores
    // then have a whole chunk to retire before the next vmcnt(0)).  Chunk 0 and the last chunk are peeled so that `orow` / `res`
    // are not loop-carried through the chunk loop.
    frag_t<T> orow[KS];
    int prow = 0;
    bool pvalid = false;
    for (int twg = blockIdx.x; twg < n_wg; twg += gridDim.x) {
        int lane = lane_;
        opaque_vgpr(lane);
        const int li = lane & 31, half = lane >> 5;
        const int rb1 = ms_rowbase_h<T>(ms_rowperm(li), half), rb2 = ms_rowbase_h<T>(li, half);
        const int tile = twg * WPB + wave;
        const int row = tile * 32 + li;
        const bool valid = tile < n_tiles && row < M;
        const bool more = twg + (int)gridDim.x < n_wg;    // (workgroup-uniform)
        frag_t<T> uf[KS];
        {
            frag_t<T> xf[KS];
            mc_load_row<T, C>(xf, xmid, row, valid, half);
            float mean, rstd;
            mc_layernorm<T, C>(xf, uf, kst + G::K_LNW, kst + G::K_LNB, valid, half, eps, mean, rstd);
        }
        f32x16 oacc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc_zero(oacc[cb]);
        auto boundary = [&](int ch, int stage_next) __attribute__((always_inline)) {
            // chunk ch has landed (this wave's pieces: vmcnt; everybody's: barrier) and every wave is done with the other stage
            if (!(ABL & 1)) pp_wait_vm<0>();
            if (!(ABL & 2)) pp_barrier();
            if (ABL & 1) return;
            if (ch + 1 < NCH) issue(ch + 1, stage_next);
            else if (more) issue(0, stage_next);
        };
        auto compute = [&](int ch, auto stage_c) __attribute__((always_inline)) {
            constexpr int STG = decltype(stage_c)::value;
            const char* const W1s = smem + STG * G::STAGE;
            const char* const W2s = W1s + G::W1C;
#pragma unroll
            for (int jj = 0; jj < JPC; jj++) {
                f32x16 h;
                acc_load_rows(h, kst + G::K_B1 + ch * G::HC + 32 * jj, half);    // fc1 bias = initial value of the accumulator
                if (ABL & 16) {                           // (probe: two accumulator chains of four instead of one of eight)
                    f32x16 h2;
                    acc_zero(h2);
#pragma unroll
                    for (int ks = 0; ks < KS; ks += 2) {
                        mma32(h, (ABL & 8) ? uf[(ks + 1) % KS] : ms_load_frag<T>(W1s + 32 * jj * 128, G::HC, rb1, 2 * ks), uf[ks]);
                        mma32(h2, (ABL & 8) ? uf[(ks + 2) % KS] : ms_load_frag<T>(W1s + 32 * jj * 128, G::HC, rb1, 2 * ks + 2), uf[ks + 1]);
                    }
#pragma unroll
                    for (int r = 0; r < 16; r++) h[r] += h2[r];
                } else {
#pragma unroll
                for (int ks = 0; ks < KS; ks++) mma32(h, (ABL & 8) ? uf[(ks + 1) % KS] : ms_load_frag<T>(W1s + 32 * jj * 128, G::HC, rb1, 2 * ks), uf[ks]);
                }
                float g[16];
                if (ABL & 4) {
#pragma unroll
                    for (int r = 0; r < 16; r++) g[r] = 0.5f;
                } else {
                    GeluTab<T>::eval16(lut, h, g);
                }
#pragma unroll
                for (int r = 0; r < 16; r++) g[r] = mul_nopack(g[r], h[r], r);
                frag_t<T> gf[2];
                gf[0] = arr_slot_frag<T>(g, 0);
                gf[1] = arr_slot_frag<T>(g, 1);
#pragma unroll
                for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                    for (int q = 0; q < 2; q++)
                        mma32(oacc[cb], (ABL & 8) ? uf[cb + q] : ms_load_frag<T>(W2s + cb * 32 * 128, C, rb2, (G::HC / 8 / JPC) * jj + 2 * q), gf[q]);
            }
        };
        // ---- chunk 0: the previous tile's rows leave behind the barrier
        boundary(0, 1);
        if (pvalid) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) frag_store<T>(xout + (size_t)prow * C + (2 * ks + half) * 8, orow[ks]);
        }
        compute(0, std::integral_constant<int, 0>());
        // ---- chunks 1 .. NCH-2 in pairs (compile-time stages)
#pragma unroll 1
        for (int ch = 1; ch < NCH - 1; ch += 2) {
            boundary(ch, 0);
            if (ch == 1 && more) {                        // next tile's rows -> L2 (32 rows x C contiguous = C * sizeof(T) / 32 KiB pieces)
                const int tn = tile + (int)gridDim.x * WPB;
                if (tn < n_tiles) {
                    const int rows = M - tn * 32 < 32 ? M - tn * 32 : 32;
                    const pp_rsrc rx = pp_make_rsrc(xmid + (size_t)tn * 32 * C, (unsigned)(rows * C * sizeof(T)));
#pragma unroll
                    for (int i = 0; i < (32 * C * (int)sizeof(T)) / 1024; i++)
                        pp_glds16(rx, smem, (int)(pf_dummy - smem) + wave * 1024, lane * 16 + i * 1024, 0);
                }
            }
            compute(ch, std::integral_constant<int, 1>());
            boundary(ch + 1, 1);
            compute(ch + 1, std::integral_constant<int, 0>());
        }
        // ---- last chunk: the residual rows come back from L2 underneath it
        boundary(NCH - 1, 0);
        frag_t<T> res[KS];                                // (no select on `valid`: a select is a USE, and the wait for it would sit right here,
#pragma unroll                                            //  in front of the chunk; rows beyond M read row 0 and are never stored)
        for (int ks = 0; ks < KS; ks++) res[ks] = frag_load<T>(xmid + (size_t)(valid ? row : 0) * C + (2 * ks + half) * 8);
        sched_fence();
        compute(NCH - 1, std::integral_constant<int, 1>());
        sched_fence();
        // LayerScale + residual (maxvit.py:51-53,269)
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            float r8[2][8];
            acc_to_rows(oacc[cb], r8);
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int ks = 2 * cb + m;
                float rs[8], gam[8], b2v[8], o[8];
                frag_to_float<T>(res[ks], rs);
                load_cols<8>(kst + G::K_GAM, 16 * ks + 8 * half, gam);
                load_cols<8>(kst + G::K_B2, 16 * ks + 8 * half, b2v);
#pragma unroll
                for (int e = 0; e < 8; e++) o[e] = rs[e] + gam[e] * (r8[m][e] + b2v[e]);
                orow[ks] = frag_from_float<T>(o);
            }
        }
        prow = row;
        pvalid = valid;
    }
    if (pvalid) {
        const int half = lane_ >> 5;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) frag_store<T>(xout + (size_t)prow * C + (2 * ks + half) * 8, orow[ks]);
    }
}

// (Round 5, measured and not kept - profiles/probes/mlps_probe.hip, profiles/r5/mlps_probe.txt: a three-stage ring with the fc1 product of
//  hidden block j+1 issued before the GELU of block j (two pre-activation accumulators), and the same ring with the eight W1 / W2
//  fragments of a product requested one product ahead in explicit register sets: 0.74 ms both, against 0.75 ms for the kernel above.
//  Ablations of the kernel above at 1.94 M tokens: no weight stream 0.73, no barrier 0.67, neither 0.62, no table gather 0.61, none of
//  the three 0.52, and additionally no fragment reads at all 0.49 ms = 1.03 PFLOP/s: the floor is the two-waves-per-SIMD MFMA / VALU
//  alternation itself, not LDS latency.)

// ============================================================================ backward: input-gradient chain
// dh[m][j] = (dxout (W2 gamma))[m][j] * GELU'(h[m][j]);  dv2 = dh W1;  dxmid = dxout + LN2'(dv2; xmid);  dln_w / dln_b += .
// W2gT = (W2 * gamma[:, None])^T stored [4C][C].  Per chunk the stream brings HC rows of W1 and of W2gT; dv2^T's A operand
// (rows c, contraction over j) comes out of the SAME image of the W1 rows through the transposing LDS read, in accumulator
// order.  The transposing reads are issued as inline assembly (ppgemm_tn.hpp: behind the builtin hipcc waits with vmcnt(0) in
// front of every such read while an LDS-DMA is in flight) and waited for by hand.

// operand "row = column col0 + lane & 31, contraction over rows row0 + (e & 3) + 8 (e >> 2) + 4 half" of a streamed chunk
// image [rows][K] (sub-tiles of [rows][128 B], ms_swz): the A fragment of dv2^T += W1^T dh^T
template <class T> struct MsTrAddr {
    int lo[2], hi[2];            // LDS addresses (image offset 0) of the two 4-row groups of this lane, for even / odd 32-column blocks
    __device__ __forceinline__ void init(int lane, const char* smem) {
        const int rl = 4 * (lane >> 5) + ((lane & 15) >> 2), cl = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        const int byte = cl * (int)sizeof(T);
#ifdef RVT_EMU
        const int base = 0;
        (void)smem;
#else
        const int base = (int)(size_t)(__attribute__((address_space(3))) const char*)(smem);
#endif
        const int l = rl * 128 + ((((byte >> 4) & 7) ^ ms_swz(rl)) << 4) + (byte & 15);
        const int h = (rl + 8) * 128 + ((((byte >> 4) & 7) ^ ms_swz(rl + 8)) << 4) + (byte & 15);
        lo[0] = base + l; lo[1] = base + (l ^ 64);        // (odd block of a 64-column sub-tile: chunk index + 4)
        hi[0] = base + h; hi[1] = base + (h ^ 64);
    }
};
#ifndef RVT_EMU
__device__ __forceinline__ void ms_lgkm_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
#else
__device__ __forceinline__ void ms_lgkm_wait() {}
#endif
// bf16: operand "row = column 32 CB + lane & 31, contraction over the rows ROW0 + (e & 3) + 8 (e >> 2) + 4 half" (ROW0 a multiple of
// 16) of the image at byte offset IMG of `smem`, ROWS rows per sub-tile: two transposing reads, address = one of four per-lane
// registers + an immediate
template <int IMG, int ROWS, int ROW0, int CB>
__device__ __forceinline__ bf16x8 ms_tr_frag(const char* smem, const MsTrAddr<bf16>& a) {
    constexpr int OFF = IMG + (CB / 2) * ROWS * 128 + ROW0 * 128;
#ifdef RVT_EMU
    return frag_from_tr<bf16>(reinterpret_cast<const bf16*>(smem + OFF + a.lo[CB & 1]), reinterpret_cast<const bf16*>(smem + OFF + a.hi[CB & 1]));
#else
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
    u32x2_t lo, hi2;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a.lo[CB & 1]), "n"(OFF));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi2) : "v"(a.hi[CB & 1]), "n"(OFF));
    const u32x4 v = {lo[0], lo[1], hi2[0], hi2[1]};
    bf16x8 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
#endif
}
// fp32 parity twin: eight scalar LDS reads (no 32-bit transposing read)
template <int ROWS>
__device__ __forceinline__ f32x8 ms_tr_frag_f32(const char* img, int row0, int col0, int lane) {
    f32x8 f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int row = row0 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = col0 + (lane & 31);
        const int byte = (col % 32) * 4;
        f[e] = *reinterpret_cast<const float*>(img + (col / 32) * ROWS * 128 + row * 128 + ((((byte >> 4) & 7) ^ ms_swz(row)) << 4) + (byte & 15));
    }
    return f;
}

template <class T, int C, int WPB, int MINW>
__global__ void __launch_bounds__(64 * WPB, MINW)
mlps_bwd_dgrad_kernel(const T* __restrict__ dxout, const T* __restrict__ xmid, T* __restrict__ dxmid,
                      const float* __restrict__ ln_w, const float* __restrict__ ln_b, const T* __restrict__ W1,
                      const float* __restrict__ b1, const T* __restrict__ W2gT, float* __restrict__ dln_w,
                      float* __restrict__ dln_b, int M, float eps) {
    // chunks of 32 hidden rows (16 per tile): the two weight stages then leave room for the cotangent rows of every wave as an LDS
    // operand tile - with dxout AND LN2(xmid) both held in registers as B operands the kernel spilled (444 - 536 bytes of scratch)
    typedef MsGeom<T, C, 32> G;
    constexpr int KS = C / 16, NCB = C / 32, HID = 4 * C, NCH = G::NCH, JPC = G::JPC;
    constexpr int NPW = (G::W1C / 1024) / WPB;
    static_assert(NPW * WPB * 1024 == G::W1C, "chunk pieces must divide over the waves");
    constexpr int STAGE = 2 * G::W1C;                     // [W1 rows | W2gT rows] of the chunk, both [HC][C]
    constexpr int DT = G::KT * 32 * 128;                  // per-wave cotangent tile: [32 rows][C] as sub-tiles of [32][128 B]
    constexpr int OFF_DT = 2 * STAGE, OFF_K = OFF_DT + WPB * DT, OFF_LUT = OFF_K + G::NCONST * 4, OFF_PF = OFF_LUT + GeluTab<T>::BYTES;
    __shared__ __attribute__((aligned(16))) char smem[OFF_PF + 1024];
    float* const kst = reinterpret_cast<float*>(smem + OFF_K);
    float* const lut = reinterpret_cast<float*>(smem + OFF_LUT);
    const int tid = threadIdx.x, lane_ = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    GeluTab<T>::template fill<true>(lut, tid, 64 * WPB);
    for (int i = tid; i < C; i += 64 * WPB) { kst[G::K_LNW + i] = ln_w[i]; kst[G::K_LNB + i] = ln_b[i]; }
    for (int i = tid; i < HID; i += 64 * WPB) kst[G::K_B1 + i] = b1[i];
    __syncthreads();

    const pp_rsrc r1 = pp_make_rsrc(W1, (unsigned)(HID * C * sizeof(T))), r2 = pp_make_rsrc(W2gT, (unsigned)(HID * C * sizeof(T)));
    auto issue = [&](int ch, int stage) __attribute__((always_inline)) {
        int lane = lane_;
        opaque_vgpr(lane);
        int v[NPW];
        ms_piece_offsets<T, NPW, WPB>(v, wave, lane, G::HC, C);
#pragma unroll
        for (int i = 0; i < NPW; i++) {
            const int p = wave + i * WPB;
            pp_glds16(r1, smem, stage * STAGE + p * 1024, v[i], ch * (G::HC * C * (int)sizeof(T)));
            pp_glds16(r2, smem, stage * STAGE + G::W1C + p * 1024, v[i], ch * (G::HC * C * (int)sizeof(T)));
        }
    };

    // LayerNorm parameter gradients through the identity MFMA (mlp_chain.hpp: column sums over tokens in accumulator form)
    float aw[NCB], ab[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { aw[cb] = 0.f; ab[cb] = 0.f; }

    const int n_tiles = (M + 31) / 32;
    const int n_wg = (n_tiles + WPB - 1) / WPB;
    char* const pf_dummy = smem + OFF_PF;                 // landing zone of the L2 warm-up (never read; shared by the waves)
    issue(0, 0);
    frag_t<T> orow[KS];
    int prow = 0;
    bool pvalid = false;
    for (int twg = blockIdx.x; twg < n_wg; twg += gridDim.x) {
        int lane = lane_;
        opaque_vgpr(lane);
        const int li = lane & 31, half = lane >> 5;
        const int rb = ms_rowbase_h<T>(li, half);
        const int rbd = OFF_DT + wave * DT + rb;          // ... in this wave's cotangent tile (tile offsets are multiples of 4096: the XOR stays in the chunk bits)
        MsTrAddr<T> ta;
        ta.init(lane, smem);
        const int tile = twg * WPB + wave;
        const int row = tile * 32 + li;
        const bool valid = tile < n_tiles && row < M;
        const bool more = twg + (int)gridDim.x < n_wg;
        frag_t<T> uf[KS];
        float mean, rstd;
        {
            frag_t<T> xf[KS], df[KS];
            mc_load_row<T, C>(xf, xmid, row, valid, half);
            mc_load_row<T, C>(df, dxout, row, valid, half);
            // cotangent rows -> this wave's LDS tile (every lane re-reads exactly the pieces it wrote: same-wave LDS order, no barrier)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) ms_store_frag<T>(smem, 32, rbd, 2 * ks, df[ks]);
            mc_layernorm<T, C>(xf, uf, kst + G::K_LNW, kst + G::K_LNB, valid, half, eps, mean, rstd);
        }
        f32x16 dacc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc_zero(dacc[cb]);
        auto boundary = [&](int ch, int stage_next) __attribute__((always_inline)) {
            pp_wait_vm<0>();
            pp_barrier();
            if (ch + 1 < NCH) issue(ch + 1, stage_next);
            else if (more) issue(0, stage_next);
        };
        auto compute = [&](int ch, auto stage_c) __attribute__((always_inline)) {
            constexpr int STG = decltype(stage_c)::value;
            const char* const W1s = smem + STG * STAGE;
            const char* const W2s = W1s + G::W1C;
#pragma unroll
            for (int jj = 0; jj < JPC; jj++) {
                f32x16 h, dg;
                acc_load_rows(h, kst + G::K_B1 + ch * G::HC + 32 * jj, half);
                acc_zero(dg);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    mma32(h, ms_load_frag<T>(W1s + 32 * jj * 128, G::HC, rb, 2 * ks), uf[ks]);
                    mma32(dg, ms_load_frag<T>(W2s + 32 * jj * 128, G::HC, rb, 2 * ks), ms_load_frag<T>(smem, 32, rbd, 2 * ks));
                }
                // the transposed W1 fragments of this 32-row block: requested behind the products above (their operand fragments
                // are dead by then), in flight under the GELU' gathers, waited for in front of the dv2 products
                sched_fence();
                frag_t<T> wt[NCB][2];
                ms_static_for<JPC>([&](auto jj_c) {
                    if (decltype(jj_c)::value != jj) return;
                    ms_static_for<NCB * 2>([&](auto i_c) {
                        constexpr int JJ = decltype(jj_c)::value, CB = decltype(i_c)::value / 2, Q = decltype(i_c)::value % 2;
                        if constexpr (sizeof(T) == 2) wt[CB][Q] = ms_tr_frag<STG * STAGE, G::HC, 32 * JJ + 16 * Q, CB>(smem, ta);
                        else wt[CB][Q] = ms_tr_frag_f32<G::HC>(W1s, 32 * JJ + 16 * Q, CB * 32, lane);
                    });
                });
                float dh[16];
                GeluTab<T>::eval16(lut, h, dh);
#pragma unroll
                for (int r = 0; r < 16; r++) dh[r] = mul_nopack(dh[r], dg[r], r);
                frag_t<T> dhf[2];
                dhf[0] = arr_slot_frag<T>(dh, 0);
                dhf[1] = arr_slot_frag<T>(dh, 1);
                if constexpr (sizeof(T) == 2) ms_lgkm_wait();
#pragma unroll
                for (int cb = 0; cb < NCB; cb++)
#pragma unroll
                    for (int q = 0; q < 2; q++) mma32(dacc[cb], wt[cb][q], dhf[q]);
            }
        };
        // ---- chunk 0: the previous tile's rows leave behind the barrier
        boundary(0, 1);
        if (pvalid) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) frag_store<T>(dxmid + (size_t)prow * C + (2 * ks + half) * 8, orow[ks]);
        }
        compute(0, std::integral_constant<int, 0>());
#pragma unroll 1
        for (int ch = 1; ch < NCH - 1; ch += 2) {
            boundary(ch, 0);
            if (ch == 1 && more) {                        // next tile's xmid and dxout rows -> L2
                const int tn = tile + (int)gridDim.x * WPB;
                if (tn < n_tiles) {
                    const int rows = M - tn * 32 < 32 ? M - tn * 32 : 32;
                    const pp_rsrc rx = pp_make_rsrc(xmid + (size_t)tn * 32 * C, (unsigned)(rows * C * sizeof(T)));
                    const pp_rsrc rd = pp_make_rsrc(dxout + (size_t)tn * 32 * C, (unsigned)(rows * C * sizeof(T)));
#pragma unroll
                    for (int i = 0; i < (32 * C * (int)sizeof(T)) / 1024; i++) {
                        pp_glds16(rx, smem, (int)(pf_dummy - smem), lane * 16 + i * 1024, 0);
                        pp_glds16(rd, smem, (int)(pf_dummy - smem), lane * 16 + i * 1024, 0);
                    }
                }
            }
            compute(ch, std::integral_constant<int, 1>());
            boundary(ch + 1, 1);
            compute(ch + 1, std::integral_constant<int, 0>());
        }
        // ---- last chunk
        boundary(NCH - 1, 0);
        compute(NCH - 1, std::integral_constant<int, 1>());
        sched_fence();
        // the raw rows again (x-hat of the LayerNorm backward): an L2 round trip at the head of the epilogue - with the cotangent rows
        // AND the LayerNorm output held as operands there is no register room to fetch them underneath the last chunk (536 bytes
        // of scratch when tried)
        frag_t<T> xf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) xf[ks] = frag_load<T>(xmid + (size_t)(valid ? row : 0) * C + (2 * ks + half) * 8);
        // LayerNorm backward + residual in operand-piece form (mlp_chain.hpp)
        frag_t<T> idf[2];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int e = 0; e < 8; e++) idf[m][e] = (T)((16 * m + 8 * half + e == li) ? 1.0f : 0.0f);
        float gsum = 0.f, gxsum = 0.f;
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {                // pass 1: row sums of the LayerNorm backward, parameter gradients
            float r8[2][8];
            acc_to_rows(dacc[cb], r8);
            f32x16 tw, tb;
            acc_zero(tw);
            acc_zero(tb);
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int ks = 2 * cb + m;
                float w[8], pw[8], d[8];
                load_cols<8>(kst + G::K_LNW, 16 * ks + 8 * half, w);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    d[e] = valid ? r8[m][e] : 0.f;
                    const float xh = valid ? ((float)xf[ks][e] - mean) * rstd : 0.f;
                    const float gw = d[e] * w[e];
                    gsum += gw;
                    gxsum += gw * xh;
                    pw[e] = d[e] * xh;
                }
                mma32(tw, frag_from_float<T>(pw), idf[m]);
                mma32(tb, frag_from_float<T>(d), idf[m]);
            }
#pragma unroll
            for (int r = 0; r < 16; r++) { aw[cb] += tw[r]; ab[cb] += tb[r]; }
        }
        gsum += __shfl_xor(gsum, 32);
        gxsum += __shfl_xor(gxsum, 32);
        const float m1 = gsum / (float)C, m2 = gxsum / (float)C;
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {                // pass 2: the rows again (eight half-exchanges per block; no fp32 copy kept:
            float r8[2][8];                               // left to itself hipcc keeps pass 1's 64 exchanged values - in scratch)
#ifndef RVT_EMU
            asm volatile("" : "+v"(dacc[cb]));
#endif
            acc_to_rows(dacc[cb], r8);
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const int ks = 2 * cb + m;
                float w[8], o[8];
                load_cols<8>(kst + G::K_LNW, 16 * ks + 8 * half, w);
                const frag_t<T> dfk = ms_load_frag<T>(smem, 32, rbd, 2 * ks);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float xh = ((float)xf[ks][e] - mean) * rstd;
                    o[e] = (float)dfk[e] + rstd * (r8[m][e] * w[e] - m1 - xh * m2);
                }
                orow[ks] = frag_from_float<T>(o);
            }
        }
        prow = row;
        pvalid = valid;
    }
    const int li = lane_ & 31, half = lane_ >> 5;
    if (pvalid) {
#pragma unroll
        for (int ks = 0; ks < KS; ks++) frag_store<T>(dxmid + (size_t)prow * C + (2 * ks + half) * 8, orow[ks]);
    }
    // fold the two halves and the waves: one atomic per channel per workgroup
    pp_wait_vm<0>();
    __syncthreads();                                      // weight stages are dead: the LDS becomes reduction scratch
    float* const red = reinterpret_cast<float*>(smem);    // [WPB][dln_w C | dln_b C]
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

// =========================================================================== backward: weight gradients (bf16, C = 128)
//   dW1[j][c] = sum_t dh[t][j] v2[t][c],  db1[j] = sum_t dh[t][j],  S2[c][j] = sum_t dxout[t][c] g[t][j],  cs2[c] = sum_t dxout[t][c]
//   with v2 = LN2(xmid), h = v2 W1^T + b1, g = GELU(h), dh = (dxout (gamma W2)) GELU'(h)      (everything recomputed)
// Weight-stationary like mlpc_bwd_wgrad_kernel (mlp_chain.hpp): wave w of a workgroup owns 32 hidden columns and their
// 2 x [32][128] fp32 accumulators (128 registers); a workgroup covers HALF of the hidden axis (256 columns; the two halves of
// a tile stream run as neighbours on one XCD and write disjoint halves of the same partial record).  What changes at C = 128 is
// where everything else lives, because 128 accumulators + 2 x 32 weight-fragment registers leave a wave nothing to work with:
//   * the wave's rows of W1 stay in registers (B operands of h); the workgroup's 256 rows of (gamma W2)^T sit in LDS (68 KiB,
//     row pitch 272 B) and are read as B operands of dg;
//   * token tiles arrive by LDS-DMA (no staging registers): xmid / dxout rows of tile i+2 are requested while tile i is multiplied,
//     into a ring of three [32][256 B] tiles each (16-byte chunk index XOR row & 15, applied to the source address); LayerNorm runs
//     IN PLACE on the landed xmid tile one phase ahead of its use (16 threads per row, DPP row sums); ONE barrier per tile;
//   * GELU / GELU': Phi from the 32-KiB nearest-entry table, the density term of GELU' = Phi + x phi(x) from one v_exp (a
//     {Phi, GELU'} pair table as in the C = 64 kernel does not fit beside the weight image).
// Partial results per tile stream in `ws`, laid out as mlp_fold_partials expects with grid = number of streams:
// [dW1: S x 4C x C][S2: S x C x 4C][db1: 2 S x 4C][cs2: S x C].
struct MswGeom {
    static constexpr int C = 128, HID = 512, HB = 256, KS = 8, NCB = 4;
    static constexpr int TILE = 32 * 256;                 // [32 tokens][128 channels] bf16, pitch 256 B, swizzled
    static constexpr int W2P = 272, W2IMG = HB * W2P;
    static constexpr int OFF_XV = 0, OFF_DX = 3 * TILE, OFF_W2 = 6 * TILE, OFF_LUT = OFF_W2 + W2IMG, OFF_K = OFF_LUT + GELU_NLUT_BYTES;
    static constexpr int SMEM = OFF_K + 2 * C * 4;
};
__device__ __forceinline__ int msw_tile_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

template <int TAG>
__global__ void __launch_bounds__(512, 2)
mlps_bwd_wgrad_kernel(const bf16* __restrict__ dxout, const bf16* __restrict__ xmid, const float* __restrict__ ln_w,
                      const float* __restrict__ ln_b, const bf16* __restrict__ W1, const float* __restrict__ b1,
                      const bf16* __restrict__ W2gT, float* __restrict__ ws, int M, float eps, int nstreams) {
    typedef bf16 T;
    typedef MswGeom G;
    constexpr int C = G::C, KS = G::KS, NCB = G::NCB, HID = G::HID, TILE = G::TILE;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM];
    float* const lut = reinterpret_cast<float*>(smem + G::OFF_LUT);
    float* const kst = reinterpret_cast<float*>(smem + G::OFF_K);          // ln_w | ln_b
    const int tid = threadIdx.x, lane_ = tid & 63;
    const int wave = wave_uniform(tid >> 6);
    // workgroup -> (tile stream, hidden half): ids b and b + 8 are the two halves of one stream (same XCD, dispatched together)
    const int bid = blockIdx.x;
    const int hb = (bid >> 3) & 1, stream = (bid >> 4) * 8 + (bid & 7);
    if (stream >= nstreams) return;
    const int n_tiles = (M + 31) / 32;
    const int count = stream < n_tiles ? (n_tiles - stream + nstreams - 1) / nstreams : 0;     // tiles stream, stream + S, ...

    gelu_nlut_fill<false>(lut, tid, 512);
    for (int i = tid; i < C; i += 512) { kst[i] = ln_w[i]; kst[C + i] = ln_b[i]; }
    for (int f = tid; f < G::HB * 16; f += 512) {         // the workgroup's rows of (gamma W2)^T
        const int r = f >> 4, c = f & 15;
        *reinterpret_cast<u32x4*>(smem + G::OFF_W2 + r * G::W2P + c * 16) =
            *reinterpret_cast<const u32x4*>(W2gT + (size_t)(hb * G::HB + r) * C + c * 8);
    }
    // this wave's rows j = 256 hb + 32 wave + li of W1, as B operands (k-step ks: channels 16 ks + 8 half ..)
    const int jrow = hb * G::HB + 32 * wave + (lane_ & 31);
    frag_t<T> w1f[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) w1f[ks] = frag_load<T>(W1 + (size_t)jrow * C + 16 * ks + 8 * (lane_ >> 5));
    const float b1v = b1[jrow];

    f32x16 dw1[NCB], s2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) { acc_zero(dw1[cb]); acc_zero(s2[cb]); }
    float db1 = 0.f;
    // column sums of dxout: wave w sums the eight tokens of ONE transposed dxout fragment per tile - channel block w & 3, token group
    // w >> 2 - that it reads anyway (lane = channel): one accumulator register per wave (the staging role kept eight per thread)
    float cs1 = 0.f;
    const int cs_cb = wave & 3, cs_q = wave >> 2;

    // ---- tile stream: piece `wave` of the xmid tile and of the dxout tile (4 rows x 256 B each) per wave
    auto issue = [&](int k, int buf) __attribute__((always_inline)) {
        int lane = lane_;
        opaque_vgpr(lane);
        const int tile = stream + k * nstreams;
        const int rows = tile < n_tiles ? (M - tile * 32 < 32 ? M - tile * 32 : 32) : 0;
        const pp_rsrc rx = pp_make_rsrc(xmid + (size_t)(rows ? tile : 0) * 32 * C, (unsigned)(rows * C * 2));
        const pp_rsrc rd = pp_make_rsrc(dxout + (size_t)(rows ? tile : 0) * 32 * C, (unsigned)(rows * C * 2));
        const int r = 4 * wave + (lane >> 4), c = (lane & 15) ^ (r & 15);
        const int voff = r * 256 + c * 16;
        pp_glds16(rx, smem, G::OFF_XV + buf * TILE + wave * 1024, voff, 0);
        pp_glds16(rd, smem, G::OFF_DX + buf * TILE + wave * 1024, voff, 0);
    };
    // ---- staging role: LayerNorm of tile k+1 in place (16 threads per row, 8 channels each)
    auto layernorm = [&](int buf, bool live) __attribute__((always_inline)) {
        if (!live) return;
        int t = tid;
        opaque_vgpr(t);                                   // (the per-thread tile offset is recomputed per tile: kept across the loop it was spilled,
        const int srow = t >> 4, spc = t & 15;            //  and a scratch reload waits with vmcnt(0) = for the tile stream)
        char* const px = smem + G::OFF_XV + buf * TILE + msw_tile_off(srow, spc);
        const bf16x8 xb = *reinterpret_cast<const bf16x8*>(px);
        float x[8], s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) { x[i] = (float)xb[i]; s += x[i]; }
        const float mean = row16_sum(s) * (1.0f / C);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; i++) { x[i] -= mean; ss += x[i] * x[i]; }
        const float rstd = 1.0f / sqrtf(row16_sum(ss) * (1.0f / C) + eps);
        float w[8], bb[8];
        load_cols<8>(kst, 8 * spc, w);
        load_cols<8>(kst + C, 8 * spc, bb);
        bf16x8 v;
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = (T)fmaf(x[i] * rstd, w[i], bb[i]);
        *reinterpret_cast<bf16x8*>(px) = v;
    };
    // ---- compute role
    auto compute = [&](int buf) __attribute__((always_inline)) {
        int lane = lane_;
        opaque_vgpr(lane);
        const int li = lane & 31, half = lane >> 5;
        const char* const v2t = smem + G::OFF_XV + buf * TILE;
        const char* const dxt = smem + G::OFF_DX + buf * TILE;
        const int arow = li * 256 + ((half ^ (li & 15)) << 4);                   // token row li, chunk (2 ks + half): ^ (ks << 5)
        const int wrow = G::OFF_W2 + (32 * wave + li) * G::W2P + half * 16;      // + ks * 32
        // accumulator column = hidden j (this lane), registers = the tile's tokens
        f32x16 h, dg;
#pragma unroll
        for (int r = 0; r < 16; r++) h[r] = b1v;
        acc_zero(dg);
        // (fragment reads in two groups of four: all eight in flight at once cost the registers a spilled W1 fragment needs)
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (ks == KS / 2) sched_fence();
            mma32(h, *reinterpret_cast<const frag_t<T>*>(v2t + (arow ^ (ks << 5))), w1f[ks]);
        }
        sched_fence();
        // transposed-operand addresses: token rows 16 q + rl (+ 8), channel block cb: base ^ (cb << 6) + q * 4096
        const int rl = 4 * half + ((lane & 15) >> 2), c3 = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), sub = 8 * (lane & 1);
        const int t_lo = rl * 256 + ((c3 ^ rl) << 4) + sub, t_hi = (rl + 8) * 256 + ((c3 ^ rl ^ 8) << 4) + sub;
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            if (ks % 2 == 0 && ks) sched_fence();
            mma32(dg, *reinterpret_cast<const frag_t<T>*>(dxt + (arow ^ (ks << 5))), *reinterpret_cast<const frag_t<T>*>(smem + wrow + ks * 32));
        }
        sched_fence();
        frag_t<T> gf[2], dhf[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {                     // eight values at a time (registers): slot q of the operands = accumulator registers 8q .. 8q+7
            int idx[8];
#pragma unroll
            for (int e = 0; e < 8; e++) idx[e] = gelu_nlut_index(h[8 * q + e]);
            sched_fence();
            float ph[8];
#pragma unroll
            for (int e = 0; e < 8; e++) ph[e] = lut[idx[e]];
            sched_fence();
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float x = h[8 * q + e];
                const float ex = fast_exp2(x * x * -0.72134752044448170f);       // exp(-x^2 / 2)
                const float gpv = fmaf(x * 0.3989422804014327f, ex, ph[e]);      // GELU'(x) = Phi(x) + x phi(x)
                const float d = mul_nopack(dg[8 * q + e], gpv, e);
                db1 += d;
                gf[q][e] = (T)mul_nopack(x, ph[e], e);                           // GELU(x)
                dhf[q][e] = (T)d;
            }
        }
        // dW1 / S2: contraction over the tokens (transposing reads issued as assembly: LDS-DMA is in flight, see ms_tr_frag)
#ifdef RVT_EMU
        const char* const vb = v2t;
        const char* const db_ = dxt;
#else
        const int vb = (int)(size_t)(__attribute__((address_space(3))) const char*)(v2t);
        const int db_ = (int)(size_t)(__attribute__((address_space(3))) const char*)(dxt);
#endif
        ms_static_for<NCB>([&](auto cb_c) {
            constexpr int CB = decltype(cb_c)::value;
            const int alo = t_lo ^ (CB << 6), ahi = t_hi ^ (CB << 6);
            frag_t<T> vT[2], dT[2];
            ms_static_for<2>([&](auto q_c) {
                constexpr int Q = decltype(q_c)::value;
#ifdef RVT_EMU
                vT[Q] = frag_from_tr<T>(reinterpret_cast<const bf16*>(vb + alo + Q * 4096), reinterpret_cast<const bf16*>(vb + ahi + Q * 4096));
                dT[Q] = frag_from_tr<T>(reinterpret_cast<const bf16*>(db_ + alo + Q * 4096), reinterpret_cast<const bf16*>(db_ + ahi + Q * 4096));
#else
                typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
                u32x2_t a0, a1, b0, b1_;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a0) : "v"(vb + alo), "n"(Q * 4096));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(a1) : "v"(vb + ahi), "n"(Q * 4096));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b0) : "v"(db_ + alo), "n"(Q * 4096));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(b1_) : "v"(db_ + ahi), "n"(Q * 4096));
                const u32x4 va = {a0[0], a0[1], a1[0], a1[1]}, vd = {b0[0], b0[1], b1_[0], b1_[1]};
                __builtin_memcpy(&vT[Q], &va, 16);
                __builtin_memcpy(&dT[Q], &vd, 16);
#endif
            });
            ms_lgkm_wait();
            if (CB == cs_cb) {                          // (wave-uniform)
                const frag_t<T> dq = cs_q ? dT[1] : dT[0];
#pragma unroll
                for (int e = 0; e < 8; e++) cs1 += (float)dq[e];
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                mma32(dw1[CB], dhf[q], vT[q]);          // rows j, columns c
                mma32(s2[CB], dT[q], gf[q]);            // rows c, columns j
            }
        });
    };

    const bool mfma_first = wave < 4;
    issue(0, 0);
    issue(1, 1);
    pp_wait_vm<2>();                                      // this wave's pieces of tile 0 (table / W2 image stores are LDS writes: barrier)
    __syncthreads();
    layernorm(0, count > 0);
    for (int k = 0; k < count; k++) {
        // tile k+1 has landed (this wave's pieces: vmcnt; everybody's: barrier); LayerNorm of tile k is visible; tile k-1 is done with
        pp_wait_vm<0>();
        lds_barrier();
        issue(k + 2, (k + 2) % 3);
        const int buf = k % 3, nbuf = (k + 1) % 3;
        if (mfma_first) compute(buf);
        layernorm(nbuf, k + 1 < count);
        if (!mfma_first) compute(buf);
    }
    pp_wait_vm<0>();                                      // (trailing pieces still write LDS)

    const int lane = lane_, li = lane & 31;
    const size_t nwg = nstreams, wg = stream;
    float* const p_dw1 = ws + wg * (size_t)(HID * C);
    float* const p_s2 = ws + nwg * (size_t)(HID * C) + wg * (size_t)(C * HID);
    float* const p_db1 = ws + 2 * nwg * (size_t)(HID * C) + (wg * 2) * (size_t)HID;
    float* const p_cs2 = ws + 2 * nwg * (size_t)(HID * C) + 2 * nwg * (size_t)HID + wg * (size_t)C;
    const int j0 = hb * G::HB + 32 * wave;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            p_dw1[(size_t)(j0 + acc_row(r, lane)) * C + 32 * cb + li] = dw1[cb][r];
            p_s2[(size_t)(32 * cb + acc_row(r, lane)) * HID + j0 + li] = s2[cb][r];
        }
    db1 += __shfl_xor(db1, 32);
    if (lane < 32) {
        p_db1[j0 + li] = db1;
        p_db1[HID + j0 + li] = 0.f;
    }
    // column sums of dxout: lane = channel 32 (w & 3) + li, the two lane halves and the two waves of a channel block hold disjoint
    // token subsets: fold through LDS (the tiles are done with); the first hidden half's copy is the record
    float* const red = reinterpret_cast<float*>(smem);
    __syncthreads();
    cs1 += __shfl_xor(cs1, 32);
    if (lane < 32) red[cs_q * C + 32 * cs_cb + li] = cs1;
    __syncthreads();
    if (hb == 0 && tid < C) p_cs2[tid] = red[tid] + red[C + tid];
}

}  // namespace rvt
