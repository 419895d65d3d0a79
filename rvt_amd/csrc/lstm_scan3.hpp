// ConvLSTM with the time loop inside the kernel for the WIDE stages (bf16, C = 256 / 512: stages 3 - 4 of RVT-Base, stage 4 of
// RVT-Tiny) — reference models/layers/rnn.py:43-67 driven by the loop of modules/detection.py:131-148, and its BPTT.
//
// lstm_scan.hpp / lstm_scan2.hpp keep the cell's weights on chip (LDS or registers); at C >= 256 they are 1 - 4 MB and the
// recurrence ran as one GEMM + gate kernel per step (42 x 3 launches per stage and step of training, each with 23 - 90 tiles on
// 256 CUs, the activated gates and the fp32 cell state crossing HBM twice per step).  Here a workgroup owns a tile of 32 RB tokens
// for ALL T steps (a 1x1-conv cell is independent per token) and STREAMS the weights from L2 every step:
//   * wave w owns the channels 64 w .. 64 w + 63 (two blocks of 32) and computes, for them, the four gate blocks f, i, o, g as
//     32x32 MFMA column blocks: the four gates of a (token, channel) land in the same lane and register index, the gate math needs
//     no staging, and c_t / dc_t / dh_t persist in registers in that layout across the time loop;
//   * the weights are PRE-PACKED in operand order (`lstm_scan3_pack`: one contiguous KiB per (wave, k-step, block, gate), i.e. one
//     fully coalesced buffer load per MFMA B operand) and arrive in a register ring that runs PD k-steps ahead of the MFMAs and
//     straight across step and tile boundaries (the next step's first k-steps ride behind the gate math of this one);
//   * x_t rows (forward) / dH_t rows (backward) arrive by LDS-DMA one step ahead (swizzle on the source address), h_t lives in an
//     LDS operand tile that is both the next step's A operand and the staging for the coalesced row store;
//   * the forward saves the ACTIVATED gates and a bf16 copy of c_t in "register-dump" order ([t][32-token block][wave][block]
//     [gate][8 registers][lane][8] — each wave store is one contiguous KiB); the reverse scan reads them back in the same order, so
//     it needs no recompute product and only W^T: one streamed product per step in each direction.
// HBM per token-step: forward x + h + c + 4 gates = 7 rows of C; backward dH + c + 4 gates + dz (4) + dx = 11 (per-step route: 16 / 22).
#pragma once
#include "common.hpp"
#include "mlp.hpp"
#include "ppgemm.hpp"

namespace rvt {

template <int C, int RB> struct Scan3Geom {
    static constexpr int NW = C / 64, NT = 64 * NW, TM = 32 * RB;
    static constexpr int KT = C / 64;                  // 128-byte K-subtiles of a [.][C] bf16 operand matrix
    static constexpr int TILE = KT * TM * 128;         // bytes of a [TM][C] operand tile
    static constexpr int NKF = 2 * C / 16;             // k-steps of z = [x | h] W^T
    static constexpr int NKB = 4 * C / 16;             // k-steps of [dx | dh] = dz W
    static constexpr int PD = RB == 1 ? 4 : 8;         // k-steps (of 4 operand pieces = 4 KiB per wave) the weight ring runs ahead
    static constexpr int WGS = RB == 1 ? 512 / C : 1;  // workgroups per CU the forward is built for (RB = 1: two waves per SIMD, so that one workgroup's gate math overlaps another's MFMAs)
    static_assert(C % 64 == 0 && NKF % PD == 0 && NKB % PD == 0, "geometry");
};

__device__ __forceinline__ bf16x8 scan3_as_frag(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

// [TM][C] rows (zeros beyond `rows`) -> swizzled LDS operand tile by LDS-DMA: one instruction = 8 rows x 128 B of one K-subtile;
// lane (row r, physical chunk p) fetches logical chunk p ^ swizzle(r) (the image is lane-linear: the swizzle goes on the source)
template <int C, int RB> __device__ __forceinline__ void scan3_dma_tile(char* smem, int tile_off, const bf16* src, int rows, int wave, int lane) {
    typedef Scan3Geom<C, RB> G;
    const pp_rsrc rs = pp_make_rsrc(src, src != nullptr ? (unsigned)(rows * C * 2) : 0u);
    constexpr int NI = G::KT * (G::TM / 8), PER = NI / G::NW;
    static_assert(NI % G::NW == 0, "DMA pieces per wave");
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const int u = wave * PER + q;
        const int kt = u / (G::TM / 8), rg = u % (G::TM / 8);
        const int r = 8 * rg + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7) ^ ((r >> 4) & 7);
        pp_glds16(rs, smem, tile_off + kt * G::TM * 128 + rg * 1024, r * C * 2 + kt * 128 + c * 16, 0);
    }
}

// ====================================================================================================== forward
// x_all [Tn][M][C], Hall [Tn+1][M][C] (slot 0 = incoming h, filled by the caller), c0 fp32 [M][C] or null, c_last fp32 [M][C],
// Wp = lstm_scan3_pack(W) (see rvt_lstm_scan3_pack_fwd), bias fp32 [4C] natural order f,i,o,g (rnn.py:57-61),
// Csave / gsave: register-dump buffers of Tn * rows_pad * C / * 4C elements (rows_pad = M rounded up to TM), nullable together.
template <int C, int RB>
__global__ void __launch_bounds__(C, (Scan3Geom<C, RB>::WGS * C) / 256)
lstm_scan3_fwd_kernel(const bf16* __restrict__ x_all, bf16* __restrict__ Hall, const float* __restrict__ c0, float* __restrict__ c_last,
                      bf16* __restrict__ Csave, const bf16* __restrict__ Wp, const float* __restrict__ bias, bf16* __restrict__ gsave,
                      int M, int Tn) {
    typedef bf16 T;
    typedef Scan3Geom<C, RB> G;
    constexpr int NW = G::NW, NT = G::NT, TM = G::TM, TILE = G::TILE, NKF = G::NKF, PD = G::PD;
    __shared__ __attribute__((aligned(1024))) char smem[4 * TILE];
    constexpr int OFF_X = 0, OFF_H = 2 * TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int li = lane & 31, half = lane >> 5;
    const size_t MC = (size_t)M * C;
    int ch[2];
    float nbf[2], nbi[2], nbo[2], tbg[2];
    int off0[RB][2];
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
        ch[cb] = 64 * wave + 32 * cb + li;
        nbf[cb] = bias[ch[cb]] * -1.4426950408889634f;
        nbi[cb] = bias[C + ch[cb]] * -1.4426950408889634f;
        nbo[cb] = bias[2 * C + ch[cb]] * -1.4426950408889634f;
        tbg[cb] = bias[3 * C + ch[cb]] * 2.8853900817779268f;
#pragma unroll
        for (int i = 0; i < RB; i++)
            off0[i][cb] = (int)(reinterpret_cast<char*>(opm_elem_ptr<T>(smem, TM, i * 32 + 4 * half, ch[cb])) - smem);
    }
    // this wave's packed weights: one stream of 2 NKF k-steps (channel block 0, then block 1) x 4 operand pieces (gates) of 1 KiB
    const pp_rsrc rw = pp_make_rsrc(Wp + (size_t)wave * NKF * 8 * 512, (unsigned)(NKF * 8 * 1024));
    const int lane16 = lane * 16;
    u32x4 bq[PD][4];
    auto issue = [&](u32x4 (&slot)[4], int p) {
#pragma unroll
        for (int j = 0; j < 4; j++) slot[j] = pp_load16(rw, ((p * 4 + j) << 10) + lane16);
    };
#pragma unroll
    for (int j = 0; j < PD; j++) issue(bq[j], j);

    const int n_tiles = (M + TM - 1) / TM;
    const size_t ntb = (size_t)n_tiles * RB;               // 32-token blocks per time step (dump addressing)
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m0 = tile * TM;
        const int rows = M - m0 < TM ? M - m0 : TM;
        float creg[RB][2][16];
        lds_barrier();                                     // the previous tile's last row copy-out has read its LDS tile
        scan3_dma_tile<C, RB>(smem, OFF_H, Hall + (size_t)m0 * C, rows, wave, lane);
        scan3_dma_tile<C, RB>(smem, OFF_X, x_all + (size_t)m0 * C, rows, wave, lane);
#pragma unroll
        for (int i = 0; i < RB; i++)
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = m0 + i * 32 + acc_row(r, lane);
                    creg[i][cb][r] = (c0 != nullptr && row < M) ? c0[(size_t)row * C + ch[cb]] : 0.f;
                }
        pp_wait_vm<0>();
        lds_barrier();
        for (int t = 0; t < Tn; t++) {
            const int cur = t & 1;
            const char* const Ax = smem + OFF_X + cur * TILE;
            const char* const Ah = smem + OFF_H + cur * TILE;
            char* const An = smem + OFF_H + (cur ^ 1) * TILE;
            // x_{t+1}: lands while this step multiplies (every wave waits for younger weight loads before the step's barrier, and
            // loads return in order, so no wait of its own)
            if (t + 1 < Tn) scan3_dma_tile<C, RB>(smem, OFF_X + (cur ^ 1) * TILE, x_all + (size_t)(t + 1) * MC + (size_t)m0 * C, rows, wave, lane);
            // the two channel blocks of this wave one after the other: 4 RB accumulator blocks live at a time, the token rows are read
            // from LDS twice (cheap), the weight stream is the same bytes in a different order
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                f32x16 acc[RB][4];
#pragma unroll
                for (int i = 0; i < RB; i++)
#pragma unroll
                    for (int g = 0; g < 4; g++) acc_zero(acc[i][g]);
#pragma unroll 1
                for (int k0 = 0; k0 < NKF; k0 += PD) {
#pragma unroll
                    for (int j = 0; j < PD; j++) {
                        const int ks = k0 + j;
                        const char* const A = ks < C / 16 ? Ax : Ah;       // K order: x columns, then h columns (rnn.py:52)
                        const int fcg = ((ks & (C / 16 - 1)) << 1) + half;
                        frag_t<T> a[RB];
#pragma unroll
                        for (int i = 0; i < RB; i++) a[i] = opm_load_frag<T>(A, TM, i * 32 + li, fcg);
#pragma unroll
                        for (int g = 0; g < 4; g++)
#pragma unroll
                            for (int i = 0; i < RB; i++) mma32(acc[i][g], a[i], scan3_as_frag(bq[j][g]));
                        int pn = cb * NKF + ks + PD;
                        if (pn >= 2 * NKF) pn -= 2 * NKF;                  // the next step's first pieces
                        issue(bq[j], pn);
                    }
                }
                // gates (rnn.py:57-67) in registers; h_t -> the other h tile; activated gates and c_t -> register-dump buffers
#pragma unroll
                for (int i = 0; i < RB; i++) {
                    const size_t blk = (((size_t)t * ntb + (size_t)tile * RB + i) * NW + wave) * 2 + cb;
#pragma unroll
                    for (int h8 = 0; h8 < 2; h8++) {
                        float f8[8], i8[8], o8[8], g8[8], c8[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const int r = 8 * h8 + e;
                            const float f = sigmoid_zb(acc[i][0][r], nbf[cb]);
                            const float ig = sigmoid_zb(acc[i][1][r], nbi[cb]);
                            const float o = sigmoid_zb(acc[i][2][r], nbo[cb]);
                            const float g = tanh_zb(acc[i][3][r], tbg[cb]);
                            const float cn = f * creg[i][cb][r] + ig * g;
                            creg[i][cb][r] = cn;
                            *reinterpret_cast<T*>(An + acc_elem_off(off0[i][cb], r)) = (T)(o * tanh_f(cn));
                            f8[e] = f; i8[e] = ig; o8[e] = o; g8[e] = g; c8[e] = cn;
                        }
                        if (gsave != nullptr) {
                            T* const gd = gsave + (blk * 8 + h8) * 512 + lane * 8;       // [gate][h8][lane][8]
                            frag_store<T>(gd, frag_from_float<T>(f8));
                            frag_store<T>(gd + 2 * 512, frag_from_float<T>(i8));
                            frag_store<T>(gd + 4 * 512, frag_from_float<T>(o8));
                            frag_store<T>(gd + 6 * 512, frag_from_float<T>(g8));
                            frag_store<T>(Csave + (blk * 2 + h8) * 512 + lane * 8, frag_from_float<T>(c8));
                        }
                    }
                }
            }
            lds_barrier();                                 // h_t tile complete, x_{t+1} landed, every wave done with this step's tiles
            {                                              // h_t rows -> HBM in 16-byte pieces (the stage's output feature)
                T* const hdst = Hall + (size_t)(t + 1) * MC + (size_t)m0 * C;
                constexpr int GP = C / 8;
#pragma unroll
                for (int q = 0; q < TM * GP / NT; q++) {
                    const int f = tid + q * NT;
                    const int row = f / GP, cg = f % GP;
                    if (row < rows) frag_store<T>(hdst + (size_t)row * C + cg * 8, opm_load_frag<T>(An, TM, row, cg));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RB; i++)
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = m0 + i * 32 + acc_row(r, lane);
                    if (row < M) c_last[(size_t)row * C + ch[cb]] = creg[i][cb][r];
                }
    }
}

// ===================================================================================================== backward
// Reverse scan on the saved gates.  dH [Tn][M][C] (cotangent of Hall[1..], null = zeros), dc_last fp32 [M][C] (null = zeros),
// Wtp = lstm_scan3_pack of W^T (rvt_lstm_scan3_pack_bwd), gsave / Csave as written by the forward with the SAME M.
// Outputs: dx_all [Tn][M][C], dz_all [Tn][M][4C] (natural gate order, for the weight-gradient GEMM), dh0 [M][C], dc0 fp32 [M][C].
template <int C>
__global__ void __launch_bounds__(C)
lstm_scan3_bwd_kernel(const bf16* __restrict__ gsave, const bf16* __restrict__ Csave, const float* __restrict__ c0,
                      const bf16* __restrict__ dH, const float* __restrict__ dc_last, const bf16* __restrict__ Wtp,
                      bf16* __restrict__ dx_all, bf16* __restrict__ dz_all, bf16* __restrict__ dh0, float* __restrict__ dc0,
                      int M, int Tn, int rb_fwd) {
    typedef bf16 T;
    typedef Scan3Geom<C, 1> G;
    constexpr int NW = G::NW, NT = G::NT, TM = G::TM, TILE = G::TILE, NKB = G::NKB, PD = 8;
    constexpr int DZ = 4 * TILE;                           // [TM][4C] operand tile
    // (measured and dropped: 80 KiB / 256 registers for two workgroups per CU - dx as 2-byte stores straight from the accumulators,
    // one dH tile: 1.45 ms against 1.21 for this one-per-CU form at M = 23040, T = 21; the reverse scan is bound by the W^T stream)
    __shared__ __attribute__((aligned(1024))) char smem[DZ + 3 * TILE];
    constexpr int OFF_DZ = 0, OFF_D = DZ, OFF_SX = DZ + 2 * TILE;
    char* const Adz = smem + OFF_DZ;
    char* const Sx = smem + OFF_SX;

    const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
    const int li = lane & 31, half = lane >> 5;
    const size_t MC = (size_t)M * C;
    int ch[2], off0[2];
#pragma unroll
    for (int cb = 0; cb < 2; cb++) {
        ch[cb] = 64 * wave + 32 * cb + li;
        off0[cb] = (int)(reinterpret_cast<char*>(opm_elem_ptr<T>(smem, TM, 4 * half, ch[cb])) - smem);
    }
    // element (row, g C + channel) of the [TM][4C] dz operand: gate g = K-subtiles g KT ..
    auto dz_off = [&](int cb, int g, int r) -> int { return g * TILE + acc_elem_off(off0[cb], r); };
    // this wave's packed W^T: NKB k-steps x 4 operand pieces (block cb, part: x / h columns) of 1 KiB
    const pp_rsrc rw = pp_make_rsrc(Wtp + (size_t)wave * NKB * 4 * 512, (unsigned)(NKB * 4 * 1024));
    const int lane16 = lane * 16;
    u32x4 bq[PD][4];
    auto issue = [&](u32x4 (&slot)[4], int ks) {
#pragma unroll
        for (int j = 0; j < 4; j++) slot[j] = pp_load16(rw, ((ks * 4 + j) << 10) + lane16);
    };
#pragma unroll
    for (int j = 0; j < PD; j++) issue(bq[j], j);

    const int n_tiles = (M + TM - 1) / TM;                 // 32-token blocks
    const size_t ntb = (size_t)((M + 32 * rb_fwd - 1) / (32 * rb_fwd)) * rb_fwd;       // ... as the forward counted them
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m0 = tile * TM;
        const int rows = M - m0 < TM ? M - m0 : TM;
        float dh_rec[2][16], dc_rec[2][16];
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + acc_row(r, lane);
                dh_rec[cb][r] = 0.f;
                dc_rec[cb][r] = (dc_last != nullptr && row < M) ? dc_last[(size_t)row * C + ch[cb]] : 0.f;
            }
        lds_barrier();                                     // previous tile fully consumed
        scan3_dma_tile<C, 1>(smem, OFF_D + ((Tn - 1) & 1) * TILE, dH != nullptr ? dH + (size_t)(Tn - 1) * MC + (size_t)m0 * C : nullptr,
                             rows, wave, lane);
        pp_wait_vm<0>();
        lds_barrier();
        for (int t = Tn - 1; t >= 0; t--) {
            const char* const Sd = smem + OFF_D + (t & 1) * TILE;
            // dH_{t-1}: lands while this step works (the waits for younger loads cover it: loads return in order)
            if (t > 0) scan3_dma_tile<C, 1>(smem, OFF_D + ((t - 1) & 1) * TILE, dH != nullptr ? dH + (size_t)(t - 1) * MC + (size_t)m0 * C : nullptr,
                                            rows, wave, lane);
            // ---- gate backward (autograd of rnn.py:57-67) in registers; dz -> LDS as the product's A operand ----
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                const size_t blk = (((size_t)t * ntb + tile) * NW + wave) * 2 + cb;
                const size_t blkp = (((size_t)(t > 0 ? t - 1 : 0) * ntb + tile) * NW + wave) * 2 + cb;
#pragma unroll
                for (int h8 = 0; h8 < 2; h8++) {
                    float f8[8], i8[8], o8[8], g8[8], cp8[8];
                    const T* const gs = gsave + (blk * 8 + h8) * 512 + lane * 8;      // [gate][h8][lane][8]
                    frag_to_float<T>(frag_load<T>(gs), f8);
                    frag_to_float<T>(frag_load<T>(gs + 2 * 512), i8);
                    frag_to_float<T>(frag_load<T>(gs + 4 * 512), o8);
                    frag_to_float<T>(frag_load<T>(gs + 6 * 512), g8);
                    if (t > 0) {
                        frag_to_float<T>(frag_load<T>(Csave + (blkp * 2 + h8) * 512 + lane * 8), cp8);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const int row = m0 + acc_row(8 * h8 + e, lane);
                            cp8[e] = (c0 != nullptr && row < M) ? (float)(T)c0[(size_t)row * C + ch[cb]] : 0.f;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int r = 8 * h8 + e;
                        const float f = f8[e], ig = i8[e], o = o8[e], g = g8[e], cp = cp8[e];
                        const float dh = (float)*reinterpret_cast<const T*>(Sd + acc_elem_off(off0[cb], r)) + dh_rec[cb][r];
                        const float tc = tanh_f(f * cp + ig * g);
                        const float dc = dc_rec[cb][r] + dh * o * (1.f - tc * tc);
                        *reinterpret_cast<T*>(Adz + dz_off(cb, 0, r)) = (T)(dc * cp * f * (1.f - f));
                        *reinterpret_cast<T*>(Adz + dz_off(cb, 1, r)) = (T)(dc * g * ig * (1.f - ig));
                        *reinterpret_cast<T*>(Adz + dz_off(cb, 2, r)) = (T)(dh * tc * o * (1.f - o));
                        *reinterpret_cast<T*>(Adz + dz_off(cb, 3, r)) = (T)(dc * ig * (1.f - g * g));
                        dc_rec[cb][r] = dc * f;
                    }
                }
            }
            lds_barrier();                                 // dz tile complete; dH tile of this step consumed
            {                                              // dz rows -> HBM (weight-gradient GEMM), fire and forget
                T* const zdst = dz_all + (size_t)t * MC * 4 + (size_t)m0 * 4 * C;
                constexpr int GP = 4 * C / 8;
#pragma unroll
                for (int q = 0; q < TM * GP / NT; q++) {
                    const int f = tid + q * NT;
                    const int row = f / GP, cg = f % GP;
                    if (row < rows) frag_store<T>(zdst + (size_t)row * 4 * C + cg * 8, opm_load_frag<T>(Adz, TM, row, cg));
                }
            }
            // ---- [dx_t | dh_{t-1}] = dz W: this wave's 64 x-columns and its 64 h-columns ----
            f32x16 acc2[2][2];
#pragma unroll
            for (int cb = 0; cb < 2; cb++) { acc_zero(acc2[cb][0]); acc_zero(acc2[cb][1]); }
#pragma unroll 1
            for (int k0 = 0; k0 < NKB; k0 += PD) {
#pragma unroll
                for (int j = 0; j < PD; j++) {
                    const int ks = k0 + j;
                    const frag_t<T> a = opm_load_frag<T>(Adz, TM, li, 2 * ks + half);
#pragma unroll
                    for (int cb = 0; cb < 2; cb++)
#pragma unroll
                        for (int part = 0; part < 2; part++) mma32(acc2[cb][part], a, scan3_as_frag(bq[j][cb * 2 + part]));
                    int kn = ks + PD;
                    if (kn >= NKB) kn -= NKB;
                    issue(bq[j], kn);
                }
            }
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    *reinterpret_cast<T*>(Sx + acc_elem_off(off0[cb], r)) = (T)acc2[cb][0][r];
                    dh_rec[cb][r] = acc2[cb][1][r];
                }
            lds_barrier();                                 // dx tile complete; dz reads done; dH_{t-1} landed
            {
                T* const xdst = dx_all + (size_t)t * MC + (size_t)m0 * C;
                constexpr int GP = C / 8;
#pragma unroll
                for (int q = 0; q < TM * GP / NT; q++) {
                    const int f = tid + q * NT;
                    const int row = f / GP, cg = f % GP;
                    if (row < rows) frag_store<T>(xdst + (size_t)row * C + cg * 8, opm_load_frag<T>(Sx, TM, row, cg));
                }
            }
        }
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + acc_row(r, lane);
                if (row < M) {
                    dh0[(size_t)row * C + ch[cb]] = (T)dh_rec[cb][r];
                    dc0[(size_t)row * C + ch[cb]] = dc_rec[cb][r];
                }
            }
    }
}

// ---- packing (one launch per direction and optimizer step): natural weights -> operand order --------------------------------
// forward:  Wp[w][cb][ks][g][lane][e]  = W[g C + 64 w + 32 cb + (lane & 31)][16 ks + 8 (lane >> 5) + e],  W [4C][2C]
// backward: Wtp[w][ks][cb][part][lane][e] = W[16 ks + 8 (lane >> 5) + e][part C + 64 w + 32 cb + (lane & 31)]   (= W^T rows)
template <bool BWD>
__global__ void __launch_bounds__(256)
lstm_scan3_pack_kernel(const bf16* __restrict__ W, bf16* __restrict__ out, int C) {
    const int nk = BWD ? 4 * C / 16 : 2 * C / 16, per = BWD ? 4 : 8;
    const size_t total = (size_t)(C / 64) * nk * per * 64;           // 16-byte pieces
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const int lane = (int)(p & 63);
        size_t q = p >> 6;
        const int li = lane & 31, half = lane >> 5;
        bf16x8 v;
        if (!BWD) {
            const int g = (int)(q & 3); q >>= 2;
            const int ks = (int)(q % nk); q /= nk;
            const int cb = (int)(q & 1), w = (int)(q >> 1);
            v = *reinterpret_cast<const bf16x8*>(W + (size_t)(g * C + 64 * w + 32 * cb + li) * 2 * C + 16 * ks + 8 * half);
        } else {
            const int j = (int)(q % per); q /= per;
            const int ks = (int)(q % nk);
            const int w = (int)(q / nk);
            const int cb = j >> 1, part = j & 1;
            const int col = part * C + 64 * w + 32 * cb + li;
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = W[(size_t)(16 * ks + 8 * half + e) * 2 * C + col];
        }
        *reinterpret_cast<bf16x8*>(out + p * 8) = v;
    }
}

}  // namespace rvt
