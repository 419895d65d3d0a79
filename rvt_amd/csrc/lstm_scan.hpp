// ConvLSTM with the TIME LOOP INSIDE the kernel (reference models/layers/rnn.py:43-67 driven by the loop of
// modules/detection.py:131-148), forward and BPTT backward, for the 1x1-conv cell every shipped config uses
// (dws_conv: False, config/model/maxvit_yolox/default.yaml:39).
//
// With a 1x1 conv the recurrence is independent per pixel, so a workgroup owns a tile of TM pixels and walks t itself:
//   forward : h_{t-1} (next step's A operand) lives in LDS, c_t in fp32 accumulator-layout registers; per step the
//             kernel reads x_t once and writes h_t (the stage's output feature, needed anyway) and a T-typed copy of c_t
//             (the only thing BPTT needs that cannot be recomputed): 3 activation rows per token-step instead of 11,
//             one launch per stage instead of T.
//   backward: gates are RECOMPUTED from (x_t, h_{t-1}) with the same GEMM, dc and dh_rec stay in registers across the
//             reverse time loop, dz goes LDS -> (a) A operand of the dz.W product, whose W^T fragments come from the ONE
//             LDS image of W through the transpose read, and (b) HBM once, for the weight-gradient GEMM.
// Register layout trick: a wave computes, for ITS 32 channels, the four gate blocks f,i,o,g as four 32x32 MFMA column
// blocks (B rows g*C + c of the natural weight layout), so the four gates of a (token, channel) land in the SAME lane at
// the same accumulator index: the gate math needs no LDS staging and no shuffles, and c / dc / dh_rec persist in
// registers in that layout.
#pragma once
#include "common.hpp"
#include "mlp.hpp"

namespace rvt {

template <class T, int C, int NW> struct LstmScanGeom {
    static constexpr int NT = 64 * NW;
    static constexpr int NWC = C / 32;                     // channel groups of 32 = waves along the channel axis
    static constexpr int NWM = NW / NWC;                   // waves along the token axis
    static constexpr int BK = TileGeom<T>::BK;
    static constexpr int KTC = (C + BK - 1) / BK;          // K-subtiles of a [.][C] operand matrix
    static_assert(C % 32 == 0 && NW % NWC == 0 && NWM >= 1, "wave layout");
};

// predicated 16/32-byte load: never dereferences when !ok
template <class T> __device__ __forceinline__ frag_t<T> frag_load(const T* p, bool ok) {
    if (!ok) return frag_zero<T>();
    return *reinterpret_cast<const frag_t<T>*>(p);
}
// fp32 row segment -> T fragment (the incoming cell state of the backward scan)
template <class T> __device__ __forceinline__ frag_t<T> frag_load_f32(const float* p) {
    float v[8];
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    return frag_from_float<T>(v);
}

// cooperative copies between a global [M][C] matrix (rows m0.., zero beyond M) and an LDS operand matrix of TM rows,
// staged through registers so that the global loads can be issued a whole time step ahead of their use
template <class T, int C, int NT, int TM> __device__ __forceinline__ void lds_tile_from_regs(char* A, const frag_t<T>* r, int tid) {
    constexpr int G = C / 8, NFX = (TM * G + NT - 1) / NT;
#pragma unroll
    for (int q = 0; q < NFX; q++) {                          // (fully unrolled: r[] must stay in registers)
        const int f = tid + q * NT;
        if (f < TM * G) opm_store_frag<T>(A, TM, f / G, f % G, r[q]);
    }
}
template <class T, int C, int NT, int TM> __device__ __forceinline__ void tile_load_regs(frag_t<T>* r, const T* src, int m0, int M, int tid) {
    constexpr int G = C / 8, NFX = (TM * G + NT - 1) / NT;
#pragma unroll
    for (int q = 0; q < NFX; q++) {
        const int f = tid + q * NT;
        const int row = f / G;
        const bool ok = f < TM * G && m0 + row < M;
        if (src == nullptr) { r[q] = frag_zero<T>(); continue; }         // (workgroup-uniform)
        const frag_t<T> v = frag_load<T>(src + (ok ? (size_t)(m0 + row) * C + (f % G) * 8 : 0));   // branch-free: clamp, select
        const frag_t<T> z = frag_zero<T>();
        r[q] = ok ? v : z;
    }
}

// ====================================================================================================== forward
// x_all [Tn][M][C], Hall [Tn+1][M][C] (slot 0 = incoming h, filled by the caller; slots 1.. written here),
// c0 fp32 [M][C] or null (zeros), c_last fp32 [M][C], Csave [Tn][M][C] (slot t = c_t as T) or null,
// W [4C][2C] natural gate order f,i,o,g and input order [x|h] (rnn.py:52-61), bias fp32 [4C].
// W_REG (bf16, C = 128, one wave per SIMD): the weights do not fit the LDS (256 KB) but they do fit the REGISTER FILE — wave w
// needs the rows {g C + 32 w + lane} of all four gates, 64 operand pieces = 256 registers, loaded once per launch; streaming
// them from L2 every step instead (W_LDS = W_REG = false) is what bounded the C = 128 scan (15.5 GB of L2 reads per launch).
// gates_out (nullable): the activated gates [Tn][M][4C] (natural order f,i,o,g), for the reverse scan that does not recompute them.
template <class T, int C, int NW, int RB, bool W_LDS, bool W_REG = false>
__global__ void __launch_bounds__(64 * NW)
lstm_scan_fwd_kernel(const T* __restrict__ x_all, T* __restrict__ Hall, const float* __restrict__ c0, float* __restrict__ c_last,
                     T* __restrict__ Csave, const T* __restrict__ W, const float* __restrict__ bias, T* __restrict__ gates_out,
                     int M, int Tn) {
    typedef LstmScanGeom<T, C, NW> Gm;
    constexpr int NT = Gm::NT, NWC = Gm::NWC, NWM = Gm::NWM, KTC = Gm::KTC;
    constexpr int TM = NWM * RB * 32;
    constexpr int TILE = KTC * TM * 128;
    constexpr int WPART = KTC * (4 * C) * 128;             // one [4C][C] half of W
    constexpr int NFX = (TM * (C / 8) + NT - 1) / NT;
    constexpr int GT = W_REG ? 4 * TILE : 0;               // gate staging [TM][4C] (four [TM][C] tiles)
    static_assert(!(W_LDS && W_REG), "one home for the weights");
    __shared__ __attribute__((aligned(16))) char smem[(W_LDS ? 2 * WPART : 0) + 4 * TILE + GT];
    char* const Wx = smem;
    char* const Wh = smem + (W_LDS ? WPART : 0);
    char* const Ax = smem + (W_LDS ? 2 * WPART : 0);
    char* const Ah0 = Ax + TILE;
    char* const Cs = Ax + 3 * TILE;
    char* const Gs = Ax + 4 * TILE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int wn = wave % NWC, wm = wave / NWC;
    const int ch = wn * 32 + li;                           // this lane's channel
    const size_t MC = (size_t)M * C;

    if (W_LDS) {                                           // W -> LDS once per workgroup: [4C][x part] and [4C][h part]
        constexpr int G = C / 8;
        for (int f = tid; f < 4 * C * 2 * G; f += NT) {
            const int n = f / (2 * G), g8 = f % (2 * G);
            const frag_t<T> v = frag_load<T>(W + (size_t)n * 2 * C + g8 * 8);
            if (g8 < G) opm_store_frag<T>(Wx, 4 * C, n, g8, v);
            else opm_store_frag<T>(Wh, 4 * C, n, g8 - G, v);
        }
    }
    frag_t<T> wreg[W_REG ? 2 : 1][W_REG ? C / 16 : 1][W_REG ? 4 : 1];
    if (W_REG) {
#pragma unroll
        for (int part = 0; part < 2; part++)
#pragma unroll
            for (int k16 = 0; k16 < C / 16; k16++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    wreg[W_REG ? part : 0][W_REG ? k16 : 0][W_REG ? g : 0] =
                        frag_load<T>(W + (size_t)(g * C + wn * 32 + (lane & 31)) * 2 * C + part * C + (2 * k16 + (lane >> 5)) * 8);
    }
    // gate biases folded into the exp2 arguments; LDS byte offsets of this lane's 16 (row, channel) elements per row block
    // (fixed for the launch: computing the swizzle per element per step costs more VALU than the gate math itself)
    const float nbf = bias[ch] * -1.4426950408889634f, nbi = bias[C + ch] * -1.4426950408889634f;
    const float nbo = bias[2 * C + ch] * -1.4426950408889634f, tbg = bias[3 * C + ch] * 2.8853900817779268f;
    int off0[RB];
#pragma unroll
    for (int i = 0; i < RB; i++)
        off0[i] = (int)(reinterpret_cast<char*>(opm_elem_ptr<T>(Ax, TM, (wm * RB + i) * 32 + 4 * half, ch)) - Ax);

    const int n_tiles = (M + TM - 1) / TM;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m0 = tile * TM;
        frag_t<T> xr[NFX];
        float creg[RB][16];
        // prologue of the tile: h_{-1} -> Ah[0], x_0 -> Ax, c_{-1} -> registers (accumulator layout)
        tile_load_regs<T, C, NT, TM>(xr, Hall, m0, M, tid);
        lds_barrier();                                     // previous tile's copy-out / MFMA reads are done
        lds_tile_from_regs<T, C, NT, TM>(Ah0, xr, tid);
        tile_load_regs<T, C, NT, TM>(xr, x_all, m0, M, tid);
        lds_tile_from_regs<T, C, NT, TM>(Ax, xr, tid);
#pragma unroll
        for (int i = 0; i < RB; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + (wm * RB + i) * 32 + acc_row(r, lane);
                creg[i][r] = (c0 != nullptr && row < M) ? c0[(size_t)row * C + ch] : 0.f;
            }
        lds_barrier();
        int cur = 0;
        for (int t = 0; t < Tn; t++) {
            char* const Ah = Ah0 + cur * TILE;
            char* const An = Ah0 + (cur ^ 1) * TILE;
            if (t + 1 < Tn) tile_load_regs<T, C, NT, TM>(xr, x_all + (size_t)(t + 1) * MC, m0, M, tid);
            sched_fence();
            f32x16 acc[RB][4];
#pragma unroll
            for (int i = 0; i < RB; i++)
#pragma unroll
                for (int g = 0; g < 4; g++) acc_zero(acc[i][g]);
            // (weights streamed from L2: keep the K loop rolled, or every B fragment of the step is hoisted and spilled)
            constexpr int KUNROLL = (W_LDS || W_REG) ? C / 16 : 1;
#pragma unroll
            for (int part = 0; part < 2; part++) {         // [x_t | h_{t-1}] . W^T, K order = x columns then h columns
                const char* const A = part ? Ah : Ax;
                const char* const Wl = part ? Wh : Wx;
#pragma clang loop unroll_count(KUNROLL)
                for (int kc = 0; kc < C; kc += 16) {
                    const int fcg = kc / 8 + half;
                    frag_t<T> a[RB], b[4];
#pragma unroll
                    for (int i = 0; i < RB; i++) a[i] = opm_load_frag<T>(A, TM, (wm * RB + i) * 32 + li, fcg);
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        if (W_LDS) b[g] = opm_load_frag<T>(Wl, 4 * C, g * C + ch, fcg);
                        else if (W_REG) b[g] = wreg[W_REG ? part : 0][W_REG ? kc / 16 : 0][W_REG ? g : 0];
                        else b[g] = frag_load<T>(W + (size_t)(g * C + ch) * 2 * C + part * C + fcg * 8);
                    }
#pragma unroll
                    for (int i = 0; i < RB; i++)
#pragma unroll
                        for (int g = 0; g < 4; g++) mma32(acc[i][g], a[i], b[g]);
                }
            }
            // gates (rnn.py:57-67), in registers
#pragma unroll
            for (int i = 0; i < RB; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float f = sigmoid_zb(acc[i][0][r], nbf);
                    const float ig = sigmoid_zb(acc[i][1][r], nbi);
                    const float o = sigmoid_zb(acc[i][2][r], nbo);
                    const float g = tanh_zb(acc[i][3][r], tbg);
                    const float cn = f * creg[i][r] + ig * g;
                    creg[i][r] = cn;
                    const float hn = o * tanh_f(cn);
                    *reinterpret_cast<T*>(An + acc_elem_off(off0[i], r)) = (T)hn;
                    if (Csave != nullptr) *reinterpret_cast<T*>(Cs + acc_elem_off(off0[i], r)) = (T)cn;
                    if (W_REG && gates_out != nullptr) {
                        *reinterpret_cast<T*>(Gs + 0 * TILE + acc_elem_off(off0[i], r)) = (T)f;
                        *reinterpret_cast<T*>(Gs + 1 * TILE + acc_elem_off(off0[i], r)) = (T)ig;
                        *reinterpret_cast<T*>(Gs + 2 * TILE + acc_elem_off(off0[i], r)) = (T)o;
                        *reinterpret_cast<T*>(Gs + 3 * TILE + acc_elem_off(off0[i], r)) = (T)g;
                    }
                }
            lds_barrier();                                 // h_t / c_t tiles complete; every wave is done with Ax
            {                                              // tile rows -> HBM in 16-byte pieces
                constexpr int G = C / 8;
                T* const hdst = Hall + (size_t)(t + 1) * MC;
                T* const cdst = Csave != nullptr ? Csave + (size_t)t * MC : nullptr;
                for (int f = tid; f < TM * G; f += NT) {
                    const int row = f / G, cg = f % G;
                    if (m0 + row < M) {
                        frag_store<T>(hdst + (size_t)(m0 + row) * C + cg * 8, opm_load_frag<T>(An, TM, row, cg));
                        if (cdst != nullptr) frag_store<T>(cdst + (size_t)(m0 + row) * C + cg * 8, opm_load_frag<T>(Cs, TM, row, cg));
                    }
                }
                if (W_REG && gates_out != nullptr) {
                    T* const gdst = gates_out + (size_t)t * MC * 4;
                    for (int f = tid; f < TM * 4 * G; f += NT) {
                        const int row = f / (4 * G), gg = (f / G) % 4, cg = f % G;
                        if (m0 + row < M)
                            frag_store<T>(gdst + (size_t)(m0 + row) * 4 * C + gg * C + cg * 8, opm_load_frag<T>(Gs + gg * TILE, TM, row, cg));
                    }
                }
            }
            if (t + 1 < Tn) lds_tile_from_regs<T, C, NT, TM>(Ax, xr, tid);
            lds_barrier();
            cur ^= 1;
        }
#pragma unroll
        for (int i = 0; i < RB; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + (wm * RB + i) * 32 + acc_row(r, lane);
                if (row < M) c_last[(size_t)row * C + ch] = creg[i][r];
            }
    }
}

// ===================================================================================================== backward
// Reverse scan.  Inputs as the forward's plus dH [Tn][M][C] (cotangent of Hall[1..], null = zeros) and dc_last fp32 [M][C]
// (null = zeros); Wt = W^T [2C][4C] (only read when !W_LDS).  Outputs: dx_all [Tn][M][C], dz_all [Tn][M][4C] (natural gate
// order, for the weight-gradient GEMM), dh0 [M][C] (T), dc0 fp32 [M][C].
// WGRAD: the weight gradients dW[4C][2C] += dz^T [x | h_prev] and db[4C] += colsum(dz) are accumulated IN REGISTERS across
// the whole launch (the dz tile and the [x | h] tiles are in LDS anyway; their token-contraction operands come out through the
// transposing LDS read) and leave as one partial per workgroup in `ws` ([grid][4C x 2C | 4C] floats): dz never goes to HBM
// (4 rows of C per token-step written + re-read by the weight-gradient GEMM otherwise).  Wave w owns dW rows 64 w .. 64 w + 63.
// GATES (bf16, C = 128, one wave per SIMD): the activated gates saved by the W_REG forward are READ instead of recomputed, so
// the only weights this kernel needs are the W^T pieces of its own 32 x- and 32 h-columns — 64 operand pieces = 256 registers,
// loaded once per launch (x_all / Hall are then not read at all).
template <class T, int C, int NW, bool W_LDS, bool WGRAD, bool GATES = false>
__global__ void __launch_bounds__(64 * NW)
lstm_scan_bwd_kernel(const T* __restrict__ x_all, const T* __restrict__ Hall, const T* __restrict__ Csave,
                     const float* __restrict__ c0, const T* __restrict__ dH, const float* __restrict__ dc_last,
                     const T* __restrict__ W, const T* __restrict__ Wt, const float* __restrict__ bias,
                     T* __restrict__ dx_all, T* __restrict__ dz_all, T* __restrict__ dh0, float* __restrict__ dc0,
                     float* __restrict__ ws, const T* __restrict__ gates, int M, int Tn) {
    typedef LstmScanGeom<T, C, NW> Gm;
    constexpr int NT = Gm::NT, NWC = Gm::NWC, NWM = Gm::NWM, KTC = Gm::KTC, BK = Gm::BK;
    constexpr int TM = NWM * 32;
    constexpr int TILE = KTC * TM * 128;
    constexpr int WPART = KTC * (4 * C) * 128;
    constexpr int KT4 = (4 * C) / BK;
    static_assert((4 * C) % BK == 0, "dz operand");
    constexpr int DZ = KT4 * TM * 128;
    constexpr int G = C / 8;
    constexpr int NFX = (TM * G + NT - 1) / NT;
    static_assert(!GATES || (!W_LDS && !WGRAD), "the gate-reading variant keeps W^T in registers and leaves dz to the weight-gradient GEMM");
    __shared__ __attribute__((aligned(16))) char smem[(W_LDS ? 2 * WPART : 0) + 5 * TILE + DZ + (GATES ? 4 * TILE : 0)];
    char* const Wx = smem;
    char* const Wh = smem + (W_LDS ? WPART : 0);
    char* const Ax = smem + (W_LDS ? 2 * WPART : 0);
    char* const Ah = Ax + TILE;
    char* const Sc = Ax + 2 * TILE;
    char* const Sd = Ax + 3 * TILE;
    char* const Sx = Ax + 4 * TILE;
    char* const Adz = Ax + 5 * TILE;
    char* const Sg = Adz + DZ;                             // saved gates of the step: four [TM][C] tiles (f, i, o, g)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, half = lane >> 5;
    const int wn = wave % NWC, wm = wave / NWC;
    const int ch = wn * 32 + li;
    const size_t MC = (size_t)M * C;

    if (W_LDS) {
        for (int f = tid; f < 4 * C * 2 * G; f += NT) {
            const int n = f / (2 * G), g8 = f % (2 * G);
            const frag_t<T> v = frag_load<T>(W + (size_t)n * 2 * C + g8 * 8);
            if (g8 < G) opm_store_frag<T>(Wx, 4 * C, n, g8, v);
            else opm_store_frag<T>(Wh, 4 * C, n, g8 - G, v);
        }
    }
    const float nbf = bias[ch] * -1.4426950408889634f, nbi = bias[C + ch] * -1.4426950408889634f;
    const float nbo = bias[2 * C + ch] * -1.4426950408889634f, tbg = bias[3 * C + ch] * 2.8853900817779268f;
    // LDS byte offsets (fixed for the launch) of this lane's 16 (row, channel) elements in a [TM][C] tile, and the map from
    // there to element (row, g*C + channel) of the [TM][4C] dz operand
    const int off0 = (int)(reinterpret_cast<char*>(opm_elem_ptr<T>(Ax, TM, wm * 32 + 4 * half, ch)) - Ax);
    auto dz_off = [&](int g, int r) -> int {
        if constexpr (C % BK == 0) return g * KTC * TM * 128 + acc_elem_off(off0, r);    // gate g = K-subtiles g*KTC ..
        else return (g / (BK / C)) * TM * 128 + acc_elem_off(off0 ^ ((g % (BK / C)) * C * (int)sizeof(T)), r);   // several gates per subtile
    };
    // transpose-read addresses of the W^T fragments (rows n = kc + tl (+4), columns j = wn*32 + fl): everything but the
    // k-step is per-lane constant; the swizzle term of the k-step is (kc >> 4) & 7
    const int tr_tl = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int tr_col = wn * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int tr_sub = (tr_col / BK) * (4 * C) * 128;
    const int tr_byte = (tr_col % BK) * (int)sizeof(T);
    const int tr_row_lo = tr_sub + tr_tl * 128 + (tr_byte & 15), tr_row_hi = tr_row_lo + 4 * 128;
    const int tr_c_lo = (tr_byte >> 4) ^ ((tr_tl >> 1) & 7), tr_c_hi = (tr_byte >> 4) ^ (((tr_tl + 4) >> 1) & 7);

    // c_{t-1} as a T fragment tile: slot t-1 of Csave, or the incoming fp32 state for t = 0
    auto load_cprev = [&](frag_t<T> (&r)[NFX], int t, int m0) {
#pragma unroll
        for (int q = 0; q < NFX; q++) {
            const int f = tid + q * NT;
            const int row = f / G;
            const bool ok = f < TM * G && m0 + row < M;
            const size_t o = ok ? (size_t)(m0 + row) * C + (f % G) * 8 : 0;
            if (t > 0) r[q] = frag_load<T>(Csave + (size_t)(t - 1) * MC + o, ok);
            else if (c0 != nullptr && ok) r[q] = frag_load_f32<T>(c0 + o);
            else r[q] = frag_zero<T>();
        }
    };

    // weight-gradient accumulators: dW row blocks (4C / 32) / NW per wave x all 2C / 32 column blocks
    constexpr int WG_RB = WGRAD ? (4 * C / 32) / NW : 1, WG_CB = WGRAD ? 2 * C / 32 : 1;
    static_assert(!WGRAD || ((4 * C / 32) % NW == 0 && TM % 16 == 0), "weight-gradient tiling");
    f32x16 dwacc[WG_RB][WG_CB];
    float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
    TrFeat<T> tr_dz[WG_RB], tr_xh[WG_CB];
    if (WGRAD) {
#pragma unroll
        for (int i = 0; i < WG_RB; i++) {
            tr_dz[i].init((wave * WG_RB + i) * 32, TM, lane);
#pragma unroll
            for (int j = 0; j < WG_CB; j++) acc_zero(dwacc[i][j]);
        }
#pragma unroll
        for (int j = 0; j < WG_CB; j++) tr_xh[j].init((j * 32) % C, TM, lane);
    }

    frag_t<T> wtreg[GATES ? 2 : 1][GATES ? 4 * C / 16 : 1];
    if (GATES) {
#pragma unroll
        for (int part = 0; part < 2; part++)
#pragma unroll
            for (int k16 = 0; k16 < 4 * C / 16; k16++)
                wtreg[GATES ? part : 0][GATES ? k16 : 0] = frag_load<T>(Wt + (size_t)(part * C + ch) * 4 * C + 16 * k16 + half * 8);
    }
    // gates of step t, staged like the other tiles: rows [4C] of the saved tensor -> four [TM][C] tiles
    auto load_gates = [&](frag_t<T> (&r)[4 * NFX], int t, int m0) {
#pragma unroll
        for (int q = 0; q < 4 * NFX; q++) {
            const int f = tid + q * NT;
            const int row = f / (4 * G);
            const bool ok = f < TM * 4 * G && m0 + row < M;
            r[q] = frag_load<T>(gates + (size_t)t * MC * 4 + (ok ? (size_t)(m0 + row) * 4 * C + (f % (4 * G)) * 8 : 0), ok);
        }
    };
    auto store_gates = [&](const frag_t<T> (&r)[4 * NFX]) {
#pragma unroll
        for (int q = 0; q < 4 * NFX; q++) {
            const int f = tid + q * NT;
            if (f < TM * 4 * G) {
                const int row = f / (4 * G), gg = (f / G) % 4, cg = f % G;
                opm_store_frag<T>(Sg + gg * TILE, TM, row, cg, r[q]);
            }
        }
    };

    const int n_tiles = (M + TM - 1) / TM;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m0 = tile * TM;
        frag_t<T> rx[NFX], rh[NFX], rc[NFX], rd[NFX];
        frag_t<T> rg[4 * NFX];                            // (only touched by the GATES variant)
        float dh_rec[16], dc_rec[16];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + acc_row(r, lane);
            dh_rec[r] = 0.f;
            dc_rec[r] = (dc_last != nullptr && row < M) ? dc_last[(size_t)row * C + ch] : 0.f;
        }
        {
            const int t = Tn - 1;
            if (!GATES) {
                tile_load_regs<T, C, NT, TM>(rx, x_all + (size_t)t * MC, m0, M, tid);
                tile_load_regs<T, C, NT, TM>(rh, Hall + (size_t)t * MC, m0, M, tid);
            } else {
                load_gates(rg, t, m0);
            }
            load_cprev(rc, t, m0);
            tile_load_regs<T, C, NT, TM>(rd, dH != nullptr ? dH + (size_t)t * MC : nullptr, m0, M, tid);
        }
        lds_barrier();                                     // previous tile fully consumed
        if (!GATES) {
            lds_tile_from_regs<T, C, NT, TM>(Ax, rx, tid);
            lds_tile_from_regs<T, C, NT, TM>(Ah, rh, tid);
        } else {
            store_gates(rg);
        }
        lds_tile_from_regs<T, C, NT, TM>(Sc, rc, tid);
        lds_tile_from_regs<T, C, NT, TM>(Sd, rd, tid);
        for (int t = Tn - 1; t >= 0; t--) {
            lds_barrier();                                 // tiles of step t are in LDS
            if (t > 0) {
                if (!GATES) {
                    tile_load_regs<T, C, NT, TM>(rx, x_all + (size_t)(t - 1) * MC, m0, M, tid);
                    tile_load_regs<T, C, NT, TM>(rh, Hall + (size_t)(t - 1) * MC, m0, M, tid);
                } else {
                    load_gates(rg, t - 1, m0);
                }
                load_cprev(rc, t - 1, m0);
                tile_load_regs<T, C, NT, TM>(rd, dH != nullptr ? dH + (size_t)(t - 1) * MC : nullptr, m0, M, tid);
            }
            sched_fence();
            // ---- recompute the pre-activations: z = [x_t | h_{t-1}] W^T   (GATES: read the saved gates instead) ----
            f32x16 acc[4];
#pragma unroll
            for (int g = 0; g < 4; g++) acc_zero(acc[g]);
            constexpr int KUNROLL = W_LDS ? C / 16 : 1;
#pragma unroll
            for (int part = 0; part < (GATES ? 0 : 2); part++) {
                const char* const A = part ? Ah : Ax;
                const char* const Wl = part ? Wh : Wx;
#pragma clang loop unroll_count(KUNROLL)
                for (int kc = 0; kc < C; kc += 16) {
                    const int fcg = kc / 8 + half;
                    const frag_t<T> a = opm_load_frag<T>(A, TM, wm * 32 + li, fcg);
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        frag_t<T> b;
                        if (W_LDS) b = opm_load_frag<T>(Wl, 4 * C, g * C + ch, fcg);
                        else b = frag_load<T>(W + (size_t)(g * C + ch) * 2 * C + part * C + fcg * 8);
                        mma32(acc[g], a, b);
                    }
                }
            }
            // ---- gate backward (autograd of rnn.py:57-67) in registers; dz -> LDS as the next product's A operand ----
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float f, ig, o, g;
                if (GATES) {
                    f = (float)*reinterpret_cast<const T*>(Sg + 0 * TILE + acc_elem_off(off0, r));
                    ig = (float)*reinterpret_cast<const T*>(Sg + 1 * TILE + acc_elem_off(off0, r));
                    o = (float)*reinterpret_cast<const T*>(Sg + 2 * TILE + acc_elem_off(off0, r));
                    g = (float)*reinterpret_cast<const T*>(Sg + 3 * TILE + acc_elem_off(off0, r));
                } else {
                    f = sigmoid_zb(acc[0][r], nbf);
                    ig = sigmoid_zb(acc[1][r], nbi);
                    o = sigmoid_zb(acc[2][r], nbo);
                    g = tanh_zb(acc[3][r], tbg);
                }
                const float cp = (float)*reinterpret_cast<const T*>(Sc + acc_elem_off(off0, r));
                const float dh = (float)*reinterpret_cast<const T*>(Sd + acc_elem_off(off0, r)) + dh_rec[r];
                const float tc = tanh_f(f * cp + ig * g);
                const float dc = dc_rec[r] + dh * o * (1.f - tc * tc);
                const float zf = dc * cp * f * (1.f - f), zi = dc * g * ig * (1.f - ig);
                const float zo = dh * tc * o * (1.f - o), zg = dc * ig * (1.f - g * g);
                *reinterpret_cast<T*>(Adz + dz_off(0, r)) = (T)zf;
                *reinterpret_cast<T*>(Adz + dz_off(1, r)) = (T)zi;
                *reinterpret_cast<T*>(Adz + dz_off(2, r)) = (T)zo;
                *reinterpret_cast<T*>(Adz + dz_off(3, r)) = (T)zg;
                if (WGRAD) { dbacc[0] += zf; dbacc[1] += zi; dbacc[2] += zo; dbacc[3] += zg; }   // (rows beyond M: dh = dc = 0)
                dc_rec[r] = dc * f;
            }
            lds_barrier();                                 // dz tile complete; Ax/Ah/Sc/Sd of step t consumed
            // ---- [dx_t | dh_{t-1}] = dz W: this wave's 32 x-columns and its 32 h-columns ----
            f32x16 acc2[2];
            acc_zero(acc2[0]); acc_zero(acc2[1]);
            constexpr int KUNROLL2 = GATES ? 4 * C / 16 : (W_LDS ? 4 : 1);
#pragma clang loop unroll_count(KUNROLL2)
            for (int kc = 0; kc < 4 * C; kc += 16) {
                const frag_t<T> a = opm_load_frag<T>(Adz, TM, wm * 32 + li, kc / 8 + half);
#pragma unroll
                for (int part = 0; part < 2; part++) {
                    frag_t<T> b;
                    if (GATES) {
                        b = wtreg[GATES ? part : 0][GATES ? kc / 16 : 0];
                    } else if (W_LDS) {       // B[j][n] = W[n][j]: transposed fragments of the LDS image of W (rows n, columns j)
                        const char* const Wl = part ? Wh : Wx;
                        if constexpr (sizeof(T) == 2) {
                            const int u = (kc >> 4) & 7;
                            b = frag_from_tr<T>(reinterpret_cast<const bf16*>(Wl + kc * 128 + tr_row_lo + ((tr_c_lo ^ u) << 4)),
                                                reinterpret_cast<const bf16*>(Wl + kc * 128 + tr_row_hi + ((tr_c_hi ^ u) << 4)));
                        } else {
                            auto at = [&](int n, int j) -> const T* { return opm_elem_ptr<T>(const_cast<char*>(Wl), 4 * C, n, j); };
                            b = load_frag_tr<T>(at, kc, wn * 32, lane);
                        }
                    } else {
                        b = frag_load<T>(Wt + (size_t)(part * C + ch) * 4 * C + kc + half * 8);
                    }
                    mma32(acc2[part], a, b);
                }
            }
            if (WGRAD) {
                // dW[n][k] += sum_tok dz[tok][n] [x | h][tok][k]: both operands transposed on the way out of LDS
#pragma unroll
                for (int k0 = 0; k0 < TM; k0 += 16) {
                    frag_t<T> a[WG_RB], b[WG_CB];
#pragma unroll
                    for (int i = 0; i < WG_RB; i++) a[i] = tr_dz[i].load(Adz, k0, lane);
#pragma unroll
                    for (int j = 0; j < WG_CB; j++) b[j] = tr_xh[j].load(j * 32 < C ? Ax : Ah, k0, lane);
#pragma unroll
                    for (int i = 0; i < WG_RB; i++)
#pragma unroll
                        for (int j = 0; j < WG_CB; j++) mma32(dwacc[i][j], a[i], b[j]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                *reinterpret_cast<T*>(Sx + acc_elem_off(off0, r)) = (T)acc2[0][r];
                dh_rec[r] = acc2[1][r];
            }
            lds_barrier();                                 // dx tile complete; dz reads done
            {
                T* const xdst = dx_all + (size_t)t * MC;
                for (int f = tid; f < TM * G; f += NT) {
                    const int row = f / G, cg = f % G;
                    if (m0 + row < M) frag_store<T>(xdst + (size_t)(m0 + row) * C + cg * 8, opm_load_frag<T>(Sx, TM, row, cg));
                }
                if (!WGRAD) {
                    T* const zdst = dz_all + (size_t)t * MC * 4;
                    for (int f = tid; f < TM * 4 * G; f += NT) {
                        const int row = f / (4 * G), cg = f % (4 * G);
                        if (m0 + row < M) frag_store<T>(zdst + (size_t)(m0 + row) * 4 * C + cg * 8, opm_load_frag<T>(Adz, TM, row, cg));
                    }
                }
            }
            if (t > 0) {
                if (!GATES) {
                    lds_tile_from_regs<T, C, NT, TM>(Ax, rx, tid);
                    lds_tile_from_regs<T, C, NT, TM>(Ah, rh, tid);
                } else {
                    store_gates(rg);
                }
                lds_tile_from_regs<T, C, NT, TM>(Sc, rc, tid);
                lds_tile_from_regs<T, C, NT, TM>(Sd, rd, tid);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + acc_row(r, lane);
            if (row < M) {
                dh0[(size_t)row * C + ch] = (T)dh_rec[r];
                dc0[(size_t)row * C + ch] = dc_rec[r];
            }
        }
    }
    if (WGRAD) {
        // per-workgroup partials: [4C][2C] weight gradient, then [NWM][4C] bias-gradient rows (one per token half of the tile)
        float* const p_dw = ws + (size_t)blockIdx.x * (4 * C * 2 * C + NWM * 4 * C);
        float* const p_db = p_dw + 4 * C * 2 * C + wm * 4 * C;
#pragma unroll
        for (int i = 0; i < WG_RB; i++)
#pragma unroll
            for (int j = 0; j < WG_CB; j++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    p_dw[(size_t)((wave * WG_RB + i) * 32 + acc_row(r, lane)) * (2 * C) + j * 32 + li] = dwacc[i][j][r];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float v = dbacc[g] + __shfl_xor(dbacc[g], 32);
            if (half == 0) p_db[g * C + ch] = v;
        }
    }
}

}  // namespace rvt
