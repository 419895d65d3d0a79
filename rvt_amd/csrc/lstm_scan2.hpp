// ConvLSTM with the time loop inside the kernel, round-4 rebuild for the LDS-resident width (bf16, C = 64; reference
// models/layers/rnn.py:43-67 driven by the loop of modules/detection.py:131-148, and its BPTT).
//
// lstm_scan.hpp computes the gate pre-activations in "N-form" (lane = channel, registers = 16 token rows): every tile that
// enters or leaves the registers (x, h, c, dH, dz, dx) goes through LDS in 2-byte pieces behind workgroup barriers, and the ISA
// audit of round 3 counted 1629 VALU + 237 LDS instructions + 4 barriers per step for 72 MFMAs at one wave per SIMD.  Here the
// products are in "T-form" (A = weight rows, B = token rows: lane = TOKEN, registers = channels), the same accumulator-to-operand
// chaining as attn_block.hpp / mlp_chain.hpp:
//   * x_t (and, in the reverse scan, h_{t-1}) rows are read from HBM directly in MFMA-operand form (lane = row, 16 B per k-step);
//   * h_t leaves the gate math in accumulator layout, which IS the B operand of the next step's product (the h-columns of W are
//     staged in accumulator order once per workgroup): the forward recurrence never touches LDS and its waves never synchronise
//     (one wave = 32 tokens x all channels, eight waves per workgroup = two per SIMD);
//   * rows leave as 16-byte pieces (bf16 pairs exchanged between the wave halves with v_permlane32_swap);
//   * the reverse scan keeps dz / [x | h] tiles in LDS only for the token-contraction of the in-kernel weight gradient (8- and
//     16-byte writes, transposing reads), two barriers per step.
// What bounds these kernels afterwards is the gate math itself: 10 quarter-rate transcendentals per (token, channel) and step.
#pragma once
#include "common.hpp"
#include "mlp.hpp"
#include "attn_block.hpp"
#include <utility>

namespace rvt {

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

// a pointer that is the same in every lane, pinned into scalar registers: accesses `p[lane_offset32 + const]` then take the
// scalar-base + 32-bit-lane-offset form.  Without it loop strength reduction turns every `base + t * stride + row` of the time
// loop into its own 64-bit per-lane induction pointer (the reverse scan spilled 25 of them per step).
template <class P> __device__ __forceinline__ P* uniform_ptr(P* p) {
#ifdef RVT_EMU
    return p;
#else
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    // (through an address_space(1) pointer: after the integer round trip the compiler would otherwise fall back to FLAT accesses)
    return (P*)(__attribute__((address_space(1))) P*)(((unsigned long long)hi << 32) | lo);
#endif
}

__device__ __forceinline__ void swap32u(unsigned& a, unsigned& b) {
    float fa = __builtin_bit_cast(float, a), fb = __builtin_bit_cast(float, b);
    swap32(fa, fb);
    a = __builtin_bit_cast(unsigned, fa);
    b = __builtin_bit_cast(unsigned, fb);
}
// accumulator-layout values (16 floats of one 32-channel block, this lane's token) -> the two operand fragments of the block
// (fragment q = registers 8q..8q+7 = channels 16q + {0..3, 8..11} + 4 half) ...
__device__ __forceinline__ void s2_pack(const float (&v)[16], frag_t<bf16> (&f)[2]) {
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int e = 0; e < 8; e++) f[q][e] = (bf16)v[8 * q + e];
}
// ... and from those the two 16-byte ROW pieces of this lane: piece m = channels 16 m + 8 half .. + 7 of the block
__device__ __forceinline__ void s2_frags_to_rows(const frag_t<bf16> (&f)[2], u32x4 (&piece)[2]) {
#pragma unroll
    for (int m = 0; m < 2; m++) {
        const u32x4 d = __builtin_bit_cast(u32x4, f[m]);
        unsigned a0 = d[0], b0 = d[2], a1 = d[1], b1 = d[3];
        swap32u(a0, b0);
        swap32u(a1, b1);
        piece[m][0] = a0; piece[m][1] = a1; piece[m][2] = b0; piece[m][3] = b1;
    }
}
// a bf16 row [C] of HBM -> operand fragment q of block cb in ACCUMULATOR order (two 8-byte loads)
__device__ __forceinline__ frag_t<bf16> s2_load_acc_frag(const bf16* row, int cb, int q, int half) {
    const bf16* p = row + 32 * cb + 16 * q + 4 * half;
    const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(p), hi = *reinterpret_cast<const u32x2_t*>(p + 8);
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(frag_t<bf16>, v);
}
__device__ __forceinline__ void s2_frags_to_float(const frag_t<bf16> (&f)[2], float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
        for (int e = 0; e < 8; e++) v[8 * q + e] = (float)f[q][e];
}
// fp32 row -> accumulator-layout registers of block cb (registers 4j..4j+3 = channels 8j + 4 half + 0..3: one 16-byte load each)
__device__ __forceinline__ void s2_load_acc_f32(const float* row, int cb, int half, float (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(row + 32 * cb + 8 * j + 4 * half);
#pragma unroll
        for (int w = 0; w < 4; w++) v[4 * j + w] = t[w];
    }
}
__device__ __forceinline__ void s2_store_acc_f32(float* row, int cb, int half, const float (&v)[16]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        f32x4 t;
#pragma unroll
        for (int w = 0; w < 4; w++) t[w] = v[4 * j + w];
        *reinterpret_cast<f32x4*>(row + 32 * cb + 8 * j + 4 * half) = t;
    }
}

struct Scan2Smem {
    static constexpr int C = 64;
    static constexpr int W_PART = 4 * C * 128;            // [4C rows][C] bf16 as one operand sub-tile column: 32 KB
    static constexpr int OFF_WX = 0, OFF_WH = W_PART, OFF_BIAS = 2 * W_PART, FWD_BYTES = 2 * W_PART + 4 * C * 4;
};

// W [4C][2C] (natural order) -> LDS: Wx = columns 0..C-1 in natural order (B operand = rows read from HBM), Wh = columns C..2C-1 with
// every 32-column block in ACCUMULATOR order (B operand = an accumulator block of h): fragment (blk, q, half) holds the channels
// 32 blk + 16 q + 4 half + {0..3, 8..11}
__device__ __forceinline__ void scan2_stage_w(char* Wx, char* Wh, const bf16* __restrict__ W, int tid, int nthreads) {
    constexpr int C = Scan2Smem::C;
    for (int f = tid; f < 4 * C * 16; f += nthreads) {
        const int n = f >> 4, g8 = f & 15;
        if (g8 < 8) {
            opm_store_frag<bf16>(Wx, 4 * C, n, g8, frag_load<bf16>(W + (size_t)n * 2 * C + g8 * 8));
        } else {
            const int fcg = g8 - 8, blk = fcg >> 2, q = (fcg >> 1) & 1, half = fcg & 1;
            const bf16* p = W + (size_t)n * 2 * C + C + blk * 32 + 16 * q + 4 * half;
            frag_t<bf16> v;
#pragma unroll
            for (int e = 0; e < 4; e++) { v[e] = p[e]; v[4 + e] = p[8 + e]; }
            opm_store_frag<bf16>(Wh, 4 * C, n, fcg, v);
        }
    }
}

// ====================================================================================================== forward
// x_all [Tn][M][C], Hall [Tn+1][M][C] (slot 0 = incoming h, caller-filled; slots 1.. written here), c0 fp32 [M][C] or null
// (zeros), c_last fp32 [M][C], Csave [Tn][M][C] (slot t = c_t) or null, W [4C][2C] natural gate order f,i,o,g and input order
// [x | h] (rnn.py:52-61), bias fp32 [4C].  One wave = 32 tokens; waves are independent.
template <int WPB>
__global__ void __launch_bounds__(64 * WPB)
lstm_scan2_fwd_kernel(const bf16* __restrict__ x_all, bf16* __restrict__ Hall, const float* __restrict__ c0, float* __restrict__ c_last,
                      bf16* __restrict__ Csave, const bf16* __restrict__ W, const float* __restrict__ bias, int M, int Tn) {
    typedef bf16 T;
    typedef Scan2Smem S;
    constexpr int C = S::C, KS = C / 16, NCB = C / 32;
    __shared__ __attribute__((aligned(16))) char smem[S::FWD_BYTES];
    char* const Wx = smem + S::OFF_WX;
    char* const Wh = smem + S::OFF_WH;
    float* const kb = reinterpret_cast<float*>(smem + S::OFF_BIAS);
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, half = lane >> 5, wave = tid >> 6;
    scan2_stage_w(Wx, Wh, W, tid, 64 * WPB);
    for (int i = tid; i < 4 * C; i += 64 * WPB) kb[i] = bias[i];
    __syncthreads();
    const size_t MC = (size_t)M * C;
    const int n_tiles = (M + 31) / 32;
    for (int tile = blockIdx.x * WPB + wave; tile < n_tiles; tile += gridDim.x * WPB) {
        const int row = tile * 32 + li;
        const bool valid = row < M;
        const unsigned ro = (unsigned)(valid ? row : 0) * C;            // rows beyond M compute on row 0 and store nothing (32-bit lane offset + scalar base)
        frag_t<T> hf[KS], xf[KS];                                        // h_{t-1} (accumulator order), x_t (natural order)
        float creg[NCB][16];
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
#pragma unroll
            for (int q = 0; q < 2; q++) hf[2 * cb + q] = s2_load_acc_frag(Hall + ro, cb, q, half);
            if (c0 != nullptr) s2_load_acc_f32(c0 + ro, cb, half, creg[cb]);
            else {
#pragma unroll
                for (int r = 0; r < 16; r++) creg[cb][r] = 0.f;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++) xf[ks] = frag_load<T>(x_all + ro + (2 * ks + half) * 8);
        for (int t = 0; t < Tn; t++) {
            frag_t<T> xn[KS];
            {                                                            // next step's rows: a whole step ahead of their use
                const T* xs = uniform_ptr(x_all + (size_t)(t + 1 < Tn ? t + 1 : t) * MC);
#pragma unroll
                for (int ks = 0; ks < KS; ks++) xn[ks] = frag_load<T>(xs + ro + (2 * ks + half) * 8);
            }
            // one 32-channel block at a time (four gate accumulators = 64 registers; both blocks at once spill at two waves per SIMD)
            frag_t<T> hnew[KS];
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                f32x16 acc[4];
#pragma unroll
                for (int g = 0; g < 4; g++) acc_load_rows(acc[g], kb + g * C + cb * 32, half);              // bias = initial value
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                    for (int g = 0; g < 4; g++) mma32(acc[g], opm_load_frag<T>(Wx, 4 * C, g * C + cb * 32 + li, 2 * ks + half), xf[ks]);
                    sched_fence();           // (the scheduler otherwise hoists all 32 weight-fragment reads of the block: 128 registers)
                }
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
#pragma unroll
                    for (int g = 0; g < 4; g++) mma32(acc[g], opm_load_frag<T>(Wh, 4 * C, g * C + cb * 32 + li, 2 * ks + half), hf[ks]);
                    sched_fence();
                }
                // gates (rnn.py:57-67), this lane's token, 16 channels of the block in the accumulator registers
                float hn[16], cn[16];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const float f = sigmoid_f(acc[0][r]);
                    const float ig = sigmoid_f(acc[1][r]);
                    const float o = sigmoid_f(acc[2][r]);
                    const float g = tanh_f(acc[3][r]);
                    const float c = f * creg[cb][r] + ig * g;
                    creg[cb][r] = c;
                    cn[r] = c;
                    hn[r] = o * tanh_f(c);
                }
                frag_t<T> f2[2];
                s2_pack(hn, f2);
                hnew[2 * cb] = f2[0];                                    // ... = the B operand of the next step, as it is
                hnew[2 * cb + 1] = f2[1];
                u32x4 piece[2];
                s2_frags_to_rows(f2, piece);
                if (valid) {
                    T* const hd = uniform_ptr(Hall + (size_t)(t + 1) * MC) + ro + 32 * cb + 8 * half;
                    *reinterpret_cast<u32x4*>(hd) = piece[0];
                    *reinterpret_cast<u32x4*>(hd + 16) = piece[1];
                }
                if (Csave != nullptr) {
                    s2_pack(cn, f2);
                    s2_frags_to_rows(f2, piece);
                    if (valid) {
                        T* const cd = uniform_ptr(Csave + (size_t)t * MC) + ro + 32 * cb + 8 * half;
                        *reinterpret_cast<u32x4*>(cd) = piece[0];
                        *reinterpret_cast<u32x4*>(cd + 16) = piece[1];
                    }
                }
            }
#pragma unroll
            for (int ks = 0; ks < KS; ks++) hf[ks] = hnew[ks];
#pragma unroll
            for (int ks = 0; ks < KS; ks++) xf[ks] = xn[ks];
        }
        if (valid) {
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) s2_store_acc_f32(c_last + ro, cb, half, creg[cb]);
        }
    }
}


// ===================================================================================================== backward
// Reverse scan with the gates recomputed and the weight gradients accumulated in the kernel.  Inputs as the forward's plus
// dH [Tn][M][C] (cotangent of Hall[1..], null = zeros) and dc_last fp32 [M][C] (null = zeros).  Outputs: dx_all [Tn][M][C],
// dh0 [M][C], dc0 fp32 [M][C], and per workgroup one partial record in ws: [4C][2C] weight gradient, then [4C] bias gradient.
// A workgroup = four waves = 64 tokens: wave (wm, wn) owns tokens 32 wm.. and channels 32 wn.. (all four gates of them).
//   P1  z^T = W [x_t | h_{t-1}]^T for its 128 gate rows: B = the token rows straight from HBM in operand form       (32 MFMAs)
//       gate backward (autograd of rnn.py:57-67) in the lane; dz -> LDS tile [64][4C] as 8-byte pieces, [x | h] rows -> LDS tiles
//   P2  [dx_t | dh_{t-1}]^T for its 32 x- and 32 h-columns: A = W^T through the transposing read of the one LDS image of W,
//       B = dz rows of its tokens from the tile; dx leaves as 16-byte row pieces, dh_{t-1} stays in the registers             (32 MFMAs)
//   P3  dW[n][k] += sum_tok dz[tok][n] [x | h][tok][k] for its 64 rows n (both operands = transposing reads of the tiles), the
//       bias gradient as one more product against a fragment of ones                                                     (32 + 8 MFMAs)
// Schedule (one wave per SIMD, so overlap has to come from inside the wave): P2 of step t, then P1 of step t - 1 queue on the
// matrix pipe while dx_t is packed and stored; the gate backward of step t - 1 (VALU) is interleaved with P3 of step t (MFMA) in
// eight chunks; two barriers per step (dz tile consumed / written); the [x | h] tiles are double-buffered by step parity so
// that the rows of step t - 1 can be parked right after P1 consumed them.
// Every LDS address is "few per-lane terms (refreshed through an opaque copy of the lane id once per step) XOR / + compile-time
// constants": left to itself hipcc hoists ~150 per-lane addresses out of the time loop and spills them.
#ifndef SCAN2_HELPER_SLEEP
#define SCAN2_HELPER_SLEEP 32        // x 64 cycles
#endif
// -DSCAN2_PROF: wave 0 of workgroup 0 accumulates the shader-clock time of its phases into ws[grid * REC + 0..7] (measurement build only)
#ifdef SCAN2_PROF
#define S2P_DECL unsigned long long s2p_t = 0, s2p_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const bool s2p_on = blockIdx.x == 0 && wave8 == 0;
#define S2P_START() do { if (s2p_on) s2p_t = __builtin_amdgcn_s_memtime(); } while (0)
#define S2P_MARK(i) do { if (s2p_on) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); s2p_acc[i] += n_ - s2p_t; s2p_t = n_; } } while (0)
#define S2P_FLUSH() do { if (s2p_on && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; i_++) ws[(size_t)gridDim.x * S::REC + i_] = (float)s2p_acc[i_]; } while (0)
#else
#define S2P_DECL
#define S2P_START()
#define S2P_MARK(i)
#define S2P_FLUSH()
#endif
struct Scan2BwdSmem {
    static constexpr int C = 64, TM = 64;
    static constexpr int W_PART = 4 * C * 128;
    static constexpr int DZ = 4 * TM * 128;                // [TM][4C] bf16: four sub-tiles of [TM][128 B]
    static constexpr int XT = TM * 128;                    // [TM][C]
    static constexpr int OFF_WX = 0, OFF_WH = W_PART, OFF_DZ = 2 * W_PART, OFF_XH = OFF_DZ + DZ, XH_BUF = 2 * XT,
                         OFF_BIAS = OFF_XH + 2 * XH_BUF, BYTES = OFF_BIAS + 4 * C * 4;
    static constexpr size_t REC = (size_t)4 * C * 2 * C + 4 * C;       // floats per workgroup record
};

// transposing-read addressing of common.hpp:TrFeat with the chunk terms pre-shifted: one v_xad per ds_read_b64_tr_b16 when
// the token step is a compile-time constant.  f[e] = X[tok0 + 8 (lane >> 5) + e][feat0 + (lane & 31)]
struct Tr2 {
    int base, lo4, hi4;
    __device__ __forceinline__ void init(int feat0, int rows, int lane) {
        const int tl = 8 * (lane >> 5) + ((lane & 15) >> 2);
        const int col = feat0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        const int byte = (col & 63) * 2;
        base = (col >> 6) * rows * 128 + tl * 128 + (byte & 15);
        lo4 = ((byte >> 4) ^ ((tl >> 1) & 7)) << 4;
        hi4 = ((byte >> 4) ^ (((tl + 4) >> 1) & 7)) << 4;
    }
    template <int TOK0> __device__ __forceinline__ frag_t<bf16> load(const char* tile) const {
        constexpr int u4 = ((TOK0 >> 4) & 7) << 4;
        const char* const plo = tile + TOK0 * 128 + ((lo4 ^ u4) + base);
        const char* const phi = tile + TOK0 * 128 + 512 + ((hi4 ^ u4) + base);
        return frag_from_tr<bf16>(reinterpret_cast<const bf16*>(plo), reinterpret_cast<const bf16*>(phi));
    }
};

template <int N> struct IntC { static constexpr int v = N; };
template <int... I, class F> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(IntC<I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__global__ void __launch_bounds__(512)
lstm_scan2_bwd_kernel(const bf16* __restrict__ x_all, const bf16* __restrict__ Hall, const bf16* __restrict__ Csave,
                      const float* __restrict__ c0, const bf16* __restrict__ dH, const float* __restrict__ dc_last,
                      const bf16* __restrict__ W, const float* __restrict__ bias, bf16* __restrict__ dx_all, bf16* __restrict__ dh0,
                      float* __restrict__ dc0, float* __restrict__ ws, int M, int Tn) {
    typedef bf16 T;
    typedef Scan2BwdSmem S;
    constexpr int C = S::C, TM = S::TM, KS = C / 16;
    __shared__ __attribute__((aligned(16))) char smem[S::BYTES];
    char* const Adz = smem + S::OFF_DZ;
    float* const kb = reinterpret_cast<float*>(smem + S::OFF_BIAS);
    // Eight waves, two roles (round 4, third cut): waves 0-3 carry the recurrence (P1, gate backward, P2) for 32 tokens x 32
    // channels each; waves 4-7 only accumulate the weight gradient (P3) from the tiles the others publish.  Neither role needs
    // more than 256 registers (the 160 weight-gradient accumulators were what pushed the single-role kernel to 512 = one wave per
    // SIMD), so a SIMD holds one wave of each role and the helper's MFMA work runs under the other's gate math.
    const int tid = threadIdx.x, wave8 = wave_uniform(tid >> 6);
    const bool helper = wave8 >= 4;
    const int wave = wave8 & 3;                            // recurrence: (wm, wn); helper: weight-gradient row blocks 2 wave, 2 wave + 1
    const int wm = wave >> 1, wn = wave & 1;
    for (int f = tid; f < 4 * C * 16; f += 512) {          // W -> LDS, both halves in natural column order
        const int n = f >> 4, g8 = f & 15;
        const frag_t<T> v = frag_load<T>(W + (size_t)n * 2 * C + g8 * 8);
        if (g8 < 8) opm_store_frag<T>(smem + S::OFF_WX, 4 * C, n, g8, v);
        else opm_store_frag<T>(smem + S::OFF_WH, 4 * C, n, g8 - 8, v);
    }
    for (int i = tid; i < 4 * C; i += 512) kb[i] = bias[i];
    // ---- per-lane address terms, recomputed from an opaque copy of the lane id at the top of every step ----
    int lane, li, half;
    int w_base, w_v;          // weight fragments (P1): row n = g C + 32 wn + li -> smem + g * 8192 + ((w_v ^ c) + w_base), c = ((2 ks) ^ ((g & 1) << 2)) << 4
    int r_base, r_hs, r_s;    // tile row R = 32 wm + li: R * 128, ((half ^ swz(R)) << 4), (swz(R) << 4)
    Tr2 tr_w, tr_dz[2], tr_xh[2];
    auto refresh_lane = [&]() __attribute__((always_inline)) {
        int l = tid & 63;
        opaque_vgpr(l);
        lane = l; li = l & 31; half = l >> 5;
        const int s0 = ((li >> 1) & 7) ^ ((wn << 1) | (li >> 4));
        w_base = (wn * 32 + li) * 128;
        w_v = (half ^ s0) << 4;
        const int R = wm * 32 + li, sR = ((R >> 1) & 7) ^ ((R >> 4) & 7);
        r_base = R * 128;
        r_hs = (half ^ sR) << 4;
        r_s = sR << 4;
        tr_w.init(wn * 32, 4 * C, lane);
        tr_dz[0].init((2 * wave) * 32, TM, lane);
        tr_dz[1].init((2 * wave + 1) * 32, TM, lane);
        tr_xh[0].init(0, TM, lane);
        tr_xh[1].init(32, TM, lane);
    };
    refresh_lane();
    __syncthreads();
    const size_t MC = (size_t)M * C;
    const int n_tiles = (M + TM - 1) / TM;

    if (helper) {
        // ================= weight-gradient waves: P3 of every step from the published tiles =================
        f32x16 dwacc[2][4], dbacc[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            acc_zero(dbacc[i]);
#pragma unroll
            for (int j = 0; j < 4; j++) acc_zero(dwacc[i][j]);
        }
        frag_t<T> ones;
#pragma unroll
        for (int e = 0; e < 8; e++) ones[e] = (T)1.0f;
        for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
            lds_barrier();                                 // (B') the recurrence waves may overwrite the tiles of the previous tile's last step
            lds_barrier();                                 // (A)  tiles of step Tn - 1 complete
            for (int t = Tn - 1; t >= 0; t--) {
                refresh_lane();
                const char* const xh_cur = smem + S::OFF_XH + (t & 1) * S::XH_BUF;
#ifndef RVT_EMU
                // the recurrence wave of this SIMD starts its own 64 MFMAs (P2, P1) at this barrier and then spends ~5k cycles on VALU
                // work: stay off the matrix pipe until it is through with them
                __builtin_amdgcn_s_sleep(SCAN2_HELPER_SLEEP);
#endif
                static_for<4>([&](auto kq) __attribute__((always_inline)) {
                    constexpr int K0 = 16 * decltype(kq)::v;
                    frag_t<T> a[2], b[4];
#pragma unroll
                    for (int i = 0; i < 2; i++) a[i] = tr_dz[i].load<K0>(Adz);
#pragma unroll
                    for (int j = 0; j < 4; j++) b[j] = tr_xh[j & 1].load<K0>(xh_cur + (j < 2 ? 0 : S::XT));
#pragma unroll
                    for (int i = 0; i < 2; i++) {
#pragma unroll
                        for (int j = 0; j < 4; j++) mma32(dwacc[i][j], a[i], b[j]);
                        mma32(dbacc[i], a[i], ones);
                    }
                });
                if (t > 0) {
                    lds_barrier();                         // (B) done with the dz tile of step t
                    lds_barrier();                         // (A) tiles of step t - 1 complete
                }
            }
        }
        // per-workgroup partial record: [4C][2C] weight gradient, [4C] bias gradient (every column of dbacc is the same sum: column 0)
        float* const p_dw = ws + (size_t)blockIdx.x * S::REC;
        float* const p_db = p_dw + (size_t)4 * C * 2 * C;
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    p_dw[(size_t)((2 * wave + i) * 32 + acc_row(r, lane)) * (2 * C) + j * 32 + li] = dwacc[i][j][r];
            if (li == 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) p_db[(2 * wave + i) * 32 + acc_row(r, lane)] = dbacc[i][r];
            }
        }
        return;
    }

    // ================= recurrence waves =================
    S2P_DECL
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int row;
        bool valid;
        unsigned ro;                                                   // 32-bit lane offset: addresses are scalar base + offset
        auto refresh_row = [&]() __attribute__((always_inline)) {
            row = tile * TM + wm * 32 + li;
            valid = row < M;
            ro = (unsigned)(valid ? row : 0) * C;                      // rows beyond M compute on row 0 and store nothing
        };
        refresh_row();
        float dh_rec[16], dc_rec[16];
#pragma unroll
        for (int r = 0; r < 16; r++) { dh_rec[r] = 0.f; dc_rec[r] = 0.f; }
        if (dc_last != nullptr && valid) s2_load_acc_f32(dc_last + ro, wn, half, dc_rec);
        // operands of a step: x_t, h_{t-1} rows (natural order, 16 B per k-step), c_{t-1} and dH_t of this wave's channel block
        // (accumulator order); fetched one step ahead
        // (requesting the set earlier - right after P1 consumed the rows, c / dH into a second register set - changed nothing:
        // 2.757 -> 2.779 ms; the HBM round trip is not what the recurrence waves wait for)
        frag_t<T> xf[KS], hf[KS], cpf[2], dhf[2];
        auto fetch = [&](int t) __attribute__((always_inline)) {
            const T* xs = uniform_ptr(x_all + (size_t)t * MC);
            const T* hs = uniform_ptr(Hall + (size_t)t * MC);
            const T* cs = uniform_ptr(Csave + (size_t)(t > 0 ? t - 1 : 0) * MC);
            const T* ds = uniform_ptr(dH != nullptr ? dH + (size_t)t * MC : x_all);
#pragma unroll
            for (int ks = 0; ks < KS; ks++) {
                xf[ks] = frag_load<T>(xs + ro + (2 * ks + half) * 8);
                hf[ks] = frag_load<T>(hs + ro + (2 * ks + half) * 8);
            }
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (t > 0) cpf[q] = s2_load_acc_frag(cs + ro, wn, q, half);
                if (dH != nullptr) dhf[q] = s2_load_acc_frag(ds + ro, wn, q, half);
                else dhf[q] = frag_zero<T>();
            }
            if (t == 0) {                              // the incoming fp32 cell state, rounded to T like the saved ones (as lstm_scan.hpp does)
                float cp0[16];
                if (c0 != nullptr) s2_load_acc_f32(c0 + ro, wn, half, cp0);
                else {
#pragma unroll
                    for (int r = 0; r < 16; r++) cp0[r] = 0.f;
                }
                s2_pack(cp0, cpf);
            }
        };

        f32x16 acc[4];                                                 // pre-activations of the step whose gates come next
        auto p1 = [&]() __attribute__((always_inline)) {               // P1: recompute the pre-activations of this wave's 32 channels
#pragma unroll
            for (int g = 0; g < 4; g++) acc_load_rows(acc[g], kb + g * C + wn * 32, half);
            // software-pipelined: a matrix instruction holds the wave's issue slot for its 32 cycles, so the LDS round trip of a
            // group's fragments has to be started BEFORE the four products of the group in front of it (requested after them it
            // is waited for with the matrix pipe idle: ~150 cycles per group of 128)
            frag_t<T> wq[2][4];
            auto wload = [&](int grp, frag_t<T> (&a)[4]) __attribute__((always_inline)) {
                const int part = grp >> 2, ks = grp & 3;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int c = ((2 * ks) ^ ((g & 1) << 2)) << 4;
                    a[g] = *reinterpret_cast<const frag_t<T>*>(smem + (part ? S::OFF_WH : S::OFF_WX) + g * 8192 + ((w_v ^ c) + w_base));
                }
            };
            wload(0, wq[0]);
#pragma unroll
            for (int grp = 0; grp < 8; grp++) {
                if (grp + 1 < 8) wload(grp + 1, wq[(grp + 1) & 1]);
                sched_fence();
#pragma unroll
                for (int g = 0; g < 4; g++) mma32(acc[g], wq[grp & 1][g], (grp >> 2) ? hf[grp & 3] : xf[grp & 3]);
                sched_fence();
            }
        };
        // gate backward (autograd of rnn.py:57-67) of accumulator registers 2p, 2p+1; dz leaves as bf16 pairs
        unsigned zp[4][8];
        auto gate_pair = [&](int p) __attribute__((always_inline)) {
            float zz[4][2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int r = 2 * p + u;
                const float f = sigmoid_f(acc[0][r]);
                const float ig = sigmoid_f(acc[1][r]);
                const float o = sigmoid_f(acc[2][r]);
                const float g = tanh_f(acc[3][r]);
                const float cp = (float)cpf[r >> 3][r & 7];
                const float dh = valid ? (float)dhf[r >> 3][r & 7] + dh_rec[r] : 0.f;       // rows beyond M: dh = dc = 0 -> dz = 0
                const float tc = tanh_f(f * cp + ig * g);
                const float dc = dc_rec[r] + dh * o * (1.f - tc * tc);
                zz[0][u] = dc * cp * f * (1.f - f);
                zz[1][u] = dc * g * ig * (1.f - ig);
                zz[2][u] = dh * tc * o * (1.f - o);
                zz[3][u] = dc * ig * (1.f - g * g);
                dc_rec[r] = dc * f;
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
                bf16x2_t v;
                v[0] = (T)zz[g][0];
                v[1] = (T)zz[g][1];
                zp[g][p] = __builtin_bit_cast(unsigned, v);
            }
        };
        auto park_rows = [&](char* xh) __attribute__((always_inline)) {    // this wave's x or h rows of the step -> its [x | h] buffer
#pragma unroll
            for (int ks = 0; ks < KS; ks++)
                *reinterpret_cast<frag_t<T>*>(xh + (wn ? S::XT : 0) + ((r_hs ^ ((2 * ks) << 4)) + r_base)) = wn ? hf[ks] : xf[ks];
        };
        auto write_dz = [&]() __attribute__((always_inline)) {         // dz pieces (8 bytes: 4 channels of one token)
#pragma unroll
            for (int g = 0; g < 4; g++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const u32x2_t v = {zp[g][2 * j], zp[g][2 * j + 1]};
                    *reinterpret_cast<u32x2_t*>(Adz + g * 8192 + 8 * half + ((r_s ^ ((wn * 4 + j) << 4)) + r_base)) = v;
                }
        };

        // ---- prologue: the last time step has nothing to overlap with ----
        refresh_lane();
        refresh_row();
        fetch(Tn - 1);
        p1();
        lds_barrier();                                     // every wave is done with the previous tile's last step
        park_rows(smem + S::OFF_XH + ((Tn - 1) & 1) * S::XH_BUF);
#pragma unroll
        for (int p = 0; p < 8; p++) gate_pair(p);
        write_dz();
        if (Tn > 1) fetch(Tn - 2);
        lds_barrier();
        for (int t = Tn - 1; t >= 0; t--) {
            // tiles of step t are in LDS; the operands of step t - 1 are in registers (t > 0)
            refresh_lane();
            refresh_row();
            S2P_START();
            // ---- P2: [dx_t | dh_{t-1}]^T for this wave's 32 + 32 columns ----
            f32x16 acc2[2];
            acc_zero(acc2[0]);
            acc_zero(acc2[1]);
            {
                frag_t<T> pb[2][2], pax[2][2], pah[2][2];   // [buffer][k-step of the pair]
                auto p2load = [&](auto pp, int buf) __attribute__((always_inline)) {
#pragma unroll
                    for (int u = 0; u < 2; u++) { }
                    constexpr int KC0 = 32 * decltype(pp)::v, KC1 = KC0 + 16;   // (compile-time k-steps: LDS address = per-lane term ^ constant + immediate)
                    pb[buf][0] = *reinterpret_cast<const frag_t<T>*>(Adz + (KC0 >> 6) * 8192 + ((r_hs ^ (((KC0 >> 3) & 6) << 4)) + r_base));
                    pb[buf][1] = *reinterpret_cast<const frag_t<T>*>(Adz + (KC1 >> 6) * 8192 + ((r_hs ^ (((KC1 >> 3) & 6) << 4)) + r_base));
                    pax[buf][0] = tr_w.load<KC0>(smem + S::OFF_WX);
                    pah[buf][0] = tr_w.load<KC0>(smem + S::OFF_WH);
                    pax[buf][1] = tr_w.load<KC1>(smem + S::OFF_WX);
                    pah[buf][1] = tr_w.load<KC1>(smem + S::OFF_WH);
                };
                p2load(IntC<0>{}, 0);
                static_for<8>([&](auto pp) __attribute__((always_inline)) {
                    constexpr int PP = decltype(pp)::v;
                    if constexpr (PP + 1 < 8) p2load(IntC<PP + 1>{}, (PP + 1) & 1);
                    sched_fence();
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        mma32(acc2[0], pax[PP & 1][u], pb[PP & 1][u]);
                        mma32(acc2[1], pah[PP & 1][u], pb[PP & 1][u]);
                    }
                    sched_fence();
                });
            }
            S2P_MARK(0);
            if (t > 0) {
                p1();                                      // P1 of step t - 1 queues behind P2 ...
                park_rows(smem + S::OFF_XH + ((t - 1) & 1) * S::XH_BUF);   // ... and its rows go to the other [x | h] buffer (xf / hf dead from here)
            }
            S2P_MARK(1);
            {                                              // dx rows out while the matrix pipe works; dh_{t-1} stays
                float dxv[16];
#pragma unroll
                for (int r = 0; r < 16; r++) { dxv[r] = acc2[0][r]; dh_rec[r] = acc2[1][r]; }
                frag_t<T> f2[2];
                u32x4 piece[2];
                s2_pack(dxv, f2);
                s2_frags_to_rows(f2, piece);
                if (valid) {
                    T* const xd = uniform_ptr(dx_all + (size_t)t * MC) + ro + 32 * wn + 8 * half;
                    *reinterpret_cast<u32x4*>(xd) = piece[0];
                    *reinterpret_cast<u32x4*>(xd + 16) = piece[1];
                }
            }
            S2P_MARK(2);
            // ---- gate backward of step t - 1 (the helper waves run P3 of step t meanwhile) ----
            if (t > 0) {
#pragma unroll
                for (int p = 0; p < 8; p++) gate_pair(p);
                S2P_MARK(3);
                lds_barrier();                             // every wave is done with the dz tile of step t
                S2P_MARK(4);
                write_dz();
                if (t > 1) fetch(t - 2);                   // (xf / hf / cpf / dhf are dead from here on)
                S2P_MARK(5);
                lds_barrier();                             // tiles of step t - 1 complete
                S2P_MARK(6);
            }
        }
        {
            frag_t<T> f2[2];
            u32x4 piece[2];
            s2_pack(dh_rec, f2);
            s2_frags_to_rows(f2, piece);
            if (valid) {
                T* const hd = dh0 + ro + 32 * wn + 8 * half;
                *reinterpret_cast<u32x4*>(hd) = piece[0];
                *reinterpret_cast<u32x4*>(hd + 16) = piece[1];
                s2_store_acc_f32(dc0 + ro, wn, half, dc_rec);
            }
        }
    }
    S2P_FLUSH();
}

}  // namespace rvt
