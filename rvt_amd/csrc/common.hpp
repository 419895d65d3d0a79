// Common device/host helpers for the MI355X (gfx950) RVT backbone kernels.
//
// Conventions used by every kernel in this directory
//   * wavefront = 64 lanes; all workgroups are 256 threads (4 waves) unless stated otherwise.
//   * activations are token-major ("channels last"): X[frame][y][x][c], frame = t*B + b.
//   * T is the storage/MFMA-input type: __bf16 (performance mode) or float (parity mode, exact
//     f32 MFMA v_mfma_f32_32x32x2_f32).  Accumulation, LayerNorm/softmax statistics, the LSTM
//     cell state and all parameter gradients are always fp32.
//   * a "frag" is 8 consecutive K elements of one row: the per-lane A/B operand of one
//     v_mfma_f32_32x32x16_bf16, or of eight v_mfma_f32_32x32x2_f32 (lane l supplies row l&31,
//     k-slots 8*(l>>5)+0..7 — A and B use the same slot order, so the pairing is exact).
//   * C/D layout of a 32x32 MFMA block: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5), r in [0,16).
#pragma once
#ifndef RVT_EMU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include <stddef.h>
#include "../../include/rvt_hip.h"

namespace rvt {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <class T> struct Frag;
template <> struct Frag<float> { typedef f32x8 type; };
template <> struct Frag<bf16> { typedef bf16x8 type; };
template <class T> using frag_t = typename Frag<T>::type;

template <class T> __device__ __forceinline__ frag_t<T> frag_zero() {
    frag_t<T> z;
#pragma unroll
    for (int i = 0; i < 8; i++) z[i] = (T)0.0f;
    return z;
}
template <class T> __device__ __forceinline__ frag_t<T> frag_load(const T* p) {
    return *reinterpret_cast<const frag_t<T>*>(p);
}
template <class T> __device__ __forceinline__ void frag_store(T* p, const frag_t<T>& v) {
    *reinterpret_cast<frag_t<T>*>(p) = v;
}
template <class T> __device__ __forceinline__ void frag_to_float(const frag_t<T>& f, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = (float)f[i];
}
template <class T> __device__ __forceinline__ frag_t<T> frag_from_float(const float (&v)[8]) {
    frag_t<T> f;
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = (T)v[i];
    return f;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- MFMA: one 32x32 output block, K = 16 ------------------------------------------------------
#ifndef RVT_EMU
__device__ __forceinline__ void mma32(f32x16& c, const bf16x8& a, const bf16x8& b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& c, const f32x8& a, const f32x8& b) {
#pragma unroll
    for (int j = 0; j < 8; j++) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
}
// first product of an accumulation: C operand = 0 (an inline constant of the instruction: no zero fill of the accumulators)
__device__ __forceinline__ void mma32_zero(f32x16& c, const bf16x8& a, const bf16x8& b) {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, z, 0, 0, 0);
}
__device__ __forceinline__ void mma32_zero(f32x16& c, const f32x8& a, const f32x8& b) {       // (parity mode: zero fill + the eight k = 2 products)
#pragma unroll
    for (int i = 0; i < 16; i++) c[i] = 0.0f;
    mma32(c, a, b);
}
#else
template <class F> inline void mma32_emu(f32x16& c, const F& a, const F& b) {
    float ab[16];
    for (int j = 0; j < 8; j++) { ab[j] = (float)a[j]; ab[8 + j] = (float)b[j]; }
    auto buf = emu::exchange(ab, sizeof(ab));
    int lane = emu::g.cur->lane, col = lane & 31;
    for (int r = 0; r < 16; r++) {
        int row = acc_row(r, lane);
        float s = c[r];
        for (int h = 0; h < 2; h++) {
            const float* pa = reinterpret_cast<const float*>(buf[row + 32 * h]);
            const float* pb = reinterpret_cast<const float*>(buf[col + 32 * h]) + 8;
            for (int e = 0; e < 8; e++) s += pa[e] * pb[e];
        }
        c[r] = s;
    }
}
inline void mma32(f32x16& c, const bf16x8& a, const bf16x8& b) { mma32_emu(c, a, b); }
inline void mma32(f32x16& c, const f32x8& a, const f32x8& b) { mma32_emu(c, a, b); }
inline void mma32_zero(f32x16& c, const bf16x8& a, const bf16x8& b) { for (int i = 0; i < 16; i++) c[i] = 0.f; mma32_emu(c, a, b); }
inline void mma32_zero(f32x16& c, const f32x8& a, const f32x8& b) { for (int i = 0; i < 16; i++) c[i] = 0.f; mma32_emu(c, a, b); }
#endif

// ---- MFMA: one 16x16 output block, K = 32 (v_mfma_f32_16x16x32_bf16) ----------------------------------------------
// lane l supplies row l & 15 of A (and of B), k-slots 8 (l >> 4) + 0..7; C/D: col = l & 15 (B's row), row = 4 (l >> 4) + r, r in [0, 4).
// Same FLOP rate as the 32x32 form; used where a product has to be cut into MORE blocks than a 32x32 tiling gives (one per wave).
#ifndef RVT_EMU
__device__ __forceinline__ void mma16(f32x4& c, const bf16x8& a, const bf16x8& b) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#else
inline void mma16(f32x4& c, const bf16x8& a, const bf16x8& b) {
    float ab[16];
    for (int j = 0; j < 8; j++) { ab[j] = (float)a[j]; ab[8 + j] = (float)b[j]; }
    auto buf = emu::exchange(ab, sizeof(ab));
    const int lane = emu::g.cur->lane, col = lane & 15;
    for (int r = 0; r < 4; r++) {
        const int row = 4 * (lane >> 4) + r;
        float s = c[r];
        for (int g = 0; g < 4; g++) {
            const float* pa = reinterpret_cast<const float*>(buf[row + 16 * g]);
            const float* pb = reinterpret_cast<const float*>(buf[col + 16 * g]) + 8;
            for (int e = 0; e < 8; e++) s += pa[e] * pb[e];
        }
        c[r] = s;
    }
}
#endif

__device__ __forceinline__ void acc_zero(f32x16& c) {
#pragma unroll
    for (int i = 0; i < 16; i++) c[i] = 0.0f;
}

// ---- exact division by a runtime constant ------------------------------------------------------
struct FastDiv {
    uint32_t d, m, s;
    FastDiv() : d(1), m(1), s(0) {}
    explicit FastDiv(uint32_t d_) : d(d_) {
        s = 0;
        while ((1ull << s) < d) s++;
        m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return (__umulhi(n, m) + n) >> s; }
    __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const { q = div(n); r = n - q * d; }
};

// ---- swizzled LDS tile of [rows][128 bytes] ------------------------------------------------------
// Every GEMM operand tile row is 128 bytes (64 bf16 or 32 f32) = 8 chunks of 16 B.  A wave's
// ds_read_b128 of one logical chunk from 32 different rows would hit two 16-B slots of the 256-B
// bank row (8-way); XOR-ing the chunk index with (row>>1)&7 spreads each 16-lane read group over all
// 16 slots (MI355X guide: LDS banks for b128 = (addr/4)%64, 16-lane groups).
// The second term, (row>>4)&7, is for the TRANSPOSING loader: its lanes write rows 8*fc+f (stride 8), which the
// first term alone maps onto only two 16-B slots (8-way conflict on ds_write_b128); with it a 16-lane write group
// spreads over all eight slots (2-way).  MFMA fragment reads (32 consecutive rows) stay conflict-free.
__device__ __forceinline__ int lds_chunk_off(int row, int chunk16) {
    return row * 128 + ((chunk16 ^ ((row >> 1) & 7) ^ ((row >> 4) & 7)) << 4);
}
template <class T> struct TileGeom;
template <> struct TileGeom<bf16> { static constexpr int BK = 64, FPR = 8, CPF = 1; };   // frags/row, 16B-chunks/frag
template <> struct TileGeom<float> { static constexpr int BK = 32, FPR = 4, CPF = 2; };

template <class T> __device__ __forceinline__ void tile_store_frag(char* tile, int row, int fc, const frag_t<T>& v) {
    const u32x4* src = reinterpret_cast<const u32x4*>(&v);
#pragma unroll
    for (int c = 0; c < TileGeom<T>::CPF; c++)
        *reinterpret_cast<u32x4*>(tile + lds_chunk_off(row, fc * TileGeom<T>::CPF + c)) = src[c];
}
template <class T> __device__ __forceinline__ frag_t<T> tile_load_frag(const char* tile, int row, int fc) {
    frag_t<T> v;
    u32x4* dst = reinterpret_cast<u32x4*>(&v);
#pragma unroll
    for (int c = 0; c < TileGeom<T>::CPF; c++)
        dst[c] = *reinterpret_cast<const u32x4*>(tile + lds_chunk_off(row, fc * TileGeom<T>::CPF + c));
    return v;
}

// ---- LDS transpose read (gfx950 ds_read_b64_tr_b16) --------------------------------------------------------
// Within every group of 16 lanes the sixteen 8-byte chunks the lanes address form a 16 x 4 matrix of 16-bit elements
// (row = lane, 4 elements each); lane i receives, as element k = 0..3, element (i & 3) of the chunk addressed by lane
// 4k + (i >> 2) of its group (mapped empirically: profiles/probes/tr_read_probe.hip).  This is what turns a row-major
// [token][feature] LDS tile into MFMA operands whose contraction index is the TOKEN (in-kernel weight gradients,
// products with W^T from a single LDS image of W): see load_frag_tr below.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
__device__ __forceinline__ s16x4 tr_read4(const bf16* p) {
#ifdef RVT_EMU
    auto buf = emu::exchange(&p, sizeof(p));
    const int lane = emu::g.cur->lane, grp = lane & ~15, i = lane & 15;
    s16x4 o;
    for (int k = 0; k < 4; k++) {
        const bf16* q;
        memcpy(&q, buf[grp + 4 * k + (i >> 2)], sizeof(q));
        short h;
        memcpy(&h, q + (i & 3), 2);
        o[k] = h;
    }
    return o;
#else
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
#endif
}
// Transposed operand fragment: f[e] = X[tok0 + 8*(lane>>5) + e][feat0 + (lane&31)], e = 0..7, i.e. the A/B fragment of
// an MFMA whose rows are FEATURES feat0..feat0+31 and whose contraction runs over TOKENS tok0..tok0+15.  `at(tok, feat)`
// returns the LDS address of element (tok, feat); 4 consecutive features starting at a multiple of 4 must be contiguous
// there (true for every swizzled tile in this directory: the swizzle moves whole 16-byte chunks).
template <class T, class At> __device__ __forceinline__ frag_t<T> load_frag_tr(const At& at, int tok0, int feat0, int lane) {
    if constexpr (sizeof(T) == 2) {
        const int tl = 8 * (lane >> 5) + ((lane & 15) >> 2), fl = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        const s16x4 lo = tr_read4(at(tok0 + tl, feat0 + fl)), hi = tr_read4(at(tok0 + tl + 4, feat0 + fl));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        frag_t<T> f;
        __builtin_memcpy(&f, &v, 16);
        return f;
    } else {            // f32 parity mode: no 32-bit transpose read; eight scalar LDS reads
        frag_t<T> f;
#pragma unroll
        for (int e = 0; e < 8; e++) f[e] = *at(tok0 + 8 * (lane >> 5) + e, feat0 + (lane & 31));
        return f;
    }
}
// the two per-lane addresses of load_frag_tr when the caller wants to hoist them out of a loop
template <class T> __device__ __forceinline__ frag_t<T> frag_from_tr(const bf16* p_lo, const bf16* p_hi) {
    const s16x4 lo = tr_read4(p_lo), hi = tr_read4(p_hi);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    frag_t<T> f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}

// Hoisted addressing of load_frag_tr for a swizzled LDS operand matrix (K-subtiles of [rows][128 B]) and ONE block of 32
// features starting at feat0: everything but the token step is a per-lane constant, and for tok0 a multiple of 16 the two
// swizzle terms of lds_chunk_off reduce to a per-lane part XOR (tok0 >> 4) & 7.  bf16: two ds_read_b64_tr_b16 and three
// integer ops per fragment; f32 (parity mode): the generic eight scalar reads.
template <class T> struct TrFeat {
    int row_lo, c_lo, c_hi, feat0_, rows_;
    __device__ __forceinline__ void init(int feat0, int rows, int lane) {
        feat0_ = feat0; rows_ = rows;
        constexpr int BK = TileGeom<T>::BK;
        const int tl = 8 * (lane >> 5) + ((lane & 15) >> 2);
        const int col = feat0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        const int byte = (col % BK) * (int)sizeof(T);
        row_lo = (col / BK) * rows * 128 + tl * 128 + (byte & 15);
        c_lo = (byte >> 4) ^ ((tl >> 1) & 7);
        c_hi = (byte >> 4) ^ (((tl + 4) >> 1) & 7);
    }
    __device__ __forceinline__ frag_t<T> load(const char* tile, int tok0, int lane) const {
        if constexpr (sizeof(T) == 2) {
            const int u = (tok0 >> 4) & 7;
            const char* const b = tile + tok0 * 128 + row_lo;
            return frag_from_tr<T>(reinterpret_cast<const bf16*>(b + ((c_lo ^ u) << 4)),
                                   reinterpret_cast<const bf16*>(b + 512 + ((c_hi ^ u) << 4)));
        } else {
            constexpr int BK = TileGeom<T>::BK;
            const int rows = rows_;
            auto at = [&](int tok, int feat) -> const T* {
                const int byte = (feat % BK) * (int)sizeof(T);
                return reinterpret_cast<const T*>(tile + (size_t)(feat / BK) * rows * 128 + lds_chunk_off(tok, byte >> 4) + (byte & 15));
            };
            return load_frag_tr<T>(at, tok0, feat0_, lane);
        }
    }
};

// A value the optimiser has to treat as freshly produced here: everything derived from it stays inside the loop it is used in
// (hipcc hoists per-lane address tables out of loops and spills them; see ppgemm.hpp / dgrad_ln.hpp).
__device__ __forceinline__ void opaque_vgpr(int& v) {
#ifndef RVT_EMU
    asm volatile("" : "+v"(v));
#endif
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory: hipcc emits
// `s_waitcnt vmcnt(0)` in front of the barrier, i.e. every wave waits until its outstanding global STORES are
// acknowledged (CDNA counts stores on vmcnt).  In these kernels barriers only protect LDS tiles / staging buffers —
// global data is never exchanged between waves inside a launch — so waiting for LDS (lgkmcnt) is sufficient and lets
// the epilogue stores drain while the next tile is already being computed.
__device__ __forceinline__ void lds_barrier() {
#ifdef RVT_EMU
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// a value that is the same in every lane of the wave, moved to a scalar register (the compiler cannot know that e.g.
// threadIdx.x >> 6 is wave-uniform): addresses built from it take the scalar-base + 32-bit lane-offset form
__device__ __forceinline__ int wave_uniform(int v) {
#ifdef RVT_EMU
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}

// wait for every outstanding global access of this wave (loads AND stores).  Besides the obvious use: stores and loads
// retire out of order with respect to each other, so while stores are pending the compiler can only wait with vmcnt(0) —
// a software-pipelined load loop entered with stores in flight (the previous item's epilogue) gets a full drain at its loop
// header in EVERY iteration.  Draining once in front of the loop restores the counted waits inside it.
__device__ __forceinline__ void drain_vmem() {
#ifndef RVT_EMU
    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0), expcnt / lgkmcnt untouched
#endif
}

// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), result in all of them: four row rotations
__device__ __forceinline__ float row16_sum(float v) {
#ifdef RVT_EMU
    auto buf = emu::exchange(&v, sizeof(v));
    const int base = emu::g.cur->lane & ~15;
    float s = 0.f;
    for (int i = 0; i < 16; i++) { float t; memcpy(&t, buf[base + i], 4); s += t; }
    return s;
#else
#define RVT_DPP_ROR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x120 + (n), 0xf, 0xf, false))
    v += RVT_DPP_ROR(v, 8);
    v += RVT_DPP_ROR(v, 4);
    v += RVT_DPP_ROR(v, 2);
    v += RVT_DPP_ROR(v, 1);
#undef RVT_DPP_ROR
    return v;
#endif
}

// exchange between the two 32-lane halves of a wave: afterwards lanes 0-31 hold (own a, partner's a) and lanes 32-63
// hold (partner's b, own b)   [v_permlane32_swap_b32: vdst.lanes[32..63] <-> src0.lanes[0..31]]
__device__ __forceinline__ void swap32(float& a, float& b) {
#ifdef RVT_EMU
    float ab[2] = {a, b};
    auto buf = emu::exchange(ab, sizeof(ab));
    const int lane = emu::g.cur->lane;
    const float* p = reinterpret_cast<const float*>(buf[lane ^ 32]);
    if (lane < 32) b = p[0]; else a = p[1];
#else
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
#endif
}

// accumulator block (col = token = lane & 31, rows = 32 features) -> two 8-feature row pieces of this lane's token:
// o[m][e] = feature 16 m + 8 (lane >> 5) + e — exactly the (k-step m, half) operand piece the rows were loaded in
__device__ __forceinline__ void acc_to_rows(const f32x16& c, float (&o)[2][8]) {
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
        for (int w = 0; w < 4; w++) {
            float a = c[8 * m + w], b = c[8 * m + 4 + w];
            swap32(a, b);
            o[m][w] = a;
            o[m][4 + w] = b;
        }
}

// keep the instruction scheduler from moving anything across this point (used to pin prefetch loads early)
__device__ __forceinline__ void sched_fence() {
#ifndef RVT_EMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}

__device__ __forceinline__ float fast_exp2(float x) {
#ifdef RVT_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);       // v_exp_f32
#endif
}

// ---- scalar math -------------------------------------------------------------------------------
__device__ __forceinline__ float fast_rcp(float x) {
#ifdef RVT_EMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);        // v_rcp_f32, 1 ulp
#endif
}
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. at fp32 round-off): one v_exp, one v_rcp, a 5-term Horner
// instead of the ~100-op libm erff.
// exact-erf GELU of the reference (layers/activations.py:138-145) and its derivative.  The fc1 epilogues that evaluate
// them are VALU-bound (79 G evaluations per RVT-Base step-stage), so the A&S form is folded for the fewest instructions:
//   q = 0.5 * poly(t) * exp(-x^2/2),  t = 1/(1 + p/sqrt2 |x|);   Phi(x) = x < 0 ? q : 1 - q;   g = x Phi,  g' = Phi + x phi(x)
// (halved coefficients, |x| as a source modifier, exp through one v_exp_f32: 14 plain + 2 quarter-rate instructions
// for both values).
__device__ __forceinline__ float gelu_phi(float x, float& e) {                    // Phi(x); e = exp(-x^2/2)
    const float t = fast_rcp(fmaf(0.23164189f, fabsf(x), 1.0f));                 // 0.3275911 / sqrt(2)
    e = fast_exp2(x * x * -0.72134752044448170f);                                 // 2^(-x^2 log2(e) / 2)
    const float poly = t * (0.127414796f + t * (-0.142248368f + t * (0.7107068705f + t * (-0.7265760135f + t * 0.5307027145f))));
    const float q = poly * e;
    return x < 0.0f ? q : 1.0f - q;
}
__device__ __forceinline__ float gelu_f(float x) {
    float e;
    return x * gelu_phi(x, e);
}
__device__ __forceinline__ float gelu_grad_f(float x) {
    float e;
    const float c = gelu_phi(x, e);
    return fmaf(x * 0.3989422804014327f, e, c);
}
// both at once (one evaluation): used by the fc1 epilogues that save GELU(x) and GELU'(x)
__device__ __forceinline__ void gelu_both_f(float x, float& g, float& gp) {
    float e;
    const float c = gelu_phi(x, e);
    g = x * c;
    gp = fmaf(x * 0.3989422804014327f, e, c);
}
// ... and for eight values: the same evaluation on float2 lanes with explicit fused multiply-adds, which map onto the
// packed-fp32 VALU ops of CDNA (v_pk_fma_f32 / v_pk_mul_f32: two values per instruction; exp, rcp and the sign select
// stay scalar).  ~40 % fewer VALU instructions than eight scalar evaluations in the fc1 epilogues.
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) {
#ifdef RVT_EMU
    return f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])};
#else
    return __builtin_elementwise_fma(a, b, c);
#endif
}
__device__ __forceinline__ void gelu_both_8(const float (&x)[8], float (&g)[8], float (&gp)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        const f32x2 v = {x[i], x[i + 1]};
        const f32x2 av = {fabsf(v[0]), fabsf(v[1])};
        const f32x2 d = fma2(av, f32x2{0.23164189f, 0.23164189f}, f32x2{1.0f, 1.0f});
        const f32x2 t = {fast_rcp(d[0]), fast_rcp(d[1])};
        const f32x2 a = v * v * -0.72134752044448170f;
        const f32x2 e = {fast_exp2(a[0]), fast_exp2(a[1])};
        f32x2 poly = fma2(t, f32x2{0.5307027145f, 0.5307027145f}, f32x2{-0.7265760135f, -0.7265760135f});
        poly = fma2(t, poly, f32x2{0.7107068705f, 0.7107068705f});
        poly = fma2(t, poly, f32x2{-0.142248368f, -0.142248368f});
        poly = fma2(t, poly, f32x2{0.127414796f, 0.127414796f});
        const f32x2 q = (t * poly) * e;
        const f32x2 omq = 1.0f - q;
        const f32x2 c = {v[0] < 0.0f ? q[0] : omq[0], v[1] < 0.0f ? q[1] : omq[1]};
        const f32x2 gg = v * c;
        const f32x2 dd = fma2(v * 0.3989422804014327f, e, c);
        g[i] = gg[0]; g[i + 1] = gg[1];
        gp[i] = dd[0]; gp[i + 1] = dd[1];
    }
}
// sigmoid / tanh of the ConvLSTM gates (rnn.py:57-67) on the hardware exp2 / rcp: 4 and 5 instructions.
//   sigmoid(x) = 1 / (1 + 2^(-x log2 e));   tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e))   (saturates to -1 / +1 through 2^-inf = 0
//   and 1/inf = 0; absolute error ~1e-7 like the (1-e)/(1+e) form it replaces — both cancel near 0, where |tanh| is tiny)
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.0f + fast_exp2(x * -1.4426950408889634f)); }
__device__ __forceinline__ float tanh_f(float x) {
    return fmaf(-2.0f, fast_rcp(1.0f + fast_exp2(x * 2.8853900817779268f)), 1.0f);
}
// gate pre-activation z + bias folded into the exp2 argument: sigmoid(z + b) with nb = -b log2 e, tanh(z + b) with
// tb = 2 b log2 e (one fma instead of add + mul)
__device__ __forceinline__ float sigmoid_zb(float z, float nb) { return fast_rcp(1.0f + fast_exp2(fmaf(z, -1.4426950408889634f, nb))); }
__device__ __forceinline__ float tanh_zb(float z, float tb) {
    return fmaf(-2.0f, fast_rcp(1.0f + fast_exp2(fmaf(z, 2.8853900817779268f, tb))), 1.0f);
}

// ---- the process-wide tuning / routing record (include/rvt_hip.h; defined in capi_core.hip)
extern ::RvtTuning g_tuning;

// ---- error plumbing for the C ABI ------------------------------------------------------------------
void set_last_error(const char* fmt, ...);
#define RVT_CHECK(cond, ...)                      \
    do {                                          \
        if (!(cond)) {                            \
            ::rvt::set_last_error(__VA_ARGS__);   \
            return 1;                             \
        }                                         \
    } while (0)
int check_launch(const char* what);

}  // namespace rvt
