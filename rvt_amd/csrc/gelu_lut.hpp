// GELU / GELU' through LDS lookup tables (shared by the fused MLP kernels).
#pragma once
#include "common.hpp"

namespace rvt {

// ---- GELU through an LDS table --------------------------------------------------------------------------------------
// The exact-erf GELU costs ~24 VALU-issue slots per element (two quarter-rate transcendentals) and these kernels are
// VALU-bound on it (SQ counters: VALU 72 % of SIMD time, MFMA 15 %), while the LDS pipe is almost idle.  Phi(x) and
// GELU'(x) = Phi(x) + x phi(x) are smooth and bounded, so each is tabulated at GELU_LUT_N points of [-X, X] as
// (value, forward difference) pairs and evaluated by ONE 8-byte LDS gather plus a linear interpolation: 7 VALU slots.
// Interpolation error <= h^2/8 max|f"| = 4e-6 (h = 2X / N = 0.0117; max|Phi"| = 0.24, max of the second derivative of
// GELU' ~ 0.5) - three orders below the 1e-3 parity bar; beyond |x| = X both functions are at their limits to 1e-9.
// The table is filled per workgroup from the A&S 7.1.26 evaluation of common.hpp (|err| <= 1.5e-7).
constexpr int GELU_LUT_N = 1024;
constexpr float GELU_LUT_X = 6.0f;
constexpr int GELU_LUT_BYTES = GELU_LUT_N * 8;
template <bool GRAD> __device__ __forceinline__ void gelu_lut_fill(float* lut, int tid, int nthreads) {
    const float h = 2.0f * GELU_LUT_X / (float)GELU_LUT_N;
    for (int i = tid; i < GELU_LUT_N; i += nthreads) {
        const float x0 = -GELU_LUT_X + h * (float)i, x1 = x0 + h;
        float e0, e1;
        float f0 = gelu_phi(x0, e0), f1 = gelu_phi(x1, e1);
        if (GRAD) { f0 = fmaf(x0 * 0.3989422804014327f, e0, f0); f1 = fmaf(x1 * 0.3989422804014327f, e1, f1); }
        lut[2 * i] = f0;
        lut[2 * i + 1] = f1 - f0;
    }
}
// table values at the 16 accumulator registers of a block: Phi (GRAD = false table) or GELU' (GRAD = true table).  Three
// phases — all indices, all gathers, all interpolations — so that the sixteen LDS round trips overlap instead of each
// being waited for in turn (two waves per SIMD do not hide a dependent gather chain).
template <int BATCH = 16>
__device__ __forceinline__ void gelu_lut_eval16(const float* lut, const f32x16& x, float (&out)[16]) {
    const float s = (float)GELU_LUT_N / (2.0f * GELU_LUT_X);
#pragma unroll
    for (int r0 = 0; r0 < 16; r0 += BATCH) {            // BATCH gathers in flight at a time (registers vs latency hiding)
        float fr[BATCH];
        const f32x2* p[BATCH];
#pragma unroll
        for (int r = 0; r < BATCH; r++) {
            float t = fmaf(x[r0 + r], s, GELU_LUT_X * s);
            t = fminf(fmaxf(t, 0.0f), (float)GELU_LUT_N - 0.001f);
            const float fl = floorf(t);
            fr[r] = t - fl;
            p[r] = reinterpret_cast<const f32x2*>(lut) + (int)fl;
        }
        sched_fence();
        f32x2 ab[BATCH];
#pragma unroll
        for (int r = 0; r < BATCH; r++) ab[r] = *p[r];
        sched_fence();
#pragma unroll
        for (int r = 0; r < BATCH; r++) out[r0 + r] = fmaf(fr[r], ab[r][1], ab[r][0]);
    }
}

// ---- nearest-entry tables (the bf16 kernels) -------------------------------------------------------------------------------
// The interpolating form above still costs 8.5 VALU slots per value in the compiled loop (scale, clamp, floor, convert,
// fraction, address, fma, three moves per pair for the packed fma, product) and the chain kernels are VALU-bound on exactly
// that (ISA count: 136 VALU per 16 values against 4 + 4 MFMAs).  A bf16 result carries 8 mantissa bits, so the table may be
// read at the NEAREST of 8192 points instead: |error| <= h/2 max|f'| = 7.3e-4 * 0.40 = 2.9e-4 for Phi (relative to
// Phi ~ 0.5 there: 6e-4, a third of the bf16 half-ulp 2e-3 the product is rounded with right after) and 5.8e-4 for GELU'
// (max|GELU"| = 0.80 at 0).  The index comes out of the float ADD itself: t = x s + (X s + 2^23) has ulp 1, so its mantissa
// is round((x + X) s) and `bits(t) - bits(2^23)` is the table index - scale, round and convert in one fma; clamp = one
// v_med3; address = one v_lshl_add.  4 VALU slots per value with the product.  The fp32 kernels (parity bar 1e-3) keep the
// interpolating table.
constexpr int GELU_NLUT_N = 8192;
constexpr int GELU_NLUT_BYTES = (GELU_NLUT_N + 4) * 4;        // entries 0 .. N
constexpr int GELU_NLUT2_BYTES = (GELU_NLUT_N + 2) * 8;       // { Phi, GELU' } pairs
__device__ __forceinline__ float gelu_nlut_point(int i, float& e) {
    const float x = -GELU_LUT_X + (2.0f * GELU_LUT_X / (float)GELU_NLUT_N) * (float)i;
    const float f = gelu_phi(x, e);
    e = fmaf(x * 0.3989422804014327f, e, f);                   // GELU'(x)
    return f;
}
template <bool GRAD> __device__ __forceinline__ void gelu_nlut_fill(float* lut, int tid, int nthreads) {
    for (int i = tid; i <= GELU_NLUT_N; i += nthreads) {
        float gp;
        const float f = gelu_nlut_point(i, gp);
        lut[i] = GRAD ? gp : f;
    }
}
__device__ __forceinline__ void gelu_nlut2_fill(float* lut, int tid, int nthreads) {
    for (int i = tid; i <= GELU_NLUT_N; i += nthreads) {
        float gp;
        const float f = gelu_nlut_point(i, gp);
        *reinterpret_cast<f32x2*>(lut + 2 * i) = f32x2{f, gp};
    }
}
__device__ __forceinline__ int gelu_nlut_index(float x) {
    const float s = (float)GELU_NLUT_N / (2.0f * GELU_LUT_X), magic = 8388608.0f;       // 2^23 = 0x4B000000
    float t = fmaf(x, s, GELU_LUT_X * s + magic);
    t = fminf(fmaxf(t, magic), magic + (float)GELU_NLUT_N);
    return (int)(__builtin_bit_cast(unsigned int, t) - 0x4B000000u);
}
// a product hipcc's SLP vectoriser will not pack: left to itself it pairs the products of REGISTERS 2i+1, 2i+2 into
// v_pk_mul_f32 (the first gather is waited for on its own), which costs a v_mov per value to line the pairs up and a
// v_alignbit per bf16 pair to undo the shift after the conversion - 76 slots per 32 values instead of 48.  Odd elements
// are written as fma(a, b, +0) (same value up to the sign of a zero; not foldable without nsz), so that neighbouring
// products are not isomorphic.  (An inline-asm v_mul is not an option: the hazard recogniser does not see it as a VALU
// instruction and an MFMA reading its result loses the wait states - wrong columns in the fp32 kernel.)
__device__ __forceinline__ float mul_nopack(float a, float b, int r) { return (r & 1) ? __builtin_fmaf(a, b, 0.0f) : a * b; }
__device__ __forceinline__ void gelu_nlut_eval16(const float* lut, const f32x16& x, float (&out)[16]) {
    int idx[16];
#pragma unroll
    for (int r = 0; r < 16; r++) idx[r] = gelu_nlut_index(x[r]);
    sched_fence();
#pragma unroll
    for (int r = 0; r < 16; r++) out[r] = lut[idx[r]];
    sched_fence();
}
// g = x Phi(x), gp = GELU'(x) for eight values from the pair table
__device__ __forceinline__ void gelu_both_nlut2_8(const float* lut, const float (&x)[8], float (&g)[8], float (&gp)[8]) {
    int idx[8];
#pragma unroll
    for (int r = 0; r < 8; r++) idx[r] = gelu_nlut_index(x[r]);
    sched_fence();
    f32x2 ab[8];
#pragma unroll
    for (int r = 0; r < 8; r++) ab[r] = reinterpret_cast<const f32x2*>(lut)[idx[r]];
    sched_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) { g[r] = mul_nopack(x[r], ab[r][0], r); gp[r] = ab[r][1]; }
}

// table flavour by storage type: float -> interpolating (exact to 4e-6), bf16 -> nearest entry
template <class T> struct GeluTab {
    static constexpr int BYTES = GELU_LUT_BYTES;
    template <bool GRAD> static __device__ __forceinline__ void fill(float* lut, int tid, int n) { gelu_lut_fill<GRAD>(lut, tid, n); }
    static __device__ __forceinline__ void eval16(const float* lut, const f32x16& x, float (&out)[16]) { gelu_lut_eval16(lut, x, out); }
};
template <> struct GeluTab<bf16> {
    static constexpr int BYTES = GELU_NLUT_BYTES;
    template <bool GRAD> static __device__ __forceinline__ void fill(float* lut, int tid, int n) { gelu_nlut_fill<GRAD>(lut, tid, n); }
    static __device__ __forceinline__ void eval16(const float* lut, const f32x16& x, float (&out)[16]) { gelu_nlut_eval16(lut, x, out); }
};

}  // namespace rvt
