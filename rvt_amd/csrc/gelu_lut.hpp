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

// (measured: in the LDS-staged kernels of mlp.hpp the table form of gelu_both_8 changes nothing — they are barrier- / LDS-bound)
// both tables side by side ([Phi: N pairs][GELU': N pairs], 2 * GELU_LUT_BYTES) and the fused evaluation of eight values:
// g = x Phi(x), gp = GELU'(x) — the LDS-table form of gelu_both_8 (common.hpp)
__device__ __forceinline__ void gelu_lut_fill_both(float* lut, int tid, int nthreads) {
    gelu_lut_fill<false>(lut, tid, nthreads);
    gelu_lut_fill<true>(lut + 2 * GELU_LUT_N, tid, nthreads);
}
__device__ __forceinline__ void gelu_both_lut8(const float* lut, const float (&x)[8], float (&g)[8], float (&gp)[8]) {
    const float s = (float)GELU_LUT_N / (2.0f * GELU_LUT_X);
    float fr[8];
    const f32x2* p[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        float t = fmaf(x[r], s, GELU_LUT_X * s);
        t = fminf(fmaxf(t, 0.0f), (float)GELU_LUT_N - 0.001f);
        const float fl = floorf(t);
        fr[r] = t - fl;
        p[r] = reinterpret_cast<const f32x2*>(lut) + (int)fl;
    }
    sched_fence();
    f32x2 a[8], b[8];
#pragma unroll
    for (int r = 0; r < 8; r++) { a[r] = p[r][0]; b[r] = p[r][GELU_LUT_N]; }
    sched_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        g[r] = x[r] * fmaf(fr[r], a[r][1], a[r][0]);
        gp[r] = fmaf(fr[r], b[r][1], b[r][0]);
    }
}

// ---- both functions from ONE 16-byte gather -------------------------------------------------------------------------------
// interleaved table: entry i = { Phi(x_i), Phi(x_i+1) - Phi(x_i), GELU'(x_i), GELU'(x_i+1) - GELU'(x_i) }
constexpr int GELU_LUT4_BYTES = GELU_LUT_N * 16;
__device__ __forceinline__ void gelu_lut4_fill(float* lut, int tid, int nthreads) {
    const float h = 2.0f * GELU_LUT_X / (float)GELU_LUT_N;
    for (int i = tid; i < GELU_LUT_N; i += nthreads) {
        const float x0 = -GELU_LUT_X + h * (float)i, x1 = x0 + h;
        float e0, e1;
        const float f0 = gelu_phi(x0, e0), f1 = gelu_phi(x1, e1);
        const float g0 = fmaf(x0 * 0.3989422804014327f, e0, f0), g1 = fmaf(x1 * 0.3989422804014327f, e1, f1);
        *reinterpret_cast<f32x4*>(lut + 4 * i) = f32x4{f0, f1 - f0, g0, g1 - g0};
    }
}
// g = x Phi(x), gp = GELU'(x) for eight values: all indices, all gathers, all interpolations (the gathers overlap)
__device__ __forceinline__ void gelu_both_lut4_8(const float* lut, const float (&x)[8], float (&g)[8], float (&gp)[8]) {
    const float s = (float)GELU_LUT_N / (2.0f * GELU_LUT_X);
    float fr[8];
    const f32x4* p[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        float t = fmaf(x[r], s, GELU_LUT_X * s);
        t = fminf(fmaxf(t, 0.0f), (float)GELU_LUT_N - 0.001f);
        const float fl = floorf(t);
        fr[r] = t - fl;
        p[r] = reinterpret_cast<const f32x4*>(lut) + (int)fl;
    }
    sched_fence();
    f32x4 ab[8];
#pragma unroll
    for (int r = 0; r < 8; r++) ab[r] = *p[r];
    sched_fence();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        g[r] = x[r] * fmaf(fr[r], ab[r][1], ab[r][0]);
        gp[r] = fmaf(fr[r], ab[r][3], ab[r][2]);
    }
}

}  // namespace rvt
