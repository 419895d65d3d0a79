// Weight-gradient (token-contraction) variant of ppgemm.hpp:
//
//     dW[n][k] += sum_m dY[m][n] * X[m][k]            (bias gradient: db[n] += sum_m dY[m][n])
//
// Same machine as the NT kernel - 256 x 256 output tile per 512-thread workgroup, 8 waves as 2 (n halves) x 4 (k quarters), two
// wave groups one barrier apart, an LDS-DMA load stream of 8 KiB units running 1.5 steps ahead behind counted vmcnt waits -
// with the contraction running over TOKENS, 64 per step:
//
//   * both operands are row-major [token][feature]; the MFMA fragments want "row = feature, 8 consecutive tokens per lane".  A
//     load-stream unit is [64 tokens][8 chunks of 16 B] (128 contiguous bytes per token): for the dY side the columns 64 i .. of
//     MFMA row block i of both wave groups (group wr owns the columns 64 i + 32 wr + r), for the X side the 64 columns of wave
//     column g.  The LDS-DMA gathers the chunks (per-lane source
//     address), so the LDS image is free to choose: chunk position ^= 4 * ((token >> 1) & 1) makes the fragment reads - two
//     ds_read_b64_tr_b16 per fragment, each a 4-token x 16-column block per 16 lanes - bank-conflict free;
//   * one (n tile, k tile, token slice) item per workgroup: the token axis is cut into as many slices as fill the chip, every
//     slice stores its fp32 partial tile TRANSPOSED ([k][n], full 512-byte rows through an LDS transpose) into ws[slice][K][N];
//     gemm.hpp's splitk_reduce_kernel folds the slices (and transposes back).  The bias gradient rides along: wave column wc
//     sums the tokens of k-step wc of the dY fragments it has in registers anyway (8 adds per lane and phase), one partial
//     record per (slice, wc, lane half);
//   * tokens beyond M are never loaded (buffer bounds -> zeros).
// Hazard bookkeeping: as ppgemm.hpp (unit order W0..W3 X0..X3 per step, two units per phase, waits 8 / 9 / 10 / 7); there is no
// epilogue inside the pipeline (one item per workgroup), so the counts are the plain ones.
#pragma once
#include "ppgemm.hpp"

namespace rvt {

struct PPTnGeom {
    static constexpr int UNIT = PPGeom::UNIT, STAGE = PPGeom::STAGE, OFF_W = PPGeom::OFF_W;
    static constexpr int T_LD = 528;                   // epilogue: [k 32][n 128] fp32 per wave and k block, rows padded to 528 B
    static constexpr int T_BYTES = 32 * T_LD;
    static constexpr int SMEM = 2 * STAGE > 8 * T_BYTES ? 2 * STAGE : 8 * T_BYTES;
};

// CONV: the X operand is im2col(in) of a k x k / stride / pad convolution over in[F][H][W][Cin] (reference maxvit.py:160-177, the
// down-sampling convs, and the PAFPN convs), never materialised: token m = output pixel (f, oy, ox) in raster order, column
// (tap, ci) = in[f][oy * stride + tap / k - pad][ox * stride + tap % k - pad][ci].  A load-stream unit is 64 columns = one tap and
// a 64-channel slab of it (Cin % 64 == 0), so only the per-lane SOURCE address of the X-side pieces changes: pixel offset of the
// lane's token for the unit's tap, or an out-of-range offset (the buffer returns zeros) for padding pixels and tokens beyond the slice.
struct PPTnConv {
    int H, W, Cin, Ho, Wo, k, stride, pad;
    unsigned in_bytes;                                 // F * H * W * Cin * 2 (< 2^31)
};

template <int ABL = 0, bool CONV = false>
__global__ void __launch_bounds__(512, 2)
ppgemm_tn_kernel(const bf16* __restrict__ dY, int ldy, const bf16* __restrict__ X, const bf16* __restrict__ X2, int kcut, int ldx,
                 float* __restrict__ ws, float* __restrict__ ws_cs, int M, int N, int K, int n_tiles, int k_tiles,
                 int tokens_per_slice, PPTnConv cv) {
    typedef PPTnGeom G;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = wave_uniform(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    // item of this workgroup.  The output tiles of one token slice read the same dY / X rows at the same pace: the items are laid
    // out slice-major and XCD x (workgroup id % 8 - the dispatcher deals workgroups round-robin over the XCDs) takes the x-th
    // run of gridDim / 8 consecutive items, so that a slice's tiles share an L2 (a slice straddles at most two XCDs) whatever the
    // tile count per slice (18 tiles per slice, 14 slices = 252 items: the earlier "slice s on XCD s % 8" dealt 8 slices = 144).
    const int tiles = n_tiles * k_tiles;
    const int item = (blockIdx.x & 7) * ((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    const int slice = item / tiles, tile = item - slice * tiles;
    const int nt = tile % n_tiles, kt = tile / n_tiles;
    if (slice * tokens_per_slice >= M) return;
    const int tok0 = slice * tokens_per_slice;
    const int tok1 = tok0 + tokens_per_slice < M ? tok0 + tokens_per_slice : M;
    const int nsteps = (tok1 - tok0 + 63) / 64;

    // ---- load stream: lane -> (token t = 8 wave + lane / 8 of the unit, chunk position lane % 8) ----
    const int t_l = wave * 8 + (lane >> 3);
    const int cpos = (lane & 7) ^ (((t_l >> 1) & 1) << 2);              // chunk stored at position lane % 8
    // dY side, unit i = columns n0 + 64 i .. + 63 (128 contiguous bytes per token): MFMA row block i of wave group wr' = cpos / 4
    // is the columns 64 i + 32 wr' + (0..31), chunk cc = cpos % 4 of it
    const int vy0 = t_l * ldy * 2 + cpos * 16;                                     // + i * 128 bytes
    // X side, unit g: chunk cpos = columns k0 + g * 64 + 8 cpos
    const int vx0 = t_l * ldx * 2 + cpos * 16;                                     // + g * 128 bytes
    const int lds0 = wave * 1024;
    const bf16* const ybase = dY + (size_t)nt * 256;
    // the X side may be two matrices side by side ([x_t | h_{t-1}] of the ConvLSTM, rnn.py:52): k tiles never straddle the cut
    // (round 6: the cut may also fall INSIDE a k tile, on a 64-column unit boundary - C = 128: [x | h] is one k tile, units 0 - 1 from x,
    // 2 - 3 from h: the units of the second segment take their own descriptor, rx2, whose column 0 is the cut)
    const bf16* const xbase = kt * 256 < kcut ? X + (size_t)kt * 256 : X2 + (size_t)(kt * 256 - kcut);
    const int cut_unit = (kcut > kt * 256 && kcut < kt * 256 + 256) ? (kcut - kt * 256) >> 6 : 4;       // first unit of the second segment (4: none)
    struct Desc { pp_rsrc ry, rx, rx2; int vx[CONV ? 4 : 1]; };
    // CONV: the four X units of this k tile = (tap, 64-channel slab) pairs, fixed per workgroup; (f, oy, ox) of the lane's token of
    // the NEXT step to be described, advanced by 64 tokens per step
    int c_dy[4], c_dx[4], c_off[4], c_f = 0, c_by = 0, c_bx = 0;
    pp_rsrc rx_all = pp_make_rsrc(X, 0u);
    if constexpr (CONV) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int col = kt * 256 + g * 64, tap = col / cv.Cin;
            const int tdy = tap / cv.k;
            c_dx[g] = tap - tdy * cv.k;
            c_off[g] = col < K ? ((tdy * cv.W + c_dx[g]) * cv.Cin + (col - tap * cv.Cin)) * 2 : 0;
            c_dy[g] = col < K ? tdy : 1 << 20;                          // (the last k tile may be partly beyond K: rows of zeros)
        }
        const int idx = tok0 + t_l, q = idx / cv.Wo;
        c_f = q / cv.Ho;
        c_by = (q - c_f * cv.Ho) * cv.stride - cv.pad;               // input row of tap (0, 0)
        c_bx = (idx - q * cv.Wo) * cv.stride - cv.pad;
        rx_all = pp_make_rsrc(X, cv.in_bytes);
    }
    int lstep = 0;
    auto next_desc = [&]() __attribute__((always_inline)) -> Desc {
        Desc d;
        const int t0 = tok0 + lstep * 64;
        if (lstep < nsteps) {
            const int rows = tok1 - t0 < 64 ? tok1 - t0 : 64;
            // the last token row of the buffer ends at its last tile column: row stride ld, tile width 256 columns
            d.ry = pp_make_rsrc(ybase + (size_t)t0 * ldy, (unsigned)(((rows - 1) * ldy + 256) * 2));
            if constexpr (!CONV) {
                d.rx = pp_make_rsrc(xbase + (size_t)t0 * ldx, (unsigned)(((rows - 1) * ldx + (cut_unit < 4 ? 64 * cut_unit : 256)) * 2));
                d.rx2 = cut_unit < 4 ? pp_make_rsrc(X2 + (size_t)t0 * ldx, (unsigned)(((rows - 1) * ldx + 64 * (4 - cut_unit)) * 2)) : d.rx;
            }
        } else {
            d.ry = pp_make_rsrc(dY, 0u);
            if constexpr (!CONV) { d.rx = pp_make_rsrc(X, 0u); d.rx2 = d.rx; }
        }
        if constexpr (CONV) {
            d.rx = rx_all; d.rx2 = rx_all;
            const bool in = lstep < nsteps && t0 + t_l < tok1;
            const int base = ((c_f * cv.H + c_by) * cv.W + c_bx) * cv.Cin * 2 + cpos * 16;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const bool ok = in && (unsigned)(c_by + c_dy[g]) < (unsigned)cv.H && (unsigned)(c_bx + c_dx[g]) < (unsigned)cv.W;
                d.vx[g] = ok ? base + c_off[g] : 0x7ffffff0;
            }
            c_bx += 64 * cv.stride;
            while (c_bx >= cv.Wo * cv.stride - cv.pad) { c_bx -= cv.Wo * cv.stride; c_by += cv.stride; }
            while (c_by >= cv.Ho * cv.stride - cv.pad) { c_by -= cv.Ho * cv.stride; c_f++; }
        }
        lstep++;
        return d;
    };
    auto issue_y = [&](const Desc& d, int stage, int i) __attribute__((always_inline)) {
        if (!(ABL & 1) && !(ABL & 32)) pp_glds16(d.ry, smem, stage * G::STAGE + i * G::UNIT + lds0, vy0 + i * 128, 0);
    };
    auto issue_x = [&](const Desc& d, int stage, int g) __attribute__((always_inline)) {
        int v = vx0 + g * 128;
        if constexpr (CONV) v = d.vx[g];
        const bool second = !CONV && g >= cut_unit;                  // (wave-uniform)
        if (second) v = vx0 + (g - cut_unit) * 128;
        if (!(ABL & 1) && !(ABL & 64)) pp_glds16(second ? d.rx2 : d.rx, smem, stage * G::STAGE + G::OFF_W + g * G::UNIT + lds0, v, 0);
    };

    // ---- fragment reads (transposing): lane -> token 16 ks + 8 hi + (lane % 16) / 4 (+ 4), 4 columns 16 ((lane / 16) % 2) + 4 (lane % 4) ----
    const int tl = 8 * hi + ((lane & 15) >> 2);                          // + 16 ks
    const int fl = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);              // column within the 32-column block
    const int sw = ((tl >> 1) & 1) << 2;                                 // (same for token + 4, + 16 ks)
    // dY unit i: chunk wr * 4 + fl / 8 ; X unit wc: chunk j * 4 + fl / 8
    const int ya0 = tl * 128 + (((wr * 4 + (fl >> 3)) ^ sw) << 4) + (fl & 7) * 2;
    const int xa0 = G::OFF_W + wc * G::UNIT + tl * 128 + (((fl >> 3) ^ sw) << 4) + (fl & 7) * 2;     // block j: chunk ^ 4 -> address ^ 64
    // The transposing reads are issued as inline assembly: behind the builtin hipcc (ROCm 7.2) waits with vmcnt(0) in front of every
    // such read while LDS-DMA is in flight (it cannot tell that the read does not alias a pending DMA write) - the load stream
    // drained in every phase (0.60 ms against 0.21 ms for the same kernel without the reads).  The asm read is invisible to that
    // bookkeeping; its own completion is waited for by hand (tr_wait) in front of the MFMAs.
#ifndef RVT_EMU
    const int smem_lds = (int)(size_t)(__attribute__((address_space(3))) char*)smem;
#endif
    auto frag_tr = [&](int addr, auto off_c) __attribute__((always_inline)) {
        constexpr int OFF = decltype(off_c)::value;
        if (ABL & 2) { bf16x8 v; for (int e = 0; e < 8; e++) v[e] = (bf16)(float)((addr & 1023) + lane); return v; }
#ifdef RVT_EMU
        return frag_from_tr<bf16>(reinterpret_cast<const bf16*>(smem + addr + OFF), reinterpret_cast<const bf16*>(smem + addr + OFF + 512));
#else
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
        u32x2_t lo, hi2;
        const int a = smem_lds + addr;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"(OFF));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi2) : "v"(a), "n"(OFF + 512));
        const u32x4 v = {lo[0], lo[1], hi2[0], hi2[1]};
        bf16x8 f;
        __builtin_memcpy(&f, &v, 16);
        return f;
#endif
    };
    auto tr_wait = [&]() __attribute__((always_inline)) {
#ifndef RVT_EMU
        if (!(ABL & 2)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    };

    f32x16 acc[4][2];
    bf16x8 yf[4], xf[2][4];
    float cs[4] = {0.f, 0.f, 0.f, 0.f};

    // ---- prologue: units 0..11 (step 0 complete, X-side units of step 1) ----
    Desc d1, d2;
    {
        const Desc d0 = next_desc();
#pragma unroll
        for (int g = 0; g < 4; g++) issue_x(d0, 0, g);
#pragma unroll
        for (int i = 0; i < 4; i++) issue_y(d0, 0, i);
        d1 = next_desc();
#pragma unroll
        for (int g = 0; g < 4; g++) issue_x(d1, 1, g);
        d2 = next_desc();
    }
    if (!(ABL & 1)) pp_wait_vm<7>();
    pp_barrier();
    if (wr == 1) pp_barrier();                        // group 1 runs one barrier behind group 0

    int s = 0;
    auto bar = [&]() __attribute__((always_inline)) { if (!(ABL & 8)) pp_barrier(); };
    auto step_body = [&](auto first_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value;
        int st = (s & 1) * G::STAGE;
#ifndef RVT_EMU
        asm volatile("" : "+v"(st));
#endif
        auto phase = [&](int i) __attribute__((always_inline)) {
            const int ya = st + i * G::UNIT + ya0, xa = st + xa0, xb = st + (xa0 ^ 64);
            yf[0] = frag_tr(ya, std::integral_constant<int, 0>());
            yf[1] = frag_tr(ya, std::integral_constant<int, 2048>());
            yf[2] = frag_tr(ya, std::integral_constant<int, 4096>());
            yf[3] = frag_tr(ya, std::integral_constant<int, 6144>());
            if (i == 0) {
                xf[0][0] = frag_tr(xa, std::integral_constant<int, 0>());
                xf[0][1] = frag_tr(xa, std::integral_constant<int, 2048>());
                xf[0][2] = frag_tr(xa, std::integral_constant<int, 4096>());
                xf[0][3] = frag_tr(xa, std::integral_constant<int, 6144>());
                xf[1][0] = frag_tr(xb, std::integral_constant<int, 0>());
                xf[1][1] = frag_tr(xb, std::integral_constant<int, 2048>());
                xf[1][2] = frag_tr(xb, std::integral_constant<int, 4096>());
                xf[1][3] = frag_tr(xb, std::integral_constant<int, 6144>());
            }
            bar();
            tr_wait();
            pp_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                if (ABL & 4) { if (FIRST && ks == 0) { acc_zero(acc[i][0]); acc_zero(acc[i][1]); } acc[i][0][ks] += (float)yf[ks][0] * (float)xf[0][ks][1] + (float)xf[1][ks][2]; continue; }
                if (FIRST && ks == 0) { mma32_zero(acc[i][0], yf[ks], xf[0][ks]); mma32_zero(acc[i][1], yf[ks], xf[1][ks]); }
                else { mma32(acc[i][0], yf[ks], xf[0][ks]); mma32(acc[i][1], yf[ks], xf[1][ks]); }
            }
            pp_setprio(0);
            // bias gradient: this wave column's share of the tokens (k-step wc) of the dY fragments
#pragma unroll
            for (int ks = 0; ks < 4; ks++)
                if (ks == wc) {
#pragma unroll
                    for (int e = 0; e < 8; e++) cs[i] += (float)yf[ks][e];
                }
            bar();
        };
        issue_y(d1, (s + 1) & 1, 0); issue_y(d1, (s + 1) & 1, 1);
        if (!(ABL & 1)) { if (ABL & 96) pp_wait_vm<2>(); else pp_wait_vm<8>(); }
        phase(0);
        issue_y(d1, (s + 1) & 1, 2); issue_y(d1, (s + 1) & 1, 3);
        if (!(ABL & 1)) { if (ABL & 96) pp_wait_vm<2>(); else pp_wait_vm<9>(); }
        phase(1);
        issue_x(d2, s & 1, 0); issue_x(d2, s & 1, 1);
        if (!(ABL & 1)) { if (ABL & 96) pp_wait_vm<2>(); else pp_wait_vm<10>(); }
        phase(2);
        issue_x(d2, s & 1, 2); issue_x(d2, s & 1, 3);
        if (!(ABL & 1)) { if (ABL & 96) pp_wait_vm<2>(); else pp_wait_vm<7>(); }
        phase(3);
        d1 = d2;
        d2 = next_desc();
        s++;
    };
    step_body(std::true_type());
    for (int it = 1; it < nsteps; it++) step_body(std::false_type());
    if (wr == 0) pp_barrier();                        // pairs with group 1's extra barrier
    pp_wait_vm<0>();                                  // (trailing empty-buffer pieces still write LDS)
    pp_barrier();                                     // every wave is done with the operand stages: reuse them for the transpose

    // ---- epilogue: accumulator block (rows n = 8 g + 4 hi + w, column k = lane % 32) -> LDS [k][n] -> 512-byte rows of ws[slice][K][N] ----
    char* const tb = smem + wave * G::T_BYTES;
    float* const wout = ws + (size_t)slice * K * N + (size_t)(kt * 256 + wc * 64) * N + nt * 256 + wr * 32;
    const bool k_in = kt * 256 + wc * 64 < K;         // (CONV: K is a multiple of 64, not of 256)
#pragma unroll
    for (int j = 0; j < 2; j++) {
        if (!(ABL & 16) && k_in) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int g = 0; g < 4; g++)
                    *reinterpret_cast<f32x4*>(tb + l31 * G::T_LD + (i * 32 + 8 * g + 4 * hi) * 4) =
                        f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            pp_wave_sync();
#pragma unroll
            for (int it = 0; it < 16; it++) {
                const int kr = it * 2 + hi;           // k row of the block, 32 lanes x 16 B = its 128 n values
                const f32x4 v = *reinterpret_cast<const f32x4*>(tb + kr * G::T_LD + l31 * 16);      // n block l31 / 8, 4 n of it
                *reinterpret_cast<f32x4*>(wout + (size_t)(j * 32 + kr) * N + (l31 >> 3) * 64 + (l31 & 7) * 4) = v;
            }
            pp_wave_sync();
        } else {
#ifndef RVT_EMU
            asm volatile("" ::"v"(acc[0][j]), "v"(acc[1][j]), "v"(acc[2][j]), "v"(acc[3][j]));
#endif
        }
    }
    if (ws_cs != nullptr && kt == 0) {
        float* const co = ws_cs + (size_t)((slice * 4 + wc) * 2 + hi) * N + nt * 256 + wr * 32 + l31;
#pragma unroll
        for (int i = 0; i < 4; i++) co[i * 64] = cs[i];
    }
}

// host side ------------------------------------------------------------------------------------------------------------
inline bool ppgemm_tn_shape_ok(int M, int N, int K, int ldy, int ldx, int kcut) {
    const int min_m = g_tuning.ppgemm_min_m;
    return N % 256 == 0 && K % 256 == 0 && kcut % 64 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && M >= min_m &&
           (size_t)64 * (size_t)(ldy > ldx ? ldy : ldx) * 2 + 512 < (1ull << 31);
}
// token slices: as many as fill the chip once (one item per CU), at least 4 steps of 64 tokens each
inline int ppgemm_tn_slices(int M, int N, int K) {
    const int items_override = g_tuning.ppgemm_tn_items;     // (tests: small)
    const int tiles = (N / 256) * (K / 256);
    int ns = (items_override > 0 ? items_override : 256) / tiles;
    if (ns < 1) ns = 1;
    const int max_ns = (M + 255) / 256;
    return ns > max_ns ? max_ns : ns;
}
inline size_t ppgemm_tn_ws_floats(int M, int N, int K, int want_colsum) {
    const size_t ns = (size_t)ppgemm_tn_slices(M, N, K);
    return ns * (size_t)N * K + (want_colsum ? ns * 8 * (size_t)N : 0);
}
// out[N][K] += dY^T X, colsum_out[N] += column sums of dY (nullable); ws: ppgemm_tn_ws_floats floats
inline void launch_ppgemm_tn(const bf16* dY, int ldy, const bf16* X, const bf16* X2, int kcut, int ldx, float* out, float* colsum_out,
                             float* ws, int M, int N, int K, hipStream_t st) {
    const int ns = ppgemm_tn_slices(M, N, K);
    const int tps = (((M + ns - 1) / ns) + 63) / 64 * 64;
    const int ns_eff = (M + tps - 1) / tps;
    const int n_tiles = N / 256, k_tiles = K / 256;
    float* ws_cs = colsum_out ? ws + (size_t)ns * N * K : nullptr;
    hipLaunchKernelGGL((ppgemm_tn_kernel<0>), dim3(8 * ((n_tiles * k_tiles * ns_eff + 7) / 8)), dim3(512), 0, st, dY, ldy, X, X2, kcut, ldx, ws, ws_cs, M, N,
                       K, n_tiles, k_tiles, tps, PPTnConv{});
    const size_t elems = (size_t)N * K;
    FoldJobs fj;                               // the weight tile and the bias column sums: one fold launch
    fj.add(ws, out, ns_eff, elems, elems, N);
    if (colsum_out) fj.add(ws_cs, colsum_out, ns_eff * 8, (size_t)N, (size_t)N);
    launch_fold_jobs(fj, st);
}

// conv weight gradient dw[Cout][k*k*Cin] += dy^T im2col(in) on the same kernel (CONV): the shapes that fit
inline bool ppgemm_tn_conv_shape_ok(long long F, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const long long M = F * Ho * Wo;
    return Cin % 64 == 0 && Cout % 256 == 0 && k * k * Cin >= 256 && pad < k && M >= g_tuning.ppgemm_min_m &&
           F * H * W * Cin * 2 < 0x7ffffff0ll && M * Cout * 2 < (1ll << 40) && (size_t)64 * Cout * 2 + 512 < (1ull << 31);
}
inline size_t ppgemm_tn_conv_ws_floats(int M, int N, int K) { return (size_t)ppgemm_tn_slices(M, N, (K + 255) / 256 * 256) * N * K; }
inline void launch_ppgemm_tn_conv(const bf16* dY, const bf16* in, float* out, float* ws, int F, int H, int W, int Cin, int Cout, int k,
                                  int stride, int pad, hipStream_t st) {
    PPTnConv cv;
    cv.H = H; cv.W = W; cv.Cin = Cin; cv.k = k; cv.stride = stride; cv.pad = pad;
    cv.Ho = (H + 2 * pad - k) / stride + 1; cv.Wo = (W + 2 * pad - k) / stride + 1;
    cv.in_bytes = (unsigned)((size_t)F * H * W * Cin * 2);
    const int M = F * cv.Ho * cv.Wo, N = Cout, K = k * k * Cin, k_tiles = (K + 255) / 256;
    const int ns = ppgemm_tn_slices(M, N, k_tiles * 256);
    const int tps = (((M + ns - 1) / ns) + 63) / 64 * 64;
    const int ns_eff = (M + tps - 1) / tps;
    const int n_tiles = N / 256;
    hipLaunchKernelGGL((ppgemm_tn_kernel<0, true>), dim3(8 * ((n_tiles * k_tiles * ns_eff + 7) / 8)), dim3(512), 0, st, dY, N, in, in, 1 << 30, 0, ws,
                       (float*)nullptr, M, N, K, n_tiles, k_tiles, tps, cv);
    const size_t elems = (size_t)N * K;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(reduce_grid(elems)), dim3(256), 0, st, (const float*)ws, out, ns_eff, elems, N);
}

}  // namespace rvt
