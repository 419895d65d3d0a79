// extern "C" entry points, part 5 of 8: fused MLP halves (mlp.hpp, mlp_chain.hpp) and the LayerNorm-backward GEMM (dgrad_ln.hpp).
#include "host.hpp"
#include "gemm.hpp"
#include "dgrad_ln.hpp"
#include "mlp.hpp"
#include "mlp_chain.hpp"
#include "mlp_stream.hpp"
#include "ln_linear.hpp"

using namespace rvt;

extern "C" {
// ------------------------------------------------------------------------------------------ fused MLP
int rvt_mlp_fused_supported(int dtype, int C) {
    if (dtype == RVT_BF16) return C == 64 || C == 128;
    if (dtype == RVT_F32) return C == 64;
    return 0;
}

// tile height and resident workgroups per CU of the fused MLP kernels (LDS: ~41 KiB at bf16 C=64 TM=64, ~57 KiB at C=128)
static int mlp_tm(int dtype, int C) {
    const int tm_override = tuning().mlp_tm;     // tuning knob (bf16 C=64 only)
    if (dtype == RVT_BF16 && C == 64 && tm_override == 128) return 128;
    return 64;
}
}  // extern "C"
template <class K> static int mlp_grid(K kernel, int M, int tm) {
    const int resident_override = tuning().gemm_resident;
    const int n_tiles = (M + tm - 1) / tm;
    const int per_cu = resident_per_cu(kernel, 256, 2);
    return imax(1, imin(n_tiles, resident_override > 0 ? resident_override : 256 * per_cu));
}
// register-chained MLP kernels (csrc/mlp_chain.hpp): C == 64; tuning.mlp_chain = 0 falls back to the LDS-staged kernels of mlp.hpp
static bool mlp_chain_on(int dtype, int C) {
    const int off = !tuning().mlp_chain;
    return !off && C == 64 && (dtype == RVT_BF16 || dtype == RVT_F32);
}
// streamed-weight chain kernels (csrc/mlp_stream.hpp): C == 128; tuning.mlp_stream = 0 falls back to mlp.hpp / the op-by-op backward
static bool mlp_stream_on(int dtype, int C) {
    return tuning().mlp_stream != 0 && tuning().mlp_chain != 0 && C == 128 && (dtype == RVT_BF16 || dtype == RVT_F32);
}
template <class T> struct MsWaves { static constexpr int V = sizeof(T) == 2 ? 8 : 4, MINW = sizeof(T) == 2 ? 2 : 1; };
// one workgroup per CU (two 32-KiB weight stages + tables); every wave of a workgroup walks the same number of tiles
template <class K> static int ms_grid(K kernel, int threads, int M, int wpb) {
    const int resident_override = tuning().chain_resident;
    const int per_cu = resident_per_cu(kernel, threads, 1);
    const int want = ((M + 31) / 32 + wpb - 1) / wpb;
    return imax(1, imin(want, resident_override > 0 ? resident_override : 256 * per_cu));
}
#ifndef MC_FWD_WPB
#define MC_FWD_WPB 8      // (6 waves x 3 per SIMD at <= 168 registers spills inside the chunk loop: 2.0 ms against 1.42)
#define MC_FWD_MINW 2
#endif
template <class T> struct McWaves { static constexpr int V = sizeof(T) == 2 ? 8 : 4; };
template <class K> static int mc_grid(K kernel, int threads, int M, int wpb) {
    const int resident_override = tuning().chain_resident;
    const int per_cu = resident_per_cu(kernel, threads, 1);
    const int want = ((M + 31) / 32 + wpb - 1) / wpb;
    return imax(1, imin(want, resident_override > 0 ? resident_override : 256 * per_cu));
}
extern "C" {

// dx = add + LN'(dy W; x) in one launch (csrc/dgrad_ln.hpp): bf16, C in {64, 128}, K = 3C or 4C.  tuning.dgrad_ln = 0 disables.
int rvt_linear_dgrad_ln_supported(int dtype, int C, int K) {
    const int on = tuning().dgrad_ln;
    return on && dtype == RVT_BF16 && (C == 64 || C == 128) && (K == 3 * C || K == 4 * C);
}
int rvt_linear_dgrad_ln(const void* dy, const void* w, const void* x, const void* add, void* dx, const float* ln_w,
                        float* dln_w, float* dln_b, int dtype, int M, int C, int K, float eps, void* stream) {
    RVT_CHECK(rvt_linear_dgrad_ln_supported(dtype, C, K), "linear_dgrad_ln: not built for dtype=%d C=%d K=%d", dtype, C, K);
    RVT_CHECK(M >= 1 && ln_w != nullptr && dln_w != nullptr && dln_b != nullptr, "linear_dgrad_ln: LayerNorm weight and gradient buffers required");
    hipStream_t st = (hipStream_t)stream;
#define RVT_DGL(CC, AH)                                                                                                    \
    do {                                                                                                                   \
        auto k = dgrad_ln_kernel<bf16, CC, 8, AH>;                                                                         \
        hipLaunchKernelGGL(k, dim3(mc_grid(k, 512, M, 8)), dim3(512), 0, st, (const bf16*)dy, (const bf16*)w, (const bf16*)x, \
                           (const bf16*)add, (bf16*)dx, ln_w, dln_w, dln_b, M, K, eps);                                    \
    } while (0)
    const bool four = (K / 32) % 4 == 0;
    if (C == 128) RVT_DGL(128, 4);          // (K = 384 / 512: both whole groups of four chunks)
    else { if (four) RVT_DGL(64, 4); else RVT_DGL(64, 2); }
#undef RVT_DGL
    return check_launch("linear_dgrad_ln");
}

// dy0 = LN'(dy W + add; y0): the qkv input gradient of a stage's first block (no norm1) carried through the down-sampling norm in front of it
int rvt_linear_dgrad_preln(const void* dy, const void* w, const void* y0, const void* add, void* dy0, const float* ln_w,
                           float* dln_w, float* dln_b, int dtype, int M, int C, int K, float eps, void* stream) {
    RVT_CHECK(rvt_linear_dgrad_ln_supported(dtype, C, K), "linear_dgrad_preln: not built for dtype=%d C=%d K=%d", dtype, C, K);
    RVT_CHECK(M >= 1 && add != nullptr && ln_w != nullptr && dln_w != nullptr && dln_b != nullptr, "linear_dgrad_preln: add, LayerNorm weight and gradient buffers required");
    hipStream_t st = (hipStream_t)stream;
#define RVT_DGLI(CC, AH)                                                                                                   \
    do {                                                                                                                   \
        auto k = dgrad_ln_kernel<bf16, CC, 8, AH, true>;                                                                   \
        hipLaunchKernelGGL(k, dim3(mc_grid(k, 512, M, 8)), dim3(512), 0, st, (const bf16*)dy, (const bf16*)w, (const bf16*)y0, \
                           (const bf16*)add, (bf16*)dy0, ln_w, dln_w, dln_b, M, K, eps);                                   \
    } while (0)
    const bool four = (K / 32) % 4 == 0;
    if (C == 128) RVT_DGLI(128, 4);
    else { if (four) RVT_DGLI(64, 4); else RVT_DGLI(64, 2); }
#undef RVT_DGLI
    return check_launch("linear_dgrad_preln");
}

// u = LN(x), y = u W^T + bias in one launch (csrc/ln_linear.hpp): the qkv projection of a C = 128 block.  tuning.ln_linear = 0 disables.
int rvt_ln_linear_supported(int dtype, int C, int N) {
    return tuning().ln_linear != 0 && dtype == RVT_BF16 && C == 128 && N == 384;
}
int rvt_ln_linear_fwd(const void* x, const float* ln_w, const float* ln_b, const void* w, const float* bias, void* u, void* y,
                      int dtype, int M, int C, int N, float eps, void* stream) {
    RVT_CHECK(rvt_ln_linear_supported(dtype, C, N), "ln_linear_fwd: not built for dtype=%d C=%d N=%d", dtype, C, N);
    RVT_CHECK(M >= 1 && (ln_w == nullptr) == (ln_b == nullptr) && y != nullptr, "ln_linear_fwd: LayerNorm weight and bias go together; the output is required");
    auto k = lnlin_fwd_kernel<bf16, 128, 384, 8, 2>;
    hipLaunchKernelGGL(k, dim3(mc_grid(k, 512, M, 8)), dim3(512), 0, (hipStream_t)stream, (const bf16*)x, ln_w, ln_b, (const bf16*)w, bias,
                       (bf16*)u, (bf16*)y, M, eps);
    return check_launch("ln_linear_fwd");
}

int rvt_mlp_fwd(const void* xmid, void* xout, void* g_out, void* gp_out, void* v2_out, const float* ln_w, const float* ln_b,
                const void* w1, const float* b1, const void* w2, const float* b2, const float* gamma, int dtype, int M,
                int C, float eps, void* stream) {
    RVT_CHECK(gp_out == nullptr || g_out != nullptr, "mlp_fwd: gp_out needs g_out (g_out alone = the pre-activation h)");
    hipStream_t st = (hipStream_t)stream;
    if (g_out == nullptr && v2_out == nullptr && mlp_stream_on(dtype, C)) {
        // nothing to save, C = 128: the chain kernel with streamed weights (csrc/mlp_stream.hpp)
        DISPATCH_DTYPE(dtype, {
            constexpr int WPB = MsWaves<T>::V;
            auto k = mlps_fwd_kernel<T, 128, WPB, MsWaves<T>::MINW>;
            hipLaunchKernelGGL(k, dim3(ms_grid(k, 64 * WPB, M, WPB)), dim3(64 * WPB), 0, st, (const T*)xmid, (T*)xout, ln_w, ln_b,
                               (const T*)w1, b1, (const T*)w2, b2, gamma, M, eps);
        });
        return check_launch("mlp_fwd(stream)");
    }
    RVT_CHECK(rvt_mlp_fused_supported(dtype, C), "mlp_fwd: fused MLP not built for dtype=%d C=%d", dtype, C);
    if (g_out == nullptr && v2_out == nullptr && mlp_chain_on(dtype, C)) {
        // nothing to save: the register-chained kernel (csrc/mlp_chain.hpp)
        DISPATCH_DTYPE(dtype, {
            constexpr int WPB = sizeof(T) == 2 ? MC_FWD_WPB : 4;
            auto k = mlpc_fwd_kernel<T, 64, WPB, (sizeof(T) == 2 ? MC_FWD_MINW : 1)>;
            hipLaunchKernelGGL(k, dim3(mc_grid(k, 64 * WPB, M, WPB)), dim3(64 * WPB), 0, st, (const T*)xmid, (T*)xout, ln_w, ln_b,
                               (const T*)w1, b1, (const T*)w2, b2, gamma, M, eps);
        });
        return check_launch("mlp_fwd(chain)");
    }
    const int tm = mlp_tm(dtype, C);
#define RVT_MLP_FWD(TT, CC, TMM)                                                                                           \
    hipLaunchKernelGGL((mlp_fwd_kernel<TT, CC, TMM>), dim3(mlp_grid(mlp_fwd_kernel<TT, CC, TMM>, M, tm)), dim3(256), 0, st, \
                       (const TT*)xmid, (TT*)xout, (TT*)g_out, (TT*)gp_out, (TT*)v2_out, ln_w, ln_b, (const TT*)w1, b1,    \
                       (const TT*)w2,                                                                                       \
                       b2, gamma, M, eps)
    if (dtype == RVT_BF16 && C == 64 && tm == 128) RVT_MLP_FWD(bf16, 64, 128);
    else if (dtype == RVT_BF16 && C == 64) RVT_MLP_FWD(bf16, 64, 64);
    else if (dtype == RVT_BF16 && C == 128) RVT_MLP_FWD(bf16, 128, 64);
    else RVT_MLP_FWD(float, 64, 64);
#undef RVT_MLP_FWD
    return check_launch("mlp_fwd");
}

int rvt_mlp_bwd_dgrad(const void* dxout, const void* gp, const void* xmid, void* dh, void* dxmid, const float* ln_w,
                      const void* w2g_t, const void* w1_t, float* dln_w, float* dln_b, int dtype, int M, int C, float eps,
                      void* stream) {
    RVT_CHECK(rvt_mlp_fused_supported(dtype, C), "mlp_bwd_dgrad: fused MLP not built for dtype=%d C=%d", dtype, C);
    hipStream_t st = (hipStream_t)stream;
    const int tm = mlp_tm(dtype, C);
#define RVT_MLP_BWD(TT, CC, TMM)                                                                                          \
    hipLaunchKernelGGL((mlp_bwd_dgrad_kernel<TT, CC, TMM>), dim3(mlp_grid(mlp_bwd_dgrad_kernel<TT, CC, TMM>, M, tm)),       \
                       dim3(256), 0, st, (const TT*)dxout, (const TT*)gp, (const TT*)xmid, (TT*)dh, (TT*)dxmid, ln_w,      \
                       (const TT*)w2g_t, (const TT*)w1_t, dln_w, dln_b, M, eps)
    if (dtype == RVT_BF16 && C == 64 && tm == 128) RVT_MLP_BWD(bf16, 64, 128);
    else if (dtype == RVT_BF16 && C == 64) RVT_MLP_BWD(bf16, 64, 64);
    else if (dtype == RVT_BF16 && C == 128) RVT_MLP_BWD(bf16, 128, 64);
    else RVT_MLP_BWD(float, 64, 64);
#undef RVT_MLP_BWD
    return check_launch("mlp_bwd_dgrad");
}

// Everything-on-chip backward of the MLP half (csrc/mlp.hpp, mlp_bwd_fused_kernel): input gradient, LayerNorm backward
// and the weight gradients from (dxout, xmid) alone.  Built where the whole set of weight-gradient accumulators fits
// the register file of one workgroup: C == 64.
int rvt_mlp_bwd_fused_supported(int dtype, int C) {
    if (C == 128) return mlp_stream_on(dtype, C);      // streamed-weight kernels (csrc/mlp_stream.hpp); fp32 (parity twin): weight gradients op by op
    return (dtype == RVT_BF16 || dtype == RVT_F32) && C == 64;
}
}  // extern "C"
template <class T, int MODE> static int mlp_bwd_fused_grid(int M) {
    const int g = mlp_grid(mlp_bwd_fused_kernel<T, 64, MODE>, M, 64);
    return MODE == 2 ? imax(1, g / 2) : g;            // MODE 2 launches two chunk groups (grid.y) per tile column
}
// fold the per-workgroup partial records of a weight-gradient launch (plain stores; device-scope float atomics execute
// memory-side on this part) into the fp32 outputs
static void mlp_fold_partials(const float* ws, int grid, int C, float* dw1, float* db1, float* s2, float* cs2, hipStream_t st) {
    const size_t wc = (size_t)4 * C * C;
    const float* p = ws;
    FoldJobs fj;                               // (one launch for the four outputs: they were four)
    fj.add(p, dw1, grid, wc, wc);
    p += (size_t)grid * wc;
    fj.add(p, s2, grid, wc, wc);
    p += (size_t)grid * wc;
    fj.add(p, db1, 2 * grid, (size_t)4 * C, (size_t)4 * C);
    p += (size_t)2 * grid * 4 * C;
    fj.add(p, cs2, grid, (size_t)C, (size_t)C);
    launch_fold_jobs(fj, st);
}
// tile streams of mlps_bwd_wgrad_kernel: two workgroups (hidden halves) per stream, one workgroup per CU
static int msw_streams(int M) {
    const int n_tiles = (M + 31) / 32;
    return imax(1, imin(2 * n_tiles, one_per_cu_grid(2 * n_tiles)) / 2);
}
extern "C" {
size_t rvt_mlp_bwd_fused_ws_floats(int dtype, int C, int M) {
    if (!rvt_mlp_bwd_fused_supported(dtype, C)) return 0;
    if (C == 128 && dtype == RVT_BF16) return (size_t)msw_streams(M) * ((size_t)2 * 4 * C * C + 2 * 4 * C + C);
    if (C == 128) {       // fp32: LN2 output + GELU + GELU' + dh recomputed into the workspace, then the split-K scratch of the larger weight gradient
        const size_t a = rvt_wgrad_workspace_floats(dtype, C, 4 * C, M, 1), b = rvt_wgrad_workspace_floats(dtype, 4 * C, C, M, 1);
        return (size_t)13 * M * C + (a > b ? a : b);
    }
    size_t grid = dtype == RVT_BF16 ? mlp_bwd_fused_grid<bf16, 2>(M) : mlp_bwd_fused_grid<float, 2>(M);
    if (grid < 256) grid = 256;                          // mlpc_bwd_wgrad_kernel: one workgroup per CU
    return grid * ((size_t)2 * 4 * C * C + 2 * 4 * C + C);
}

int rvt_mlp_bwd_recompute_dgrad(const void* dxout, const void* xmid, void* dxmid, const float* ln_w, const float* ln_b,
                                const void* w1, const float* b1, const void* w2g_t, const void* w1_t, float* dln_w,
                                float* dln_b, int dtype, int M, int C, float eps, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (mlp_stream_on(dtype, C)) {
        // C = 128: the chain kernel with streamed weights (csrc/mlp_stream.hpp); w1_t unused (W1^T fragments = transposing reads of W1)
        DISPATCH_DTYPE(dtype, {
            constexpr int WPB = MsWaves<T>::V;
            auto k = mlps_bwd_dgrad_kernel<T, 128, WPB, MsWaves<T>::MINW>;
            hipLaunchKernelGGL(k, dim3(ms_grid(k, 64 * WPB, M, WPB)), dim3(64 * WPB), 0, st, (const T*)dxout, (const T*)xmid,
                               (T*)dxmid, ln_w, ln_b, (const T*)w1, b1, (const T*)w2g_t, dln_w, dln_b, M, eps);
        });
        return check_launch("mlp_bwd_recompute_dgrad(stream)");
    }
    RVT_CHECK(rvt_mlp_bwd_fused_supported(dtype, C), "mlp_bwd_recompute_dgrad: not built for dtype=%d C=%d", dtype, C);
    if (mlp_chain_on(dtype, C)) {
        DISPATCH_DTYPE(dtype, {
            constexpr int WPB = McWaves<T>::V;
            auto k = mlpc_bwd_dgrad_kernel<T, 64, WPB>;
            hipLaunchKernelGGL(k, dim3(mc_grid(k, 64 * WPB, M, WPB)), dim3(64 * WPB), 0, st, (const T*)dxout, (const T*)xmid,
                               (T*)dxmid, ln_w, ln_b, (const T*)w1, b1, (const T*)w2g_t, dln_w, dln_b, M, eps);
        });
        return check_launch("mlp_bwd_recompute_dgrad(chain)");
    }
    DISPATCH_DTYPE(dtype, {
        const int grid = mlp_bwd_fused_grid<T, 1>(M);
        hipLaunchKernelGGL((mlp_bwd_fused_kernel<T, 64, 1>), dim3(grid), dim3(256), 0, st, (const T*)dxout, (const T*)xmid, (T*)dxmid,
                           ln_w, ln_b, (const T*)w1, b1, (const T*)w2g_t, (const T*)w1_t, dln_w, dln_b, (float*)nullptr, M, eps);
    });
    return check_launch("mlp_bwd_recompute_dgrad");
}

int rvt_mlp_bwd_recompute_wgrad(const void* dxout, const void* xmid, const float* ln_w, const float* ln_b, const void* w1,
                                const float* b1, const void* w2g_t, float* dw1, float* db1, float* s2, float* cs2, float* ws,
                                int dtype, int M, int C, float eps, void* stream) {
    RVT_CHECK(rvt_mlp_bwd_fused_supported(dtype, C), "mlp_bwd_recompute_wgrad: not built for dtype=%d C=%d", dtype, C);
    RVT_CHECK(ws != nullptr && M >= 1, "mlp_bwd_recompute_wgrad: workspace required");
    hipStream_t st = (hipStream_t)stream;
    if (C == 128 && dtype == RVT_F32) {
        // fp32 parity twin of the streamed route: the forward and the input-gradient kernel are the fp32 instantiations of
        // mlp_stream.hpp; the weight gradients recompute LN2 / fc1 / GELU / GELU' / dh with the op-by-op entry points (the
        // weight-stationary kernel is bf16-only: transposing LDS reads, 2-byte tiles)
        const size_t MC = (size_t)M * C;
        float* v2 = ws; float* g = v2 + MC; float* gp = g + 4 * MC; float* dhd = gp + 4 * MC; float* wsk = dhd + 4 * MC;
        if (rvt_layernorm_fwd(xmid, ln_w, ln_b, v2, dtype, M, C, eps, stream)) return 1;
        if (rvt_linear_gelu_fwd(v2, w1, b1, g, gp, dtype, M, 4 * C, C, stream)) return 1;
        if (rvt_linear_dgrad(dxout, w2g_t, nullptr, nullptr, gp, dhd, dtype, M, C, 4 * C, stream)) return 1;
        if (rvt_linear_wgrad(dxout, g, s2, cs2, wsk, dtype, M, C, 4 * C, 0, stream)) return 1;
        return rvt_linear_wgrad(dhd, v2, dw1, db1, wsk, dtype, M, 4 * C, C, 0, stream);
    }
    if (C == 128) {
        const int S = msw_streams(M);
        hipLaunchKernelGGL(mlps_bwd_wgrad_kernel<0>, dim3(16 * ((S + 7) / 8)), dim3(512), 0, st, (const bf16*)dxout, (const bf16*)xmid, ln_w, ln_b,
                           (const bf16*)w1, b1, (const bf16*)w2g_t, ws, M, eps, S);
        mlp_fold_partials(ws, S, C, dw1, db1, s2, cs2, st);
        return check_launch("mlp_bwd_recompute_wgrad(stream)");
    }
    int grid = 0;
    const int chain_wgrad = tuning().mlp_chain_wgrad;
    if (chain_wgrad && dtype == RVT_BF16 && mlp_chain_on(dtype, C)) {
        grid = one_per_cu_grid((M + 31) / 32);          // one workgroup per CU (tests: tuning.one_per_cu_grid)
        hipLaunchKernelGGL(mlpc_bwd_wgrad_kernel<false>, dim3(grid), dim3(512), 0, st, (const bf16*)dxout, (const bf16*)xmid, ln_w, ln_b,
                           (const bf16*)w1, b1, (const bf16*)w2g_t, ws, M, eps, (const bf16*)nullptr, (bf16*)nullptr, (float*)nullptr,
                           (float*)nullptr);
        mlp_fold_partials(ws, grid, C, dw1, db1, s2, cs2, st);
        return check_launch("mlp_bwd_recompute_wgrad(chain)");
    }
    DISPATCH_DTYPE(dtype, {
        grid = mlp_bwd_fused_grid<T, 2>(M);
        hipLaunchKernelGGL((mlp_bwd_fused_kernel<T, 64, 2>), dim3(grid, 2), dim3(256), 0, st, (const T*)dxout, (const T*)xmid,
                           (T*)nullptr, ln_w, ln_b, (const T*)w1, b1, (const T*)w2g_t, (const T*)nullptr, (float*)nullptr,
                           (float*)nullptr, ws, M, eps);
    });
    mlp_fold_partials(ws, grid, C, dw1, db1, s2, cs2, st);
    return check_launch("mlp_bwd_recompute_wgrad");
}

// Both halves of the recompute backward in ONE launch (mlp_chain.hpp, mlpc_bwd_wgrad_kernel<true>): one recompute of fc1 / GELU /
// GELU', one read of xmid / dxout for the weight gradients AND the input gradient.
int rvt_mlp_bwd_both_supported(int dtype, int C) {
    return tuning().route_mlp_bwd_both != 0 && tuning().mlp_chain_wgrad != 0 && dtype == RVT_BF16 && C == 64 && mlp_chain_on(dtype, C);
}
int rvt_mlp_bwd_recompute_both(const void* dxout, const void* xmid, void* dxmid, const float* ln_w, const float* ln_b, const void* w1,
                               const float* b1, const void* w2g_t, const void* w1_t, float* dln_w, float* dln_b, float* dw1, float* db1,
                               float* s2, float* cs2, float* ws, int dtype, int M, int C, float eps, void* stream) {
    RVT_CHECK(rvt_mlp_bwd_both_supported(dtype, C), "mlp_bwd_recompute_both: not built for dtype=%d C=%d (or routed off)", dtype, C);
    RVT_CHECK(ws && dxmid && w1_t && dln_w && dln_b && M >= 1, "mlp_bwd_recompute_both: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int grid = one_per_cu_grid((M + 31) / 32);
    hipLaunchKernelGGL(mlpc_bwd_wgrad_kernel<true>, dim3(grid), dim3(512), 0, st, (const bf16*)dxout, (const bf16*)xmid, ln_w, ln_b,
                       (const bf16*)w1, b1, (const bf16*)w2g_t, ws, M, eps, (const bf16*)w1_t, (bf16*)dxmid, dln_w, dln_b);
    mlp_fold_partials(ws, grid, C, dw1, db1, s2, cs2, st);
    return check_launch("mlp_bwd_recompute_both");
}

}  // extern "C"
