// Host-side launch helpers of the GEMM engines (gemm.hpp, ppgemm.hpp, ppgemm_tn.hpp) shared by capi_conv / capi_linear / capi_lstm.
#pragma once
#include "host.hpp"
#include "gemm.hpp"
#include "ppgemm.hpp"
#include "ppgemm_tn.hpp"

namespace rvt {
// split the token contraction of a weight gradient so that the launch fills the chip
// output-tile width of the weight-gradient kernels
static inline int wgrad_bn(int out_cols) {
    const int forced = tuning().wgrad_bn;                   // tuning knob
    if (forced == 64 || forced == 128) return forced;
    return out_cols <= 64 ? 64 : 128;
}
static inline int wgrad_ksplit(int out_rows, int out_cols, int tokens, int bn) {
    const int split_override = tuning().wgrad_blocks;   // tuning knob
    int tiles = ((out_rows + 127) / 128) * ((out_cols + bn - 1) / bn);
    // as many workgroups as are resident at once: two per CU (measured on dW[512][128], 1.9 M tokens: 0.51 ms at 512
    // workgroups vs 0.75 at 256); but at least 8192 tokens per K slice, or the partial tiles and their reduction
    // cost more than the extra parallelism brings (dW[128][128]: 0.32 ms at 256 slices, 0.42 at 512)
    int want = imax(1, (split_override > 0 ? split_override : 512) / imax(1, tiles));
    const int slice_tokens = imax(64, tuning().wgrad_slice_tokens);   // (tests: small)
    int maxs = imax(1, tokens / slice_tokens);
    // problems whose 8192-token slices do not fill the chip once (RVT-Tiny on Gen1: 104 workgroups at stage 1, 24 at stage 2,
    // each walking hundreds of K tiles - 1.4 TB/s): slices down to 1024 tokens until one workgroup per CU is reached (the
    // partial tiles of such launches are a few MB against >= 100 MB of operands)
    if (maxs * tiles < 256 && slice_tokens > 1024) maxs = imax(maxs, imin(imax(1, tokens / 1024), (256 + tiles - 1) / tiles));
    int ks = imin(want, maxs);
    if (ks >= 16) ks = ks / 8 * 8;             // multiple of 8 slices: tiles of one slice can share an XCD's L2
    return ks;
}
}  // namespace rvt

using namespace rvt;

// Two-stage split-K weight gradient: out[Mg][Ng] += A^T B with the token contraction cut into slices whose partial
// tiles go to `ws` (plain stores) and are folded by splitk_reduce_kernel; the A-side column sums (bias gradient)
// ride along.  ws must hold rvt_wgrad_workspace_floats(...) floats.
static inline size_t wgrad_ws_floats(int Mg, int Ng, int tokens, int bn, int bk, int want_colsum) {
    int ns = gemm_slices(tokens, wgrad_ksplit(Mg, Ng, tokens, bn), bk);
    return (size_t)ns * ((size_t)Mg * Ng + (want_colsum ? Mg : 0));
}
template <class T, int BN, class ASrc, class BSrc, class BXf>
static void launch_wgrad(const ASrc& a, const BSrc& b, const BXf& bxf, float* out, float* colsum_out, float* ws,
                         int Mg, int Ng, int tokens, hipStream_t st, bool transpose_out = false) {
    const int BK = TileGeom<T>::BK;
    const int ks = wgrad_ksplit(Mg, Ng, tokens, BN);
    const int ns = gemm_slices(tokens, ks, BK);
    if (ws == nullptr) {                       // no workspace: direct atomics (correct, slow on large split counts)
        EpAtomicF32 ep{out, Ng};
        launch_gemm<T, BN, true>(a, XfNone(), b, bxf, ep, Mg, Ng, tokens, ks, st, nullptr);
        return;
    }
    const size_t tile_elems = (size_t)Mg * Ng;
    float* ws_cs = colsum_out ? ws + (size_t)ns * tile_elems : nullptr;
    EpPartialStore ep{ws, Ng, tile_elems, 0};
    launch_gemm<T, BN, true>(a, XfNone(), b, bxf, ep, Mg, Ng, tokens, ks, st, ws_cs);
    FoldJobs fj;                               // the weight tile and the bias column sums: one fold launch
    fj.add(ws, out, ns, tile_elems, tile_elems, transpose_out ? Ng : 0);
    if (colsum_out) fj.add(ws_cs, colsum_out, ns, (size_t)Mg, (size_t)Mg);
    launch_fold_jobs(fj, st);
}

#define DISPATCH_WGRAD_BN(N, ...)                                    \
    do {                                                             \
        if (wgrad_bn(N) == 64) { constexpr int BN = 64; __VA_ARGS__; } \
        else { constexpr int BN = 128; __VA_ARGS__; }                \
    } while (0)

// Route of a bf16 "row, k" x "row, k" product: the 256 x 256 LDS-DMA ping-pong kernel (ppgemm.hpp) where its tile shape
// divides the problem and there are enough rows to fill it, else the 128-row register-staged engine (gemm.hpp).
// RVT_PPGEMM=0 disables (A/B measurements); RVT_PPGEMM_MIN_M lowers the row threshold (tests).
// contraction length from which an epilogue flavour goes to ppgemm (RVT_PPGEMM_ALL=1: always - the parity tests)
static inline int pp_min_k(int k) {
    const int all = tuning().ppgemm_all;
    return all ? 0 : k;
}
static inline bool use_ppgemm_tn(int dtype, int M, int N, int K, int ldy, int ldx, int kcut) {
    const int enabled = tuning().ppgemm;
    return enabled && dtype == RVT_BF16 && ppgemm_tn_shape_ok(M, N, K, ldy, ldx, kcut);
}
static inline bool use_ppgemm(int dtype, int M, int N, int K, int ldx, int ldw, int kcut) {
    const int enabled = tuning().ppgemm;
    const int min_m = tuning().ppgemm_min_m;
    return enabled && dtype == RVT_BF16 && M >= min_m && ppgemm_shape_ok(M, N, K, ldx, ldw, kcut);
}
