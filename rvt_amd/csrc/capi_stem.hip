// extern "C" entry points, part 4 of 8: the stem on the loader's uint8 planes (stem.hpp).
#include "host.hpp"
#include "stem.hpp"

using namespace rvt;

extern "C" {
// ---- the stem on the uint8 planes (stem.hpp) ----
static const int STEM_FWD_PB = 4;
static int stem_fwd_depth() {                          // software-pipeline depth of the forward's plane loads (k-steps in flight)
    const int d = tuning().stem_depth;
    return d == 5 ? 5 : 4;
}

int rvt_stem_supported(int dtype, int src_u8, int Cin, int Cout, int k, int stride, int pad, int w) {
    const int on = tuning().stem;
    return on && dtype == RVT_BF16 && src_u8 && Cout == STEM_CO && k == STEM_K && stride == STEM_STRIDE && pad == STEM_PAD &&
           Cin >= 1 && Cin * STEM_K <= 2 * STEM_KSP_MAX && Cin * STEM_K <= STEM_WG_ROWS && (w % 4) == 0;
}

int rvt_stem_fwd(const void* src, const void* w, const float* ln_w, const float* ln_b, void* y0, void* x, int dtype, int F,
                 int Cin, int cp, int h, int wd, int H, int W, float eps, void* stream) {
    RVT_CHECK(rvt_stem_supported(dtype, 1, Cin, STEM_CO, STEM_K, STEM_STRIDE, STEM_PAD, wd), "stem_fwd: unsupported shape Cin=%d w=%d", Cin, wd);
    RVT_CHECK(h <= H && wd <= W && cp >= Cin, "stem_fwd: planes %dx%d larger than the model resolution %dx%d", h, wd, H, W);
    StemGeom g;
    g.F = F; g.Cin = Cin; g.cp = cp; g.h = h; g.w = wd;
    g.Ho = (H + 2 * STEM_PAD - STEM_K) / STEM_STRIDE + 1; g.Wo = (W + 2 * STEM_PAD - STEM_K) / STEM_STRIDE + 1;
    const int D = stem_fwd_depth();
    g.NR = Cin * STEM_K; g.KS = (g.NR + 1) / 2; g.KSP = (g.KS + D - 1) / D * D;
    RVT_CHECK(g.KSP <= STEM_KSP_MAX, "stem_fwd: %d k-steps do not fit the LDS", g.KSP);
    g.XS = (g.Wo + 31) / 32; g.OG = (g.Ho + STEM_FWD_PB - 1) / STEM_FWD_PB;
    const int og8 = (g.OG * g.XS + 7) / 8;              // sets of eight (row group, segment) units per frame
    g.n_items = F * og8;
    g.dOG = FastDiv(og8); g.d7 = FastDiv(STEM_K); g.dXS = FastDiv(g.XS);
    const int grid = one_per_cu_grid(g.n_items);
    if (D == 5)
        hipLaunchKernelGGL((stem_fwd_kernel<STEM_FWD_PB, 5>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (const uint8_t*)src,
                           (const bf16*)w, ln_w, ln_b, (bf16*)y0, (bf16*)x, g, eps);
    else
        hipLaunchKernelGGL((stem_fwd_kernel<STEM_FWD_PB, 4>), dim3(grid), dim3(512), 0, (hipStream_t)stream, (const uint8_t*)src,
                           (const bf16*)w, ln_w, ln_b, (bf16*)y0, (bf16*)x, g, eps);
    return check_launch("stem_fwd");
}

size_t rvt_stem_wgrad_ws_floats(int Cin, int F, int H, int W) {
    const int Ho = (H + 2 * STEM_PAD - STEM_K) / STEM_STRIDE + 1, Wo = (W + 2 * STEM_PAD - STEM_K) / STEM_STRIDE + 1;
    const int NJB = (Cin * STEM_K + 3) / 4;
    return (size_t)one_per_cu_grid(F * Ho * ((Wo + 31) / 32)) * (size_t)(NJB * 32) * STEM_CO;
}

static int stem_wgrad_launch(const void* src, const void* dy, const void* y0, const float* ln_w, float* dln_w, float* dln_b, float* dw,
                             float* ws, int dtype, int F, int Cin, int cp, int h, int wd, int H, int W, float eps, void* stream) {
    RVT_CHECK(rvt_stem_supported(dtype, 1, Cin, STEM_CO, STEM_K, STEM_STRIDE, STEM_PAD, wd), "stem_wgrad: unsupported shape Cin=%d w=%d", Cin, wd);
    RVT_CHECK(h <= H && wd <= W && cp >= Cin && ws != nullptr, "stem_wgrad: bad arguments");
    StemWgGeom g;
    g.F = F; g.Cin = Cin; g.h = h; g.w = wd;
    g.Ho = (H + 2 * STEM_PAD - STEM_K) / STEM_STRIDE + 1; g.Wo = (W + 2 * STEM_PAD - STEM_K) / STEM_STRIDE + 1;
    g.NR = Cin * STEM_K; g.NJB = (g.NR + 3) / 4;
    g.XS = (g.Wo + 31) / 32; g.n_tiles = F * g.Ho * g.XS;
    const int grid = one_per_cu_grid(g.n_tiles);
    g.per_wg = (g.n_tiles + grid - 1) / grid;
    g.dXS = FastDiv(g.XS); g.dHo = FastDiv(g.Ho); g.d7 = FastDiv(STEM_K);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(stem_wgrad_kernel<false>, dim3(grid), dim3(512), 0, st, (const uint8_t*)src, (const bf16*)dy, (const bf16*)nullptr,
                           (const float*)nullptr, (float*)nullptr, (float*)nullptr, ws, g, 0.f);
    const int total = STEM_CO * STEM_K * STEM_K * Cin;
    hipLaunchKernelGGL(stem_wgrad_fold_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)ws, dw, grid, Cin, cp, g.NJB);
    return check_launch("stem_wgrad");
}

int rvt_stem_wgrad(const void* src, const void* dy, float* dw, float* ws, int dtype, int F, int Cin, int cp, int h, int wd,
                   int H, int W, void* stream) {
    return stem_wgrad_launch(src, dy, nullptr, nullptr, nullptr, nullptr, dw, ws, dtype, F, Cin, cp, h, wd, H, W, 0.f, stream);
}

}  // extern "C"
