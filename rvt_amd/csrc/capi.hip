// extern "C" entry points of librvt_hip.so (declared in include/rvt_hip.h).
// Host-side only: argument checks, source/epilogue descriptors, launch geometry.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <type_traits>

#include "common.hpp"
#include "gemm.hpp"
#include "ppgemm.hpp"
#include "ppgemm_tn.hpp"
#include "dgrad_ln.hpp"
#include "rowops.hpp"
#include "attn.hpp"
#include "attn_block.hpp"
#include "attn_core2.hpp"
#include "mlp.hpp"
#include "mlp_chain.hpp"
#include "events.hpp"
#include "pack.hpp"
#include "lstm_scan.hpp"
#include "stem.hpp"
#include "../../include/rvt_hip.h"

namespace rvt {
static thread_local char g_err[512] = "";
void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_last_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return 1;
    }
    return 0;
}
static inline int pow2_ge(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int grid_for(size_t units, int cap = 2048) {
    size_t g = (units + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > (size_t)cap ? cap : g);
}
// split the token contraction of a weight gradient so that the launch fills the chip
// output-tile width of the weight-gradient kernels
static inline int wgrad_bn(int out_cols) {
    static const int forced = getenv("RVT_WGRAD_BN") ? atoi(getenv("RVT_WGRAD_BN")) : 0;                   // tuning knob
    if (forced == 64 || forced == 128) return forced;
    return out_cols <= 64 ? 64 : 128;
}
static inline int wgrad_ksplit(int out_rows, int out_cols, int tokens, int bn) {
    static const int split_override = getenv("RVT_WGRAD_BLOCKS") ? atoi(getenv("RVT_WGRAD_BLOCKS")) : 0;   // tuning knob
    int tiles = ((out_rows + 127) / 128) * ((out_cols + bn - 1) / bn);
    // as many workgroups as are resident at once: two per CU (measured on dW[512][128], 1.9 M tokens: 0.51 ms at 512
    // workgroups vs 0.75 at 256); but at least 8192 tokens per K slice, or the partial tiles and their reduction
    // cost more than the extra parallelism brings (dW[128][128]: 0.32 ms at 256 slices, 0.42 at 512)
    int want = imax(1, (split_override > 0 ? split_override : 512) / imax(1, tiles));
    static const int slice_tokens = getenv("RVT_WGRAD_SLICE_TOKENS") ? imax(64, atoi(getenv("RVT_WGRAD_SLICE_TOKENS"))) : 8192;   // (tests: small)
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
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(reduce_grid(tile_elems)), dim3(256), 0, st, (const float*)ws, out, ns,
                       tile_elems, transpose_out ? Ng : 0);
    if (colsum_out)
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(reduce_grid((size_t)Mg)), dim3(256), 0, st, (const float*)ws_cs,
                           colsum_out, ns, (size_t)Mg, 0);
}

#define DISPATCH_DTYPE(dtype, ...)                                   \
    do {                                                             \
        if ((dtype) == RVT_F32) { typedef float T; __VA_ARGS__; }    \
        else if ((dtype) == RVT_BF16) { typedef bf16 T; __VA_ARGS__; } \
        else { set_last_error("bad dtype %d", (int)(dtype)); return 1; } \
    } while (0)

#define DISPATCH_BN(N, ...)                                          \
    do {                                                             \
        if ((N) <= 64) { constexpr int BN = 64; __VA_ARGS__; }       \
        else { constexpr int BN = 128; __VA_ARGS__; }                \
    } while (0)

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
    static const int all = getenv("RVT_PPGEMM_ALL") ? atoi(getenv("RVT_PPGEMM_ALL")) : 0;
    return all ? 0 : k;
}
static inline bool use_ppgemm_tn(int dtype, int M, int N, int K, int ldy, int ldx, int kcut) {
    static const int enabled = getenv("RVT_PPGEMM") ? atoi(getenv("RVT_PPGEMM")) : 1;
    return enabled && dtype == RVT_BF16 && ppgemm_tn_shape_ok(M, N, K, ldy, ldx, kcut);
}
static inline bool use_ppgemm(int dtype, int M, int N, int K, int ldx, int ldw, int kcut) {
    static const int enabled = getenv("RVT_PPGEMM") ? atoi(getenv("RVT_PPGEMM")) : 1;
    static const int min_m = getenv("RVT_PPGEMM_MIN_M") ? atoi(getenv("RVT_PPGEMM_MIN_M")) : 4096;
    return enabled && dtype == RVT_BF16 && M >= min_m && ppgemm_shape_ok(M, N, K, ldx, ldw, kcut);
}

extern "C" {

const char* rvt_last_error(void) { return g_err; }

int rvt_is_emulator(void) {
#ifdef RVT_EMU
    return 1;
#else
    return 0;
#endif
}

size_t rvt_wgrad_workspace_floats(int dtype, int out_rows, int out_cols, int tokens, int want_colsum) {
    size_t pp = 0;
    if (use_ppgemm_tn(dtype, tokens, out_rows, out_cols, out_rows, out_cols, out_cols))
        pp = ppgemm_tn_ws_floats(tokens, out_rows, out_cols, want_colsum);     // (an upper bound is all the callers need)
    int bn = wgrad_bn(out_cols);
    int bk = dtype == RVT_F32 ? TileGeom<float>::BK : TileGeom<bf16>::BK;
    size_t n = wgrad_ws_floats(out_rows, out_cols, tokens, bn, bk, want_colsum);
    if (out_rows <= 64) {                    // rvt_conv_wgrad may compute the transposed product (see there)
        size_t nt = wgrad_ws_floats(out_cols, out_rows, tokens, 64, bk, want_colsum);
        if (nt > n) n = nt;
    }
    return n > pp ? n : pp;
}

int rvt_prepack_input(const void* src, int src_u8, void* dst, int dtype, int F, int Cin, int h, int w, int H, int W,
                      int Cp, void* stream) {
    RVT_CHECK(Cp % 8 == 0 && Cp >= Cin && H >= h && W >= w, "prepack: bad shape Cp=%d Cin=%d", Cp, Cin);
    hipStream_t st = (hipStream_t)stream;
    RVT_CHECK(Cin <= 32, "prepack: Cin=%d > 32 staged channels", Cin);
    const int seg = src_u8 ? PrepackSeg<unsigned char>::value : PrepackSeg<float>::value;
    size_t items = (size_t)F * H * ((W + seg - 1) / seg);
    int grid = (int)(items < 16384 ? (items < 1 ? 1 : items) : 16384);
    DISPATCH_DTYPE(dtype, {
        if (src_u8)
            hipLaunchKernelGGL((prepack_kernel<T, unsigned char>), dim3(grid), dim3(256), 0, st,
                               (const unsigned char*)src, (T*)dst, F, Cin, h, w, H, W, Cp);
        else
            hipLaunchKernelGGL((prepack_kernel<T, float>), dim3(grid), dim3(256), 0, st, (const float*)src, (T*)dst, F,
                               Cin, h, w, H, W, Cp);
    });
    return check_launch("prepack");
}

}  // extern "C"
template <class T>
static Im2colSrc<T> make_im2col(const void* in, int F, int H, int W, int Cin, int k, int stride, int pad) {
    Im2colSrc<T> s;
    s.p = (const T*)in; s.H = H; s.W = W; s.Cin = Cin;
    s.Ho = (H + 2 * pad - k) / stride + 1; s.Wo = (W + 2 * pad - k) / stride + 1;
    s.kw = k; s.stride = stride; s.pad = pad;
    s.rows = F * s.Ho * s.Wo; s.cols = k * k * Cin;
    s.dHoWo = FastDiv(s.Ho * s.Wo); s.dWo = FastDiv(s.Wo); s.dkw = FastDiv(k); s.dCin = FastDiv(Cin);
    return s;
}
extern "C" {
// ---------------------------------------------------------------------------------------------- conv
int rvt_conv_fwd(const void* in, const void* w, void* out, int dtype, int F, int H, int W, int Cin, int Cout, int k,
                 int stride, int pad, void* stream) {
    RVT_CHECK(Cin % 8 == 0 && Cout % 8 == 0, "conv_fwd: channels must be multiples of 8 (Cin=%d Cout=%d)", Cin, Cout);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, {
        Im2colSrc<T> a = make_im2col<T>(in, F, H, W, Cin, k, stride, pad);
        PlainSrc<T> b{(const T*)w, a.cols, Cout, a.cols};
        EpStore<T> ep{(T*)out, Cout, nullptr, nullptr};
        DISPATCH_BN(Cout, (launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, a.rows, Cout, a.cols, 1, st)));
    });
    return check_launch("conv_fwd");
}

int rvt_conv_wgrad(const void* in, const void* dy, float* dw, float* ws, int dtype, int F, int H, int W, int Cin, int Cout,
                   int k, int stride, int pad, void* stream) {
    RVT_CHECK(Cin % 8 == 0 && Cout % 8 == 0, "conv_wgrad: channels must be multiples of 8");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, {
        Im2colSrc<T> b = make_im2col<T>(in, F, H, W, Cin, k, stride, pad);
        PlainSrc<T> a{(const T*)dy, Cout, b.rows, Cout};
        if (Cout <= 64 && ws != nullptr) {
            // narrow output-channel count (the stem): dW^T = im2col^T dy, so that the 128-row operand is the wide one
            // (k*k*Cin patch columns) and dy fills a 64-column tile exactly — a [Cout <= 64][.] tile would leave half of
            // every MFMA empty; the reduction writes the transpose back
            constexpr int BN = 64;
            launch_wgrad<T, BN>(b, a, XfNone(), dw, nullptr, ws, b.cols, Cout, b.rows, st, true);
        } else {
            DISPATCH_WGRAD_BN(b.cols, (launch_wgrad<T, BN>(a, b, XfNone(), dw, nullptr, ws, Cout, b.cols, b.rows, st)));
        }
    });
    return check_launch("conv_wgrad");
}

// ---- the stem on the uint8 planes (stem.hpp) ----
static const int STEM_FWD_PB = 4;
static int stem_fwd_depth() {                          // software-pipeline depth of the forward's plane loads (k-steps in flight)
    static const int d = getenv("RVT_STEM_D") ? atoi(getenv("RVT_STEM_D")) : 4;
    return d == 5 ? 5 : 4;
}
static int stem_wgrad_grid(int n_tiles) {            // one workgroup per CU; RVT_STEM_GRID: a smaller grid (tests: multi-tile walks)
    static const int cap = getenv("RVT_STEM_GRID") ? atoi(getenv("RVT_STEM_GRID")) : 256;
    return n_tiles < 1 ? 1 : (n_tiles < cap ? n_tiles : cap);
}

int rvt_stem_supported(int dtype, int src_u8, int Cin, int Cout, int k, int stride, int pad, int w) {
    static const int on = getenv("RVT_STEM") ? atoi(getenv("RVT_STEM")) : 1;
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
    const int og8 = (g.OG + 7) / 8;
    g.n_items = F * og8;
    g.dOG = FastDiv(og8); g.d7 = FastDiv(STEM_K);
    const int grid = stem_wgrad_grid(g.n_items);
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
    return (size_t)stem_wgrad_grid(F * Ho * ((Wo + 31) / 32)) * (size_t)(NJB * 32) * STEM_CO;
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
    const int grid = stem_wgrad_grid(g.n_tiles);
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

// The same input gradient for the 3 x 3 / stride 2 / pad 1 convs of stages 2-4 as ONE product over 2 x 2 input-pixel blocks
// (ppgemm.hpp, GATHER): wd4 = [4 Cin][4 Cout] block-sparse weights (PACK_CONV_DGRAD4).  Measured against the four
// parity-class launches above: profiles/r3/microbench_conv_dgrad4.txt.
int rvt_conv_dgrad4_supported(int dtype, int H, int W, int Cin, int Cout, int k, int stride, int pad, int F) {
    if (dtype != RVT_BF16 || k != 3 || stride != 2 || pad != 1 || (H & 1) || (W & 1)) return 0;
    if (Cin % 64 != 0 || Cin > 512 || Cout % 64 != 0) return 0;
    const long long M = (long long)F * (H / 2) * (W / 2);
    if (M * Cout * 2 >= (1ll << 31) || (long long)F * H * W * Cin * 2 >= (1ll << 32)) return 0;
    return use_ppgemm(dtype, (int)M, 4 * Cin, 4 * Cout, Cout, 4 * Cout, 4 * Cout) ? 1 : 0;
}
int rvt_conv_dgrad4(const void* dy, const void* wd4, const void* add, void* din, int dtype, int F, int H, int W, int Cin, int Cout,
                    void* stream) {
    RVT_CHECK(rvt_conv_dgrad4_supported(dtype, H, W, Cin, Cout, 3, 2, 1, F), "conv_dgrad4: unsupported shape H=%d W=%d Cin=%d Cout=%d", H, W, Cin, Cout);
    hipStream_t st = (hipStream_t)stream;
    PPConv cv;
    cv.Ho = H / 2; cv.Wo = W / 2; cv.Cout = Cout; cv.H = H; cv.W = W; cv.Cin = Cin;
    cv.dHoWo = FastDiv(cv.Ho * cv.Wo); cv.dWo = FastDiv(cv.Wo);
    const int N = 4 * Cin, n_tiles = N / 256;
    for (int nt = 0; nt < 4; nt++) {
        int mask = 0;
        if (nt < n_tiles)
            for (int cls = (nt * 256) / Cin; cls <= (nt * 256 + 255) / Cin; cls++)
                for (int da = 0; da <= (cls >> 1); da++)
                    for (int db = 0; db <= (cls & 1); db++) mask |= 1 << (2 * da + db);
        cv.taps[nt] = mask ? mask : 1;
    }
    const int M = F * cv.Ho * cv.Wo;
    const PPMat xs{(const bf16*)dy, (const bf16*)dy, Cout, 1 << 30}, ws{(const bf16*)wd4, (const bf16*)wd4, 4 * Cout, 1 << 30};
    const PPEpArgs ep{(bf16*)din, nullptr, (const bf16*)add, nullptr, nullptr, Cin};
    if (add) launch_ppgemm<PP_ADD, 1>(xs, ws, ep, M, N, 4 * Cout, st, cv);
    else launch_ppgemm<PP_STORE, 1>(xs, ws, ep, M, N, 4 * Cout, st, cv);
    return check_launch("conv_dgrad4");
}

int rvt_conv_dgrad(const void* dy, const void* wd, const void* add, void* din, int dtype, int F, int H, int W, int Cin,
                   int Cout, int k, int stride, int pad, void* stream) {
    RVT_CHECK(Cin % 8 == 0 && Cout % 8 == 0, "conv_dgrad: channels must be multiples of 8");
    RVT_CHECK(stride >= 1 && stride <= 4 && k <= 4 * stride, "conv_dgrad: unsupported k=%d stride=%d", k, stride);
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    DISPATCH_DTYPE(dtype, {
        size_t woff = 0;
        for (int py = 0; py < stride; py++)
            for (int px = 0; px < stride; px++) {
                DgradSrc<T> a;
                a.dy = (const T*)dy; a.Ho = Ho; a.Wo = Wo; a.Cout = Cout;
                a.s = stride; a.pad = pad; a.py = py; a.px = px;
                a.Hc = (H - py + stride - 1) / stride; a.Wc = (W - px + stride - 1) / stride;
                a.nky = 0; a.nkx = 0;
                for (int t = 0; t < k; t++) {
                    if (t % stride == (py + pad) % stride) a.ky[a.nky++] = t;
                    if (t % stride == (px + pad) % stride) a.kx[a.nkx++] = t;
                }
                if (a.Hc <= 0 || a.Wc <= 0) continue;
                a.rows = F * a.Hc * a.Wc; a.cols = a.nky * a.nkx * Cout;
                a.dHcWc = FastDiv(a.Hc * a.Wc); a.dWc = FastDiv(a.Wc); a.dCout = FastDiv(Cout);
                PlainSrc<T> b{(const T*)wd + woff, a.cols, Cin, a.cols};
                EpDgradScatter<T> ep{(T*)din, (const T*)add, H, W, Cin, stride, py, px, a.dHcWc, a.dWc};
                DISPATCH_BN(Cin, (launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, a.rows, Cin, a.cols, 1, st)));
                woff += (size_t)Cin * a.cols;
            }
    });
    return check_launch("conv_dgrad");
}

// ----------------------------------------------------------------------------------------- layernorm
int rvt_layernorm_fwd(const void* x, const float* w, const float* b, void* y, int dtype, int rows, int C, float eps,
                      void* stream) {
    RVT_CHECK(C % 8 == 0 && C <= 512, "layernorm: C=%d must be a multiple of 8 and <= 512", C);
    hipStream_t st = (hipStream_t)stream;
    int G = pow2_ge(C / 8);
    int rows_per_block = 4 * (64 / G);
    int grid = imin(4096, imax(1, (rows + rows_per_block - 1) / rows_per_block));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_fwd_kernel<T>), dim3(grid), dim3(256), 0, st, (const T*)x, w, b, (T*)y,
                                             rows, C, G, eps));
    return check_launch("layernorm_fwd");
}

int rvt_layernorm_bwd(const void* x, const float* w, const void* dy, const void* dres, void* dx, float* dw, float* db,
                      int dtype, int rows, int C, float eps, void* stream) {
    RVT_CHECK(C % 8 == 0 && C <= 512, "layernorm: C=%d must be a multiple of 8 and <= 512", C);
    hipStream_t st = (hipStream_t)stream;
    int G = pow2_ge(C / 8);
    int rows_per_block = 4 * (64 / G);
    int grid = imin(2048, imax(1, (rows + rows_per_block - 1) / rows_per_block));     // 8 workgroups (32 waves) per CU
    // (parameter gradients: one device atomic per column per workgroup.  Measured on MI355X against per-workgroup partial rows +
    // a column-sum fold: 0.90 / 0.42 / 0.21 / 0.12 ms vs 0.99 / 0.43 / 0.22 / 0.14 ms at the four RVT-Base stage shapes,
    // profiles/r2/microbench_ln.txt — the kernel is HBM-bound at 4-4.6 TB/s either way)
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((ln_bwd_kernel<T>), dim3(grid), dim3(256), 0, st, (const T*)x, w,
                                             (const T*)dy, (const T*)dres, (T*)dx, dw, db, rows, C, G, eps));
    return check_launch("layernorm_bwd");
}

// -------------------------------------------------------------------------------------------- linear
int rvt_linear_fwd(const void* x, const void* w, const float* bias, void* y, int dtype, int M, int N, int K, int gelu_in,
                   void* stream) {
    RVT_CHECK(N % 8 == 0 && K % 8 == 0, "linear_fwd: N=%d K=%d must be multiples of 8", N, K);
    hipStream_t st = (hipStream_t)stream;
    if (!gelu_in && use_ppgemm(dtype, M, N, K, K, K, K)) {
        launch_ppgemm<PP_STORE>(PPMat{(const bf16*)x, (const bf16*)x, K, K}, PPMat{(const bf16*)w, (const bf16*)w, K, K},
                                PPEpArgs{(bf16*)y, nullptr, nullptr, bias, nullptr, N}, M, N, K, st);
        return check_launch("linear_fwd");
    }
    DISPATCH_DTYPE(dtype, {
        PlainSrc<T> a{(const T*)x, K, M, K};
        PlainSrc<T> b{(const T*)w, K, N, K};
        EpStore<T> ep{(T*)y, N, bias, nullptr};
        DISPATCH_BN(N, {
            if (gelu_in) launch_gemm<T, BN, false>(a, XfGelu(), b, XfNone(), ep, M, N, K, 1, st);
            else launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, M, N, K, 1, st);
        });
    });
    return check_launch("linear_fwd");
}

int rvt_linear_gelu_fwd(const void* x, const void* w, const float* bias, void* g, void* gp, int dtype, int M, int N, int K,
                        void* stream) {
    RVT_CHECK(N % 8 == 0 && K % 8 == 0 && bias, "linear_gelu_fwd: N=%d K=%d must be multiples of 8, bias required", N, K);
    hipStream_t st = (hipStream_t)stream;
    if (K >= pp_min_k(512) && use_ppgemm(dtype, M, N, K, K, K, K)) {        // (measured: 0.51 vs 0.58 ms at K = 512, 0.87 vs 0.76 at K = 256)
        launch_ppgemm<PP_GELU_DUAL>(PPMat{(const bf16*)x, (const bf16*)x, K, K}, PPMat{(const bf16*)w, (const bf16*)w, K, K},
                                    PPEpArgs{(bf16*)g, (bf16*)gp, nullptr, bias, nullptr, N}, M, N, K, st);
        return check_launch("linear_gelu_fwd");
    }
    DISPATCH_DTYPE(dtype, {
        PlainSrc<T> a{(const T*)x, K, M, K};
        PlainSrc<T> b{(const T*)w, K, N, K};
        EpGeluDual<T> ep{(T*)g, (T*)gp, N, bias};
        DISPATCH_BN(N, (launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, M, N, K, 1, st)));
    });
    return check_launch("linear_gelu_fwd");
}

int rvt_linear_scale_res_fwd(const void* x, const void* w, const float* bias, const float* gamma, const void* res,
                             void* y, int dtype, int M, int N, int K, int gelu_in, void* stream) {
    RVT_CHECK(N % 8 == 0 && K % 8 == 0, "linear_scale_res_fwd: N=%d K=%d must be multiples of 8", N, K);
    RVT_CHECK(bias && gamma && res, "linear_scale_res_fwd: bias, gamma and res are required");
    hipStream_t st = (hipStream_t)stream;
    if (!gelu_in && N <= PPGeom::MAX_CST / 2 && use_ppgemm(dtype, M, N, K, K, K, K)) {
        launch_ppgemm<PP_SCALE_RES>(PPMat{(const bf16*)x, (const bf16*)x, K, K}, PPMat{(const bf16*)w, (const bf16*)w, K, K},
                                    PPEpArgs{(bf16*)y, nullptr, (const bf16*)res, bias, gamma, N}, M, N, K, st);
        return check_launch("linear_scale_res_fwd");
    }
    DISPATCH_DTYPE(dtype, {
        PlainSrc<T> a{(const T*)x, K, M, K};
        PlainSrc<T> b{(const T*)w, K, N, K};
        EpScaleRes<T> ep{(T*)y, (const T*)res, N, bias, gamma};
        DISPATCH_BN(N, {
            if (gelu_in) launch_gemm<T, BN, false>(a, XfGelu(), b, XfNone(), ep, M, N, K, 1, st);
            else launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, M, N, K, 1, st);
        });
    });
    return check_launch("linear_scale_res_fwd");
}

int rvt_linear_dgrad(const void* dy, const void* wt, const void* gelu_pre, const void* add, const void* mul, void* dx,
                     int dtype, int M, int N, int K, void* stream) {
    RVT_CHECK(N % 8 == 0 && K % 8 == 0, "linear_dgrad: N=%d K=%d must be multiples of 8", N, K);
    RVT_CHECK((gelu_pre != nullptr) + (add != nullptr) + (mul != nullptr) <= 1,
              "linear_dgrad: gelu_pre, add and mul are mutually exclusive");
    hipStream_t st = (hipStream_t)stream;
    // dx[M][K] = dy[M][N] . wt[K][N]^T: output width K, contraction N
    if (!gelu_pre && use_ppgemm(dtype, M, K, N, N, N, N)) {
        const PPMat xs{(const bf16*)dy, (const bf16*)dy, N, N}, ws{(const bf16*)wt, (const bf16*)wt, N, N};
        if (mul) launch_ppgemm<PP_MUL>(xs, ws, PPEpArgs{(bf16*)dx, nullptr, (const bf16*)mul, nullptr, nullptr, K}, M, K, N, st);
        else if (add) launch_ppgemm<PP_ADD>(xs, ws, PPEpArgs{(bf16*)dx, nullptr, (const bf16*)add, nullptr, nullptr, K}, M, K, N, st);
        else launch_ppgemm<PP_STORE>(xs, ws, PPEpArgs{(bf16*)dx, nullptr, nullptr, nullptr, nullptr, K}, M, K, N, st);
        return check_launch("linear_dgrad");
    }
    DISPATCH_DTYPE(dtype, {
        PlainSrc<T> a{(const T*)dy, N, M, N};
        PlainSrc<T> b{(const T*)wt, N, K, N};
        DISPATCH_BN(K, {
            if (gelu_pre) {
                EpGeluBwd<T> ep{(T*)dx, (const T*)gelu_pre, K};
                launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, M, K, N, 1, st);
            } else if (mul) {
                EpMul<T> ep{(T*)dx, (const T*)mul, K};
                launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, M, K, N, 1, st);
            } else {
                EpStore<T> ep{(T*)dx, K, nullptr, (const T*)add};
                launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, M, K, N, 1, st);
            }
        });
    });
    return check_launch("linear_dgrad");
}

int rvt_linear_wgrad(const void* dy, const void* x, float* dw, float* dy_colsum, float* ws, int dtype, int M, int N, int K,
                     int gelu_in, void* stream) {
    RVT_CHECK(N % 8 == 0 && K % 8 == 0, "linear_wgrad: N=%d K=%d must be multiples of 8", N, K);
    hipStream_t st = (hipStream_t)stream;
    if (!gelu_in && ws != nullptr && use_ppgemm_tn(dtype, M, N, K, N, K, K)) {
        launch_ppgemm_tn((const bf16*)dy, N, (const bf16*)x, (const bf16*)x, K, K, dw, dy_colsum, ws, M, N, K, st);
        return check_launch("linear_wgrad");
    }
    DISPATCH_DTYPE(dtype, {
        PlainSrc<T> a{(const T*)dy, N, M, N};
        PlainSrc<T> b{(const T*)x, K, M, K};
        DISPATCH_WGRAD_BN(K, {
            if (gelu_in) launch_wgrad<T, BN>(a, b, XfGelu(), dw, dy_colsum, ws, N, K, M, st);
            else launch_wgrad<T, BN>(a, b, XfNone(), dw, dy_colsum, ws, N, K, M, st);
        });
    });
    return check_launch("linear_wgrad");
}

// ------------------------------------------------------------------------------------------ fused MLP
int rvt_mlp_fused_supported(int dtype, int C) {
    if (dtype == RVT_BF16) return C == 64 || C == 128;
    if (dtype == RVT_F32) return C == 64;
    return 0;
}

// tile height and resident workgroups per CU of the fused MLP kernels (LDS: ~41 KiB at bf16 C=64 TM=64, ~57 KiB at C=128)
static int mlp_tm(int dtype, int C) {
    static const int tm_override = getenv("RVT_MLP_TM") ? atoi(getenv("RVT_MLP_TM")) : 0;     // tuning knob (bf16 C=64 only)
    if (dtype == RVT_BF16 && C == 64 && tm_override == 128) return 128;
    return 64;
}
}  // extern "C"
// persistent grid = exactly the workgroups the chip holds at once for THIS kernel instantiation (registers + LDS)
// resident workgroups per CU of a kernel, queried once per kernel (the occupancy API is not free and must not run per
// launch; keyed by the kernel's address because several instantiations share one function type)
template <class K> static int resident_per_cu(K kernel, int threads, int fallback) {
#ifdef RVT_EMU
    return fallback;
#else
    struct Entry { const void* k; int v; };
    static Entry cache[64];
    static int n = 0;
    const void* key = reinterpret_cast<const void*>(kernel);
    for (int i = 0; i < n; i++) if (cache[i].k == key) return cache[i].v;
    int nb = 0, v = fallback;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, 0) == hipSuccess && nb > 0) v = nb;
    if (n < 64) cache[n++] = Entry{key, v};
    return v;
#endif
}
template <class K> static int mlp_grid(K kernel, int M, int tm) {
    static const int resident_override = getenv("RVT_GEMM_RESIDENT") ? atoi(getenv("RVT_GEMM_RESIDENT")) : 0;
    const int n_tiles = (M + tm - 1) / tm;
    const int per_cu = resident_per_cu(kernel, 256, 2);
    return imax(1, imin(n_tiles, resident_override > 0 ? resident_override : 256 * per_cu));
}
// register-chained MLP kernels (csrc/mlp_chain.hpp): C == 64; RVT_MLP_CHAIN=0 falls back to the LDS-staged kernels of mlp.hpp
static bool mlp_chain_on(int dtype, int C) {
    static const int off = getenv("RVT_MLP_CHAIN") ? atoi(getenv("RVT_MLP_CHAIN")) == 0 : 0;
    return !off && C == 64 && (dtype == RVT_BF16 || dtype == RVT_F32);
}
#ifndef MC_FWD_WPB
#define MC_FWD_WPB 8      // (6 waves x 3 per SIMD at <= 168 registers spills inside the chunk loop: 2.0 ms against 1.42)
#define MC_FWD_MINW 2
#endif
template <class T> struct McWaves { static constexpr int V = sizeof(T) == 2 ? 8 : 4; };
template <class K> static int mc_grid(K kernel, int threads, int M, int wpb) {
    static const int resident_override = getenv("RVT_MC_RESIDENT") ? atoi(getenv("RVT_MC_RESIDENT")) : 0;
    const int per_cu = resident_per_cu(kernel, threads, 1);
    const int want = ((M + 31) / 32 + wpb - 1) / wpb;
    return imax(1, imin(want, resident_override > 0 ? resident_override : 256 * per_cu));
}
extern "C" {

// dx = add + LN'(dy W; x) in one launch (csrc/dgrad_ln.hpp): bf16, C in {64, 128}, K = 3C or 4C.  RVT_DGRAD_LN=0 disables.
int rvt_linear_dgrad_ln_supported(int dtype, int C, int K) {
    static const int on = getenv("RVT_DGRAD_LN") ? atoi(getenv("RVT_DGRAD_LN")) : 1;
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

int rvt_mlp_fwd(const void* xmid, void* xout, void* g_out, void* gp_out, void* v2_out, const float* ln_w, const float* ln_b,
                const void* w1, const float* b1, const void* w2, const float* b2, const float* gamma, int dtype, int M,
                int C, float eps, void* stream) {
    RVT_CHECK(rvt_mlp_fused_supported(dtype, C), "mlp_fwd: fused MLP not built for dtype=%d C=%d", dtype, C);
    RVT_CHECK((g_out == nullptr) == (gp_out == nullptr), "mlp_fwd: g_out and gp_out go together");
    hipStream_t st = (hipStream_t)stream;
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
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(reduce_grid(wc)), dim3(256), 0, st, p, dw1, grid, wc, 0);
    p += (size_t)grid * wc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(reduce_grid(wc)), dim3(256), 0, st, p, s2, grid, wc, 0);
    p += (size_t)grid * wc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(reduce_grid((size_t)4 * C)), dim3(256), 0, st, p, db1, 2 * grid, (size_t)4 * C, 0);
    p += (size_t)2 * grid * 4 * C;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(reduce_grid((size_t)C)), dim3(256), 0, st, p, cs2, grid, (size_t)C, 0);
}
extern "C" {
size_t rvt_mlp_bwd_fused_ws_floats(int dtype, int C, int M) {
    if (!rvt_mlp_bwd_fused_supported(dtype, C)) return 0;
    size_t grid = dtype == RVT_BF16 ? mlp_bwd_fused_grid<bf16, 2>(M) : mlp_bwd_fused_grid<float, 2>(M);
    if (grid < 256) grid = 256;                          // mlpc_bwd_wgrad_kernel: one workgroup per CU
    return grid * ((size_t)2 * 4 * C * C + 2 * 4 * C + C);
}

int rvt_mlp_bwd_recompute_dgrad(const void* dxout, const void* xmid, void* dxmid, const float* ln_w, const float* ln_b,
                                const void* w1, const float* b1, const void* w2g_t, const void* w1_t, float* dln_w,
                                float* dln_b, int dtype, int M, int C, float eps, void* stream) {
    RVT_CHECK(rvt_mlp_bwd_fused_supported(dtype, C), "mlp_bwd_recompute_dgrad: not built for dtype=%d C=%d", dtype, C);
    hipStream_t st = (hipStream_t)stream;
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
    int grid = 0;
    static const int chain_wgrad = getenv("RVT_MLP_CHAIN_WGRAD") ? atoi(getenv("RVT_MLP_CHAIN_WGRAD")) : 1;
    if (chain_wgrad && dtype == RVT_BF16 && mlp_chain_on(dtype, C)) {
        grid = stem_wgrad_grid((M + 31) / 32);          // one workgroup per CU (tests: RVT_STEM_GRID)
        hipLaunchKernelGGL(mlpc_bwd_wgrad_kernel, dim3(grid), dim3(512), 0, st, (const bf16*)dxout, (const bf16*)xmid, ln_w, ln_b,
                           (const bf16*)w1, b1, (const bf16*)w2g_t, ws, M, eps);
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

// ----------------------------------------------------------------------------------------- attention
static int make_attn_geom(AttnGeom& g, int F, int H, int W, int C, int dh, int ph, int pw, int window) {
    RVT_CHECK(C % 8 == 0 && dh % 8 == 0 && dh <= 32 && C % dh == 0, "attn: bad C=%d dim_head=%d", C, dh);
    RVT_CHECK(H % ph == 0 && W % pw == 0, "attn: %dx%d not divisible by partition %dx%d", H, W, ph, pw);
    RVT_CHECK(ph * pw <= 96, "attn: partition of %d tokens > 96 unsupported", ph * pw);
    g.F = F; g.H = H; g.W = W; g.C = C; g.dh = dh; g.heads = C / dh; g.ph = ph; g.pw = pw; g.L = ph * pw;
    g.window = window;
    g.nPw = W / pw; g.P = (H / ph) * (W / pw);
    g.scale = 1.0f / sqrtf((float)dh);
    g.dGroups = FastDiv(g.heads); g.dP = FastDiv(g.P); g.dnPw = FastDiv(g.nPw); g.dpw = FastDiv(pw);
    return 0;
}
}  // extern "C"

// heads per workgroup: the largest of 4 / 2 / 1 that divides the head count and whose backward LDS slices fit
constexpr int ATTN_LDS_BUDGET = 80 * 1024;
template <class T, int NB> static int attn_head_group(int heads) {
    for (int hg = 4; hg > 1; hg >>= 1)
        if (heads % hg == 0 && hg * AbBwdScratch<T, NB>::BYTES <= ATTN_LDS_BUDGET) return hg;
    return 1;
}
template <class T, int NB, int HG>
static void launch_attn(bool bwd, const void* qkv, const void* dout, void* out, AttnGeom g, hipStream_t st) {
    g.dGroups = FastDiv(g.heads / HG);
    dim3 grid((unsigned)(g.F * g.P * (g.heads / HG)));
    if constexpr (HG == 1 || HG * AbBwdScratch<T, NB>::BYTES <= ATTN_LDS_BUDGET) {
        if (bwd)
            hipLaunchKernelGGL((attn_core_bwd_kernel<T, NB, HG>), grid, dim3(64 * HG), 0, st, (const T*)qkv, (const T*)dout, (T*)out, g);
        else
            hipLaunchKernelGGL((attn_core_fwd_kernel<T, NB, HG>), grid, dim3(64 * HG), 0, st, (const T*)qkv, (T*)out, g);
    }
}
template <class T, int NB>
static void launch_attn_nb(bool bwd, const void* qkv, const void* dout, void* out, const AttnGeom& g, hipStream_t st) {
    const int hg = attn_head_group<T, NB>(g.heads);
    if (hg == 4) launch_attn<T, NB, 4>(bwd, qkv, dout, out, g, st);
    else if (hg == 2) launch_attn<T, NB, 2>(bwd, qkv, dout, out, g, st);
    else launch_attn<T, NB, 1>(bwd, qkv, dout, out, g, st);
}
template <class T>
static void launch_attn_any(bool bwd, const void* qkv, const void* dout, void* out, const AttnGeom& g, hipStream_t st) {
    const int NB = (g.L + 31) / 32;
    if (NB == 1) launch_attn_nb<T, 1>(bwd, qkv, dout, out, g, st);
    else if (NB == 2) launch_attn_nb<T, 2>(bwd, qkv, dout, out, g, st);
    else launch_attn_nb<T, 3>(bwd, qkv, dout, out, g, st);
}

extern "C" {

int rvt_attn_fwd(const void* qkv, void* out, int dtype, int F, int H, int W, int C, int dim_head, int ph, int pw,
                 int window, void* stream) {
    AttnGeom g;
    if (make_attn_geom(g, F, H, W, C, dim_head, ph, pw, window)) return 1;
    DISPATCH_DTYPE(dtype, (launch_attn_any<T>(false, qkv, nullptr, out, g, (hipStream_t)stream)));
    return check_launch("attn_fwd");
}

int rvt_attn_bwd(const void* qkv, const void* dout, void* dqkv, int dtype, int F, int H, int W, int C, int dim_head,
                 int ph, int pw, int window, void* stream) {
    AttnGeom g;
    if (make_attn_geom(g, F, H, W, C, dim_head, ph, pw, window)) return 1;
    DISPATCH_DTYPE(dtype, (launch_attn_any<T>(true, qkv, dout, dqkv, g, (hipStream_t)stream)));
    return check_launch("attn_bwd");
}

// ------------------------------------------------------------------------- fused attention half (csrc/attn_block.hpp)
int rvt_attn_block_supported(int dtype, int C, int dim_head, int L) {
    if (dim_head != 32 || C != 64 || L <= 32 || L > 96) return 0;
    return dtype == RVT_BF16 || dtype == RVT_F32;
}
}  // extern "C"
// waves per workgroup: what the LDS holds (weights + per-wave backward scratch)
template <class T, int NB> struct AbWaves { static constexpr int V = sizeof(T) == 2 ? (NB == 2 ? 4 : 3) : 2; };
template <class K> static int ab_grid(K kernel, int threads, int n_part, int wpb) {
    static const int resident_override = getenv("RVT_AB_RESIDENT") ? atoi(getenv("RVT_AB_RESIDENT")) : 0;
    const int per_cu = resident_per_cu(kernel, threads, 1);
    const int want = (n_part + wpb - 1) / wpb;
    return imax(1, imin(want, resident_override > 0 ? resident_override : 256 * per_cu));
}
template <class T, int NB, bool LN>
static void launch_ab_fwd(const void* x, void* xmid, void* a_out, const float* ln_w, const float* ln_b, const void* wqkv,
                          const float* bqkv, const void* wp, const float* bp, const float* gamma, const AttnGeom& g, float eps,
                          hipStream_t st) {
    constexpr int WPB = AbWaves<T, NB>::V;
    auto k = attn_block_fwd_kernel<T, 64, NB, LN, WPB>;
    hipLaunchKernelGGL(k, dim3(ab_grid(k, 64 * WPB, g.F * g.P, WPB)), dim3(64 * WPB), 0, st, (const T*)x, (T*)xmid, (T*)a_out, ln_w,
                       ln_b, (const T*)wqkv, bqkv, (const T*)wp, bp, gamma, g, eps);
}
template <class T, int NB, bool LN>
static void launch_ab_bwd(const void* x, const void* dxmid, void* dx, void* dqkv, void* u_out, const float* ln_w,
                          const float* ln_b, const void* wqkv, const float* bqkv, const void* wpg_t, float* dln_w, float* dln_b,
                          const AttnGeom& g, float eps, hipStream_t st) {
    constexpr int WPB = AbWaves<T, NB>::V;
    auto k = attn_block_bwd_kernel<T, 64, NB, LN, WPB>;
    hipLaunchKernelGGL(k, dim3(ab_grid(k, 64 * WPB, g.F * g.P, WPB)), dim3(64 * WPB), 0, st, (const T*)x, (const T*)dxmid, (T*)dx,
                       (T*)dqkv, (T*)u_out, ln_w, ln_b, (const T*)wqkv, bqkv, (const T*)wpg_t, dln_w, dln_b, g, eps);
}
extern "C" {
int rvt_attn_block_fwd(const void* x, void* xmid, void* a_out, const float* ln_w, const float* ln_b, const void* wqkv,
                       const float* bqkv, const void* wp, const float* bp, const float* gamma, int dtype, int F, int H, int W,
                       int C, int dim_head, int ph, int pw, int window, float eps, void* stream) {
    RVT_CHECK(rvt_attn_block_supported(dtype, C, dim_head, ph * pw), "attn_block_fwd: not built for dtype=%d C=%d dim_head=%d L=%d",
              dtype, C, dim_head, ph * pw);
    RVT_CHECK((ln_w == nullptr) == (ln_b == nullptr), "attn_block_fwd: ln_w and ln_b go together");
    AttnGeom g;
    if (make_attn_geom(g, F, H, W, C, dim_head, ph, pw, window)) return 1;
    const int NB = (g.L + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
#define RVT_AB_FWD(NBB, LNN) launch_ab_fwd<T, NBB, LNN>(x, xmid, a_out, ln_w, ln_b, wqkv, bqkv, wp, bp, gamma, g, eps, st)
    DISPATCH_DTYPE(dtype, {
        if (NB == 2) { if (ln_w) RVT_AB_FWD(2, true); else RVT_AB_FWD(2, false); }
        else { if (ln_w) RVT_AB_FWD(3, true); else RVT_AB_FWD(3, false); }
    });
#undef RVT_AB_FWD
    return check_launch("attn_block_fwd");
}

int rvt_attn_block_bwd(const void* x, const void* dxmid, void* dx, void* dqkv, void* u_out, const float* ln_w,
                       const float* ln_b, const void* wqkv, const float* bqkv, const void* wpg_t, float* dln_w, float* dln_b,
                       int dtype, int F, int H, int W, int C, int dim_head, int ph, int pw, int window, float eps,
                       void* stream) {
    RVT_CHECK(rvt_attn_block_supported(dtype, C, dim_head, ph * pw), "attn_block_bwd: not built for dtype=%d C=%d dim_head=%d L=%d",
              dtype, C, dim_head, ph * pw);
    RVT_CHECK((ln_w == nullptr) == (ln_b == nullptr), "attn_block_bwd: ln_w and ln_b go together");
    RVT_CHECK(ln_w == nullptr || (dln_w != nullptr && dln_b != nullptr), "attn_block_bwd: LayerNorm gradients need dln_w / dln_b");
    AttnGeom g;
    if (make_attn_geom(g, F, H, W, C, dim_head, ph, pw, window)) return 1;
    const int NB = (g.L + 31) / 32;
    hipStream_t st = (hipStream_t)stream;
#define RVT_AB_BWD(NBB, LNN) launch_ab_bwd<T, NBB, LNN>(x, dxmid, dx, dqkv, u_out, ln_w, ln_b, wqkv, bqkv, wpg_t, dln_w, dln_b, g, eps, st)
    // Three 32-token blocks per partition (the 8 x 10 Gen1 partitions) exist for the forward only: the backward's per-wave
    // state for three blocks needs more than 512 registers (584-868 bytes per lane of scratch, 1.8-2.7 ms against 1.2-1.5 ms for
    // the op-by-op chain at RVT-Base / Gen1, profiles/r3/microbench_attn_block_gen1.txt) - training takes the chain there.
    RVT_CHECK(NB == 2, "attn_block_bwd: partitions of %d tokens (more than two 32-token blocks) are forward-only", g.L);
    DISPATCH_DTYPE(dtype, { if (ln_w) RVT_AB_BWD(2, true); else RVT_AB_BWD(2, false); });
#undef RVT_AB_BWD
    return check_launch("attn_block_bwd");
}

// ---------------------------------------------------------------------------------------------- lstm
int rvt_lstm_fwd(const void* x, const void* h_prev, const float* c_prev, const void* w_perm, const float* b_perm,
                 void* h_out, float* c_out, void* gates, int dtype, int M, int C, void* stream) {
    RVT_CHECK(C % 8 == 0, "lstm_fwd: C=%d must be a multiple of 8", C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, {
        ConcatSrc<T> a{(const T*)x, (const T*)h_prev, C, M, 2 * C};
        PlainSrc<T> b{(const T*)w_perm, 2 * C, 4 * C, 2 * C};
        EpLstm<T> ep{b_perm, c_prev, c_out, (T*)h_out, (T*)gates, C};
        DISPATCH_BN(4 * C, (launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, M, 4 * C, 2 * C, 1, st)));
    });
    return check_launch("lstm_fwd");
}

int rvt_lstm_gates_bwd(const void* dh_in, const void* dh_rec, float* dc_rec, const void* gates, const float* c_new,
                       const float* c_prev, void* dz, int dtype, int M, int C, void* stream) {
    RVT_CHECK(C % 8 == 0, "lstm_gates_bwd: C=%d must be a multiple of 8", C);
    hipStream_t st = (hipStream_t)stream;
    int grid = grid_for((size_t)M * (C / 8), 4096);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((lstm_gates_bwd_kernel<T>), dim3(grid), dim3(256), 0, st, (const T*)dh_in,
                                             (const T*)dh_rec, dc_rec, (const T*)gates, c_new, c_prev, (T*)dz, M, C));
    return check_launch("lstm_gates_bwd");
}

int rvt_lstm_dgrad(const void* dz, const void* wt, void* dx, void* dh_rec, int dtype, int M, int C, void* stream) {
    RVT_CHECK(C % 8 == 0, "lstm_dgrad: C=%d must be a multiple of 8", C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, {
        PlainSrc<T> a{(const T*)dz, 4 * C, M, 4 * C};
        PlainSrc<T> b{(const T*)wt, 4 * C, 2 * C, 4 * C};
        EpSplit2<T> ep{(T*)dx, (T*)dh_rec, C};
        DISPATCH_BN(2 * C, (launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, M, 2 * C, 4 * C, 1, st)));
    });
    return check_launch("lstm_dgrad");
}

int rvt_lstm_wgrad(const void* dz, const void* x, const void* h_prev, float* dw, float* dz_colsum, float* ws, int dtype,
                   int M, int C, void* stream) {
    RVT_CHECK(C % 8 == 0, "lstm_wgrad: C=%d must be a multiple of 8", C);
    hipStream_t st = (hipStream_t)stream;
    if (ws != nullptr && use_ppgemm_tn(dtype, M, 4 * C, 2 * C, 4 * C, C, C)) {       // [x | h]: two [M][C] matrices side by side
        launch_ppgemm_tn((const bf16*)dz, 4 * C, (const bf16*)x, (const bf16*)h_prev, C, C, dw, dz_colsum, ws, M, 4 * C, 2 * C, st);
        return check_launch("lstm_wgrad");
    }
    DISPATCH_DTYPE(dtype, {
        PlainSrc<T> a{(const T*)dz, 4 * C, M, 4 * C};
        ConcatSrc<T> b{(const T*)x, (const T*)h_prev, C, M, 2 * C};
        DISPATCH_WGRAD_BN(2 * C, (launch_wgrad<T, BN>(a, b, XfNone(), dw, dz_colsum, ws, 4 * C, 2 * C, M, st)));
    });
    return check_launch("lstm_wgrad");
}

// ------------------------------------------------------------------------- ConvLSTM, time loop in the kernel
// configurations built (waves per workgroup, weights resident in LDS or streamed from L2):
//   bf16: C = 32 (4 waves, LDS), 64 (8 fwd / 4 bwd waves, LDS), 128 (weights from L2);  f32 (parity): C = 32, 64, 128 from L2
int rvt_lstm_scan_supported(int dtype, int C) {
    if (dtype != RVT_BF16 && dtype != RVT_F32) return 0;
    return C == 32 || C == 64 || C == 128;
}
}  // extern "C"
template <class K> static int scan_grid(K kernel, int threads, int M, int tm) {
    static const int resident_override = getenv("RVT_GEMM_RESIDENT") ? atoi(getenv("RVT_GEMM_RESIDENT")) : 0;
    const int n_tiles = (M + tm - 1) / tm;
    const int per_cu = resident_per_cu(kernel, threads, 1);
    return imax(1, imin(n_tiles, resident_override > 0 ? resident_override : 256 * per_cu));
}
template <class T, int C, int NW, int RB, bool W_LDS, bool W_REG = false>
static void launch_lstm_scan_fwd(const void* x_all, void* Hall, const float* c0, float* c_last, void* Csave, const void* W,
                                 const float* bias, void* gates_out, int M, int Tn, hipStream_t st) {
    constexpr int TM = (NW / (C / 32)) * RB * 32;
    auto k = lstm_scan_fwd_kernel<T, C, NW, RB, W_LDS, W_REG>;
    hipLaunchKernelGGL(k, dim3(scan_grid(k, 64 * NW, M, TM)), dim3(64 * NW), 0, st, (const T*)x_all, (T*)Hall, c0, c_last,
                       (T*)Csave, (const T*)W, bias, (T*)gates_out, M, Tn);
}
// C = 128 in bf16: weights resident in the register file (forward) / gates saved for a reverse scan that keeps W^T in registers
static bool scan_regw_built(int dtype, int C) { return dtype == RVT_BF16 && C == 128; }
// in-kernel weight gradients of the reverse scan: where the weights are LDS-resident (bf16, C <= 64)
static bool scan_wgrad_built(int dtype, int C) { return dtype == RVT_BF16 && (C == 32 || C == 64); }
template <class T, int C, int NW, bool W_LDS, bool WGRAD>
static int lstm_scan_bwd_grid(int M) {
    constexpr int TM = (NW / (C / 32)) * 32;
    auto k = lstm_scan_bwd_kernel<T, C, NW, W_LDS, WGRAD>;
    return scan_grid(k, 64 * NW, M, TM);
}
template <class T, int C, int NW, bool W_LDS, bool WGRAD>
static void launch_lstm_scan_bwd(const void* x_all, const void* Hall, const void* Csave, const float* c0, const void* dH,
                                 const float* dc_last, const void* W, const void* Wt, const float* bias, void* dx_all,
                                 void* dz_all, void* dh0, float* dc0, float* dw, float* db, float* ws, int M, int Tn,
                                 hipStream_t st) {
    auto k = lstm_scan_bwd_kernel<T, C, NW, W_LDS, WGRAD>;
    const int grid = lstm_scan_bwd_grid<T, C, NW, W_LDS, WGRAD>(M);
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * NW), 0, st, (const T*)x_all, (const T*)Hall,
                       (const T*)Csave, c0, (const T*)dH, dc_last, (const T*)W, (const T*)Wt, bias, (T*)dx_all, (T*)dz_all,
                       (T*)dh0, dc0, ws, (const T*)nullptr, M, Tn);
    if (WGRAD) {
        // fold the per-workgroup partial records: the [4C][2C] weight block and the NWM bias rows are column sums over records
        constexpr int NWM = NW / (C / 32);
        const size_t rec = (size_t)4 * C * 2 * C + (size_t)NWM * 4 * C;
        hipLaunchKernelGGL(strided_reduce_kernel, dim3(reduce_grid((size_t)4 * C * 2 * C)), dim3(256), 0, st, (const float*)ws, dw,
                           grid, rec, (size_t)4 * C * 2 * C);
        for (int m = 0; m < NWM; m++)
            hipLaunchKernelGGL(strided_reduce_kernel, dim3(reduce_grid((size_t)4 * C)), dim3(256), 0, st,
                               (const float*)(ws + (size_t)4 * C * 2 * C + (size_t)m * 4 * C), db, grid, rec, (size_t)4 * C);
    }
}
extern "C" {
size_t rvt_lstm_scan_bwd_ws_floats(int dtype, int C, int M) {
    if (!scan_wgrad_built(dtype, C)) return 0;
    const int grid = C == 32 ? lstm_scan_bwd_grid<bf16, 32, 4, true, true>(M) : lstm_scan_bwd_grid<bf16, 64, 4, true, true>(M);
    const int NWM = 4 / (C / 32);
    return (size_t)grid * ((size_t)4 * C * 2 * C + (size_t)NWM * 4 * C);
}

int rvt_lstm_scan_saves_gates(int dtype, int C) { return scan_regw_built(dtype, C) ? 1 : 0; }
int rvt_lstm_scan_fwd(const void* x_all, void* Hall, const float* c0, float* c_last, void* Csave, const void* w,
                      const float* bias, void* gates_out, int dtype, int M, int C, int T_steps, void* stream) {
    RVT_CHECK(rvt_lstm_scan_supported(dtype, C), "lstm_scan_fwd: not built for dtype=%d C=%d", dtype, C);
    RVT_CHECK(M >= 1 && T_steps >= 1, "lstm_scan_fwd: empty problem");
    hipStream_t st = (hipStream_t)stream;
    RVT_CHECK(gates_out == nullptr || scan_regw_built(dtype, C), "lstm_scan_fwd: gates are only saved by the bf16 C = 128 variant");
#define RVT_SCAN_FWD(TT, CC, NWW, RBB, LDS) launch_lstm_scan_fwd<TT, CC, NWW, RBB, LDS>(x_all, Hall, c0, c_last, Csave, w, bias, nullptr, M, T_steps, st)
    if (dtype == RVT_BF16) {
        if (C == 32) RVT_SCAN_FWD(bf16, 32, 4, 1, true);
        else if (C == 64) RVT_SCAN_FWD(bf16, 64, 8, 1, true);
        else launch_lstm_scan_fwd<bf16, 128, 4, 1, false, true>(x_all, Hall, c0, c_last, Csave, w, bias, gates_out, M, T_steps, st);     // (64-token tiles spill: 2.3 ms against 1.65)
    } else {             // (four waves: the f32 variants need more than the 256 registers an 8-wave workgroup leaves)
        if (C == 32) RVT_SCAN_FWD(float, 32, 4, 1, false);
        else if (C == 64) RVT_SCAN_FWD(float, 64, 4, 1, false);
        else RVT_SCAN_FWD(float, 128, 4, 1, false);
    }
#undef RVT_SCAN_FWD
    return check_launch("lstm_scan_fwd");
}

int rvt_lstm_scan_bwd(const void* x_all, const void* Hall, const void* Csave, const float* c0, const void* dH,
                      const float* dc_last, const void* w, const void* wt, const float* bias, void* dx_all, void* dz_all,
                      void* dh0, float* dc0, float* dw, float* db, float* ws, const void* gates, int dtype, int M, int C,
                      int T_steps, void* stream) {
    RVT_CHECK(rvt_lstm_scan_supported(dtype, C), "lstm_scan_bwd: not built for dtype=%d C=%d", dtype, C);
    RVT_CHECK(M >= 1 && T_steps >= 1 && Csave != nullptr, "lstm_scan_bwd: empty problem / missing saved cell states");
    hipStream_t st = (hipStream_t)stream;
    if (gates != nullptr) {       // reverse scan on the saved gates, W^T in registers (bf16, C = 128)
        RVT_CHECK(scan_regw_built(dtype, C) && dw == nullptr && dz_all != nullptr && M >= 1 && T_steps >= 1 && Csave != nullptr,
                  "lstm_scan_bwd: the saved-gates variant is bf16 C = 128, writes dz_all and has no in-kernel weight gradient");
        hipStream_t st = (hipStream_t)stream;
        auto k = lstm_scan_bwd_kernel<bf16, 128, 4, false, false, true>;
        hipLaunchKernelGGL(k, dim3(scan_grid(k, 256, M, 32)), dim3(256), 0, st, (const bf16*)x_all, (const bf16*)Hall,
                           (const bf16*)Csave, c0, (const bf16*)dH, dc_last, (const bf16*)w, (const bf16*)wt, bias, (bf16*)dx_all,
                           (bf16*)dz_all, (bf16*)dh0, dc0, (float*)nullptr, (const bf16*)gates, M, T_steps);
        return check_launch("lstm_scan_bwd(gates)");
    }
    const bool wgrad = dw != nullptr;
    RVT_CHECK(!wgrad || (scan_wgrad_built(dtype, C) && db != nullptr && ws != nullptr),
              "lstm_scan_bwd: in-kernel weight gradients need dtype bf16, C in {32, 64}, db and a workspace");
    RVT_CHECK(wgrad || dz_all != nullptr, "lstm_scan_bwd: dz_all required without in-kernel weight gradients");
#define RVT_SCAN_BWD(TT, CC, NWW, LDS, WG) launch_lstm_scan_bwd<TT, CC, NWW, LDS, WG>(x_all, Hall, Csave, c0, dH, dc_last, w, wt, bias, dx_all, dz_all, dh0, dc0, dw, db, ws, M, T_steps, st)
    if (dtype == RVT_BF16) {
        if (C == 32) { if (wgrad) RVT_SCAN_BWD(bf16, 32, 4, true, true); else RVT_SCAN_BWD(bf16, 32, 4, true, false); }
        else if (C == 64) { if (wgrad) RVT_SCAN_BWD(bf16, 64, 4, true, true); else RVT_SCAN_BWD(bf16, 64, 4, true, false); }
        else RVT_SCAN_BWD(bf16, 128, 4, false, false);
    } else {
        if (C == 32) RVT_SCAN_BWD(float, 32, 4, false, false);
        else if (C == 64) RVT_SCAN_BWD(float, 64, 4, false, false);
        else RVT_SCAN_BWD(float, 128, 4, false, false);
    }
#undef RVT_SCAN_BWD
    return check_launch("lstm_scan_bwd");
}

// ------------------------------------------------------------------------------------ depth-wise conv
int rvt_dwconv_fwd(const void* x, int ldx, const float* w, const float* b, void* y, int ldy, int dtype, int N, int H,
                   int W, int C, int k, int transpose, void* stream) {
    RVT_CHECK(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (k == 1 || k == 3), "dwconv: C=%d k=%d unsupported", C, k);
    hipStream_t st = (hipStream_t)stream;
    int grid = grid_for((size_t)N * H * W * (C / 8), 8192);
    DISPATCH_DTYPE(dtype, {
        if (transpose)
            hipLaunchKernelGGL((dwconv_kernel<T, true>), dim3(grid), dim3(256), 0, st, (const T*)x, ldx, w, b, (T*)y, ldy,
                               N, H, W, C, k);
        else
            hipLaunchKernelGGL((dwconv_kernel<T, false>), dim3(grid), dim3(256), 0, st, (const T*)x, ldx, w, b, (T*)y,
                               ldy, N, H, W, C, k);
    });
    return check_launch("dwconv");
}

int rvt_dwconv_wgrad(const void* x, int ldx, const void* dy, int ldy, float* dw, float* db, int dtype, int N, int H,
                     int W, int C, int k, void* stream) {
    RVT_CHECK(C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (k == 1 || k == 3), "dwconv_wgrad: C=%d k=%d unsupported", C, k);
    hipStream_t st = (hipStream_t)stream;
    int NC = C / 8;
    int CP = imin(256, pow2_ge(NC));
    int gy = (NC + CP - 1) / CP;
    int npl = 256 / CP;
    size_t npix = (size_t)N * H * W;
    int gx = (int)imin(1024, imax(1, (int)((npix + (size_t)npl * 16 - 1) / ((size_t)npl * 16))));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((dwconv_wgrad_kernel<T>), dim3(gx, gy), dim3(256), 0, st, (const T*)x, ldx,
                                             (const T*)dy, ldy, dw, db, N, H, W, C, k, CP));
    return check_launch("dwconv_wgrad");
}

// ---------------------------------------------------------------------------------------- token mask
int rvt_token_mask_fwd(void* x, const unsigned char* mask, const float* token, int dtype, int M, int C, void* stream) {
    RVT_CHECK(C % 8 == 0, "token_mask: C=%d must be a multiple of 8", C);
    hipStream_t st = (hipStream_t)stream;
    int grid = grid_for((size_t)M * (C / 8), 4096);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((token_mask_fwd_kernel<T>), dim3(grid), dim3(256), 0, st, (T*)x, mask, token, M, C));
    return check_launch("token_mask_fwd");
}

int rvt_token_mask_bwd(void* dx, const unsigned char* mask, float* dtoken, int dtype, int M, int C, void* stream) {
    RVT_CHECK(C % 8 == 0, "token_mask: C=%d must be a multiple of 8", C);
    hipStream_t st = (hipStream_t)stream;
    int NC = C / 8;
    int NCP = imin(256, pow2_ge(NC));
    int gy = (NC + NCP - 1) / NCP;
    int nrl = 256 / NCP;
    int gx = imin(512, imax(1, (M + nrl * 8 - 1) / (nrl * 8)));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((token_mask_bwd_kernel<T>), dim3(gx, gy), dim3(256), 0, st, (T*)dx, mask, dtoken,
                                             M, C, NCP));
    return check_launch("token_mask_bwd");
}

int rvt_stacked_histogram(const long long* x, const long long* y, const long long* pol, const long long* time,
                          size_t n_events, int bins, int H, int W, int count_cutoff, int fastmode, unsigned* scratch,
                          unsigned char* out, void* stream) {
    RVT_CHECK(bins >= 1 && H >= 1 && W >= 1 && count_cutoff >= 1 && count_cutoff <= 255,
              "stacked_histogram: bad geometry bins=%d H=%d W=%d cutoff=%d", bins, H, W, count_cutoff);
    hipStream_t st = (hipStream_t)stream;
    const size_t cells = (size_t)2 * bins * H * W;
    hipMemsetAsync(scratch, 0, cells * sizeof(unsigned), st);
    if (n_events > 0)
        hipLaunchKernelGGL(hist_count_kernel, dim3(grid_for(n_events, 4096)), dim3(256), 0, st, x, y, pol, time, n_events, bins,
                           H, W, scratch);
    hipLaunchKernelGGL(hist_finalize_kernel, dim3(grid_for(cells, 4096)), dim3(256), 0, st, (const unsigned*)scratch, out,
                       cells, count_cutoff, fastmode);
    return check_launch("stacked_histogram");
}

// ------------------------------------------------------------------------------ parameter-side tables
int rvt_pack_table(const void* descs, int n_desc, int total_blocks, int dtype, void* stream) {
    RVT_CHECK(n_desc >= 1 && total_blocks >= 1 && descs != nullptr, "pack_table: empty table");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pack_table_kernel<T>), dim3(total_blocks), dim3(256), 0, st,
                                             (const PackDesc*)descs, n_desc));
    return check_launch("pack_table");
}

int rvt_layerscale_grad_table(const void* descs, int n_desc, int total_blocks, void* stream) {
    RVT_CHECK(n_desc >= 1 && total_blocks >= 1 && descs != nullptr, "layerscale_grad_table: empty table");
    hipLaunchKernelGGL(layerscale_grad_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const LayerScaleDesc*)descs, n_desc);
    return check_launch("layerscale_grad_table");
}

int rvt_gather_frames(const void* src, const int* idx, void* dst, int n_sel, size_t frame_bytes, int scatter, void* stream) {
    RVT_CHECK(frame_bytes % 16 == 0, "gather_frames: frames of %zu bytes are not a whole number of 16-byte vectors", frame_bytes);
    if (n_sel <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const size_t fv = frame_bytes / 16;
    const int grid = grid_for((size_t)n_sel * fv, 8192);
    if (scatter)
        hipLaunchKernelGGL((gather_frames_kernel<true>), dim3(grid), dim3(256), 0, st, (const u32x4*)src, idx, (u32x4*)dst, n_sel, fv);
    else
        hipLaunchKernelGGL((gather_frames_kernel<false>), dim3(grid), dim3(256), 0, st, (const u32x4*)src, idx, (u32x4*)dst, n_sel, fv);
    return check_launch("gather_frames");
}

int rvt_state_reset_masked(void* st_, const unsigned char* mask, int dtype, int B, size_t per_sample, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int grid = grid_for((size_t)B * per_sample, 4096);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((state_reset_kernel<T>), dim3(grid), dim3(256), 0, st, (T*)st_, mask, B,
                                             per_sample));
    return check_launch("state_reset_masked");
}

}  // extern "C"
