// extern "C" entry points, part 2 of 8: down-sampling convolutions on the GEMM engines.
#include "gemm_host.hpp"

extern "C" {
size_t rvt_wgrad_workspace_floats(int dtype, int out_rows, int out_cols, int tokens, int want_colsum) {
    size_t pp = 0;
    if (use_ppgemm_tn(dtype, tokens, out_rows, out_cols, out_rows, out_cols, out_cols))
        pp = ppgemm_tn_ws_floats(tokens, out_rows, out_cols, want_colsum);     // (an upper bound is all the callers need)
    if (tuning().ppgemm != 0 && dtype == RVT_BF16 && out_rows % 256 == 0 && out_cols % 64 == 0 && out_cols >= 256) {
        const size_t pc = ppgemm_tn_conv_ws_floats(tokens, out_rows, out_cols);  // rvt_conv_wgrad on the same kernel (K a multiple of 64 only)
        if (pc > pp) pp = pc;
    }
    int bn = wgrad_bn(out_cols);
    int bk = dtype == RVT_F32 ? TileGeom<float>::BK : TileGeom<bf16>::BK;
    size_t n = wgrad_ws_floats(out_rows, out_cols, tokens, bn, bk, want_colsum);
    if (out_rows <= 64) {                    // rvt_conv_wgrad may compute the transposed product (see there)
        size_t nt = wgrad_ws_floats(out_cols, out_rows, tokens, 64, bk, want_colsum);
        if (nt > n) n = nt;
    }
    return n > pp ? n : pp;
}
}  // extern "C"
// conv weight gradients that take the 256-wide token-contraction kernel (bf16, Cout % 256 == 0, k*k*Cin % 256 == 0, Cin % 64 == 0)
static bool conv_wgrad_tn(int dtype, int F, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    return tuning().ppgemm != 0 && tuning().conv_wgrad_tn != 0 && dtype == RVT_BF16 && ppgemm_tn_conv_shape_ok(F, H, W, Cin, Cout, k, stride, pad);
}
// the 3 x 3 / stride 2 / pad 1 down-sampling convs that take the 256-wide kernel with the im2col gather in its load stream
// (ppgemm.hpp GATHER = 2; bf16, Cin % 64 == 0, Cout % 256 == 0, even H and W, input below 2 GiB)
static bool conv_fwd_pp(int dtype, int F, int H, int W, int Cin, int Cout, int k, int stride, int pad) {
    if (dtype != RVT_BF16 || k != 3 || stride != 2 || pad != 1 || (H & 1) || (W & 1) || Cin % 64 != 0) return false;
    const long long M = (long long)F * (H / 2) * (W / 2);
    if ((long long)F * H * W * Cin * 2 >= 0x7ffffff0ll || M * Cout * 2 >= (1ll << 40) || M >= (1ll << 31)) return false;
    return tuning().conv_fwd_pp != 0 && use_ppgemm(dtype, (int)M, Cout, 9 * Cin, Cin, 9 * Cin, 1 << 30);
}
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
    if (conv_fwd_pp(dtype, F, H, W, Cin, Cout, k, stride, pad)) {
        // 256 x 256 tiles, im2col as a per-lane source address of the LDS-DMA load stream (ppgemm.hpp, GATHER = 2)
        PPConv cv;
        cv.Ho = H / 2; cv.Wo = W / 2; cv.Cout = Cin; cv.H = H; cv.W = W; cv.Cin = Cin;
        cv.dHoWo = FastDiv(cv.Ho * cv.Wo); cv.dWo = FastDiv(cv.Wo);
        for (int nt = 0; nt < 8; nt++) cv.taps[nt] = 0;
        const int M = F * cv.Ho * cv.Wo, K = 9 * Cin;
        const PPMat xs{(const bf16*)in, (const bf16*)in, Cin, 1 << 30}, ws{(const bf16*)w, (const bf16*)w, K, 1 << 30};
        const PPEpArgs ep{(bf16*)out, nullptr, nullptr, nullptr, nullptr, Cout};
        launch_ppgemm<PP_STORE, 2>(xs, ws, ep, M, Cout, K, st, cv);
        return check_launch("conv_fwd");
    }
    DISPATCH_DTYPE(dtype, {
        Im2colSrc<T> a = make_im2col<T>(in, F, H, W, Cin, k, stride, pad);
        PlainSrc<T> b{(const T*)w, a.cols, Cout, a.cols};
        EpStore<T> ep{(T*)out, Cout, nullptr, nullptr};
        DISPATCH_BN(Cout, (launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, a.rows, Cout, a.cols, 1, st)));
    });
    return check_launch("conv_fwd");
}

// inference-mode Conv2d + BatchNorm2d + SiLU in ONE launch: the BatchNorm affine (from the running statistics, rvt_bn_finalize with
// training = 0) and the activation run in the GEMM epilogue on the fp32 accumulator
int rvt_conv_bn_act_fwd(const void* in, const void* w, const float* scale, const float* shift, void* out, int dtype, int F, int H, int W,
                        int Cin, int Cout, int k, int stride, int pad, int act, void* stream) {
    RVT_CHECK(Cin % 8 == 0 && Cout % 8 == 0 && scale && shift && (act == 0 || act == 1), "conv_bn_act_fwd: channels must be multiples of 8, scale / shift required");
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DTYPE(dtype, {
        Im2colSrc<T> a = make_im2col<T>(in, F, H, W, Cin, k, stride, pad);
        PlainSrc<T> b{(const T*)w, a.cols, Cout, a.cols};
        EpAffineAct<T> ep{(T*)out, Cout, scale, shift, act};
        DISPATCH_BN(Cout, (launch_gemm<T, BN, false>(a, XfNone(), b, XfNone(), ep, a.rows, Cout, a.cols, 1, st)));
    });
    return check_launch("conv_bn_act_fwd");
}

int rvt_conv_wgrad(const void* in, const void* dy, float* dw, float* ws, int dtype, int F, int H, int W, int Cin, int Cout,
                   int k, int stride, int pad, void* stream) {
    RVT_CHECK(Cin % 8 == 0 && Cout % 8 == 0, "conv_wgrad: channels must be multiples of 8");
    hipStream_t st = (hipStream_t)stream;
    if (conv_wgrad_tn(dtype, F, H, W, Cin, Cout, k, stride, pad) && ws != nullptr) {
        // 256 x 256 tiles, both operands by LDS-DMA, im2col as a per-lane source address (ppgemm_tn.hpp, CONV)
        launch_ppgemm_tn_conv((const bf16*)dy, (const bf16*)in, dw, ws, F, H, W, Cin, Cout, k, stride, pad, st);
        return check_launch("conv_wgrad");
    }
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

// The same input gradient for the 3 x 3 / stride 2 / pad 1 convs of stages 2-4 as ONE product over 2 x 2 input-pixel blocks
// (ppgemm.hpp, GATHER): wd4 = [4 Cin][4 Cout] block-sparse weights (PACK_CONV_DGRAD4).  Measured against the four
// parity-class launches above: profiles/r3/microbench_conv_dgrad_r3{i,j,k,l}.txt.
int rvt_conv_dgrad4_supported(int dtype, int H, int W, int Cin, int Cout, int k, int stride, int pad, int F) {
    if (dtype != RVT_BF16 || k != 3 || stride != 2 || pad != 1 || (H & 1) || (W & 1)) return 0;
    if (Cin % 64 != 0 || Cin > 512 || Cout % 64 != 0) return 0;
    // every N tile must walk at least two K tiles (the u > 0 tile switch runs a MODE 1 step, then a MODE 2 step): a tile with a
    // single tap has Cout / 64 of them
    if (Cout < 128) return 0;
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
    for (int nt = 0; nt < 8; nt++) {
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

}  // extern "C"
