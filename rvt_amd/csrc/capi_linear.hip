// extern "C" entry points, part 3 of 8: linear layers (forward / input gradient / weight gradient) on the GEMM engines.
#include "gemm_host.hpp"
#include "ln_linear.hpp"

extern "C" {
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
    if (dtype == RVT_BF16 && K == 256 && N == 1024 && tuning().ln_linear != 0) {
        // fc1 of a C = 256 block: weight-stationary column groups (csrc/ln_linear.hpp), 8 groups x token streams, one workgroup per CU
        const int resident_override = tuning().chain_resident;
        const int total = resident_override > 0 ? resident_override : 256;
        int streams = imax(1, total / 8);
        const int want = ((M + 31) / 32 + 7) / 8;        // token streams that still get a tile per wave
        if (streams > want) streams = want;
        if (streams >= 8) streams -= streams % 8;        // (the XCD-aware placement wants a multiple of 8)
        if (gp != nullptr)
            hipLaunchKernelGGL((lin_gelu_ws_kernel<bf16, 256, 1024, 8, true>), dim3(8 * streams), dim3(512), 0, st, (const bf16*)x, (const bf16*)w, bias,
                               (bf16*)g, (bf16*)gp, M);
        else
            hipLaunchKernelGGL((lin_gelu_ws_kernel<bf16, 256, 1024, 8, false>), dim3(8 * streams), dim3(512), 0, st, (const bf16*)x, (const bf16*)w, bias,
                               (bf16*)g, (bf16*)nullptr, M);
        return check_launch("linear_gelu_fwd");
    }
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

}  // extern "C"
