// extern "C" entry points, part 7 of 8: ConvLSTM cell, one launch per time step (GEMM engines).
#include "gemm_host.hpp"
#include "rowops.hpp"

extern "C" {
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

}  // extern "C"
