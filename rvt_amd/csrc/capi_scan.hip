// extern "C" entry points, part 8 of 8: ConvLSTM with the time loop inside the kernel (lstm_scan.hpp).
#include "host.hpp"
#include "gemm.hpp"
#include "lstm_scan.hpp"
#include "lstm_scan2.hpp"
#include "lstm_scan3.hpp"

using namespace rvt;

extern "C" {
// configurations built (waves per workgroup, weights resident in LDS or streamed from L2):
//   bf16: C = 32 (4 waves, LDS), 64 (8 fwd / 4 bwd waves, LDS), 128 (weights from L2);  f32 (parity): C = 32, 64, 128 from L2
int rvt_lstm_scan_supported(int dtype, int C) {
    if (dtype != RVT_BF16 && dtype != RVT_F32) return 0;
    return C == 32 || C == 64 || C == 128;
}
}  // extern "C"
template <class K> static int scan_grid(K kernel, int threads, int M, int tm) {
    const int resident_override = tuning().gemm_resident;
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
        FoldJobs fj;
        fj.add((const float*)ws, dw, grid, rec, (size_t)4 * C * 2 * C);
        fj.add((const float*)(ws + (size_t)4 * C * 2 * C), db, grid, rec, (size_t)4 * C, 0, NWM, (size_t)4 * C);     // NWM bias rows per record -> one output
        launch_fold_jobs(fj, st);
    }
}
extern "C" {
size_t rvt_lstm_scan_bwd_ws_floats(int dtype, int C, int M) {
    if (!scan_wgrad_built(dtype, C)) return 0;
    const int grid = C == 32 ? lstm_scan_bwd_grid<bf16, 32, 4, true, true>(M) : lstm_scan_bwd_grid<bf16, 64, 4, true, true>(M);
    const int NWM = 4 / (C / 32);
    size_t n = (size_t)grid * ((size_t)4 * C * 2 * C + (size_t)NWM * 4 * C);
    if (C == 64) {                          // the T-form kernel (lstm_scan2.hpp): one record per workgroup
        const size_t n2 = (size_t)scan_grid(lstm_scan2_bwd_kernel, 512, M, Scan2BwdSmem::TM) * Scan2BwdSmem::REC + 64;      // (+ 64: phase timers of the -DSCAN2_PROF measurement build)
        if (n2 > n) n = n2;
    }
    return n;
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
        else if (C == 64 && tuning().lstm_scan_v2) {          // T-form rebuild (lstm_scan2.hpp): one wave = 32 tokens, no barriers
            auto k = lstm_scan2_fwd_kernel<8>;
            hipLaunchKernelGGL(k, dim3(scan_grid(k, 512, M, 256)), dim3(512), 0, st, (const bf16*)x_all, (bf16*)Hall, c0, c_last,
                               (bf16*)Csave, (const bf16*)w, bias, M, T_steps);
        }
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
        else if (C == 64 && wgrad && tuning().lstm_scan_v2) {       // T-form rebuild (lstm_scan2.hpp)
            auto k = lstm_scan2_bwd_kernel;
            const int grid = scan_grid(k, 512, M, Scan2BwdSmem::TM);
            hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, st, (const bf16*)x_all, (const bf16*)Hall, (const bf16*)Csave, c0,
                               (const bf16*)dH, dc_last, (const bf16*)w, bias, (bf16*)dx_all, (bf16*)dh0, dc0, ws, M, T_steps);
            const size_t wn = (size_t)4 * C * 2 * C;
            FoldJobs fj;
            fj.add((const float*)ws, dw, grid, Scan2BwdSmem::REC, wn);
            fj.add((const float*)(ws + wn), db, grid, Scan2BwdSmem::REC, (size_t)4 * C);
            launch_fold_jobs(fj, st);
        }
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


// ---------------------------------------------------------------- wide stages: streamed weights, saved gates (lstm_scan3.hpp)
// RB of the forward tile (32 RB tokens per workgroup); the reverse scan always walks 32-token blocks
static int scan3_rb(int C) { return (C == 256 ? tuning().lstm_scan3_rb256 : tuning().lstm_scan3_rb128) == 2 ? 2 : 1; }
// tuning().lstm_scan3: bit 0 = C 256, bit 1 = C 128
int rvt_lstm_scan3_supported(int dtype, int C) {
    return dtype == RVT_BF16 && ((C == 256 && (tuning().lstm_scan3 & 1)) || (C == 128 && (tuning().lstm_scan3 & 2)));
}
int rvt_lstm_scan3_rows(int C, int M) {
    const int tm = 32 * scan3_rb(C);
    return (M + tm - 1) / tm * tm;
}
int rvt_lstm_scan3_pack(const void* w, void* wp_fwd, void* wtp_bwd, int C, void* stream) {
    RVT_CHECK(C % 64 == 0 && w != nullptr, "lstm_scan3_pack: C=%d must be a multiple of 64", C);
    hipStream_t st = (hipStream_t)stream;
    const int pieces = (C / 64) * (2 * C / 16) * 8 * 64;        // (same count in both directions)
    const int grid = imax(1, imin(1024, (pieces + 255) / 256));
    if (wp_fwd != nullptr) hipLaunchKernelGGL(lstm_scan3_pack_kernel<false>, dim3(grid), dim3(256), 0, st, (const bf16*)w, (bf16*)wp_fwd, C);
    if (wtp_bwd != nullptr) hipLaunchKernelGGL(lstm_scan3_pack_kernel<true>, dim3(grid), dim3(256), 0, st, (const bf16*)w, (bf16*)wtp_bwd, C);
    return check_launch("lstm_scan3_pack");
}
int rvt_lstm_scan3_fwd(const void* x_all, void* Hall, const float* c0, float* c_last, void* Csave, const void* wp, const float* bias,
                       void* gsave, int dtype, int M, int C, int T_steps, void* stream) {
    RVT_CHECK(rvt_lstm_scan3_supported(dtype, C), "lstm_scan3_fwd: not built for dtype=%d C=%d", dtype, C);
    RVT_CHECK(M >= 1 && T_steps >= 1 && (Csave == nullptr) == (gsave == nullptr), "lstm_scan3_fwd: empty problem, or only one of Csave / gsave");
    hipStream_t st = (hipStream_t)stream;
#define RVT_SCAN3_FWD(CC, RBB) do { auto k = lstm_scan3_fwd_kernel<CC, RBB>;                                                     \
        hipLaunchKernelGGL(k, dim3(scan_grid(k, CC, M, 32 * RBB)), dim3(CC), 0, st, (const bf16*)x_all, (bf16*)Hall, c0, c_last, \
                           (bf16*)Csave, (const bf16*)wp, bias, (bf16*)gsave, M, T_steps); } while (0)
    if (C == 256) { if (scan3_rb(C) == 2) RVT_SCAN3_FWD(256, 2); else RVT_SCAN3_FWD(256, 1); }
    else { if (scan3_rb(C) == 2) RVT_SCAN3_FWD(128, 2); else RVT_SCAN3_FWD(128, 1); }
#undef RVT_SCAN3_FWD
    return check_launch("lstm_scan3_fwd");
}
int rvt_lstm_scan3_bwd(const void* gsave, const void* Csave, const float* c0, const void* dH, const float* dc_last, const void* wtp,
                       void* dx_all, void* dz_all, void* dh0, float* dc0, int dtype, int M, int C, int T_steps, void* stream) {
    RVT_CHECK(rvt_lstm_scan3_supported(dtype, C), "lstm_scan3_bwd: not built for dtype=%d C=%d", dtype, C);
    RVT_CHECK(M >= 1 && T_steps >= 1 && gsave != nullptr && Csave != nullptr && dx_all != nullptr && dz_all != nullptr,
              "lstm_scan3_bwd: empty problem / missing buffers");
    hipStream_t st = (hipStream_t)stream;
#define RVT_SCAN3_BWD(CC) do { auto k = lstm_scan3_bwd_kernel<CC>;                                                                       \
        hipLaunchKernelGGL(k, dim3(scan_grid(k, CC, M, 32)), dim3(CC), 0, st, (const bf16*)gsave, (const bf16*)Csave, c0, (const bf16*)dH, \
                           dc_last, (const bf16*)wtp, (bf16*)dx_all, (bf16*)dz_all, (bf16*)dh0, dc0, M, T_steps, scan3_rb(C)); } while (0)
    if (C == 256) RVT_SCAN3_BWD(256); else RVT_SCAN3_BWD(128);
#undef RVT_SCAN3_BWD
    return check_launch("lstm_scan3_bwd");
}

}  // extern "C"
