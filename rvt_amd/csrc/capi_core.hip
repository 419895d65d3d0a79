// extern "C" entry points of librvt_hip.so (declared in include/rvt_hip.h), part 1 of 8: error / tuning plumbing and the
// row-wise operators.  Host-side only: argument checks, launch geometry.
#include "host.hpp"
#include "rowops.hpp"
#include "events.hpp"
#include "pack.hpp"
#include "bnact.hpp"

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

RvtTuning g_tuning = RVT_TUNING_DEFAULTS;
}  // namespace rvt

using namespace rvt;

namespace rvt {
// Measurement aid (bench.py `mfma_peak_sustained`): nothing but dependent-free v_mfma_f32_32x32x16_bf16 on four accumulator
// blocks per wave, 16 waves per CU - the rate the matrix pipes of THIS part sustain at the clock it actually runs under load
// (the 2.5 PFLOP/s figure assumes 2.4 GHz).
__global__ void __launch_bounds__(256) mfma_probe_kernel(float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    frag_t<bf16> a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = (bf16)(float)((lane + i) & 3); b[i] = (bf16)(float)((lane ^ i) & 1); }
    f32x16 c0, c1, c2, c3;
#pragma unroll
    for (int i = 0; i < 16; i++) { c0[i] = 0.f; c1[i] = 0.f; c2[i] = 0.f; c3[i] = 0.f; }
    for (int it = 0; it < iters; it++) {
        mma32(c0, a, b); mma32(c1, a, b); mma32(c2, a, b); mma32(c3, a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    if (s == -1.0f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;       // never true: keeps the MFMAs alive
}
}  // namespace rvt

extern "C" {

const char* rvt_last_error(void) { return g_err; }

int rvt_is_emulator(void) {
#ifdef RVT_EMU
    return 1;
#else
    return 0;
#endif
}

double rvt_probe_mfma(float* scratch, int iters, int workgroups, void* stream) {
    if (iters < 1 || workgroups < 1) return 0.0;
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, scratch, iters);
    return (double)workgroups * 4.0 /*waves*/ * (double)iters * 4.0 /*MFMAs*/ * (2.0 * 32 * 32 * 16);
}

void rvt_tuning_defaults(RvtTuning* t) {
    if (t) { const RvtTuning d = RVT_TUNING_DEFAULTS; *t = d; }
}
int rvt_get_tuning(RvtTuning* t) {
    RVT_CHECK(t != nullptr && t->struct_bytes == (int)sizeof(RvtTuning), "get_tuning: struct_bytes must be sizeof(RvtTuning) = %d", (int)sizeof(RvtTuning));
    *t = g_tuning;
    return 0;
}
int rvt_set_tuning(const RvtTuning* t) {
    RVT_CHECK(t != nullptr && t->struct_bytes == (int)sizeof(RvtTuning), "set_tuning: struct_bytes must be sizeof(RvtTuning) = %d", (int)sizeof(RvtTuning));
    RVT_CHECK(t->wgrad_bn == 0 || t->wgrad_bn == 64 || t->wgrad_bn == 128, "set_tuning: wgrad_bn %d not in {0, 64, 128}", t->wgrad_bn);
    RVT_CHECK(t->wgrad_slice_tokens >= 64, "set_tuning: wgrad_slice_tokens %d < 64", t->wgrad_slice_tokens);
    RVT_CHECK(t->stem_depth == 4 || t->stem_depth == 5, "set_tuning: stem_depth %d not in {4, 5}", t->stem_depth);
    RVT_CHECK(t->mlp_tm == 0 || t->mlp_tm == 64 || t->mlp_tm == 128, "set_tuning: mlp_tm %d not in {0, 64, 128}", t->mlp_tm);
    g_tuning = *t;
    return 0;
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
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pack_table_kernel<T>), dim3((total_blocks + PACK_LBPW - 1) / PACK_LBPW), dim3(256), 0, st,
                                             (const PackDesc*)descs, n_desc, (unsigned)total_blocks));
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

// ---------------------------------------------------------------- BatchNorm2d + SiLU of the PAFPN's BaseConv units (bnact.hpp)
int rvt_bn_stats(const void* x, float* sum, float* sumsq, int dtype, int rows, int C, void* stream) {
    RVT_CHECK(C % 8 == 0 && C >= 8 && C <= 1024 && rows >= 1, "bn_stats: C=%d must be a multiple of 8 in [8, 1024]", C);
    const int Gp = pow2_ge(C / 8), nrl = 256 / Gp;
    const int grid = imin(1024, imax(1, (rows + nrl * 8 - 1) / (nrl * 8)));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_stats_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, sum, sumsq, rows, C, Gp));
    return check_launch("bn_stats");
}
int rvt_bn_finalize(const float* sum, const float* sumsq, int rows, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, float* mean_out, float* rstd_out, float* scale, float* shift, int C,
                    int training, void* stream) {
    RVT_CHECK(C >= 1 && gamma && beta && scale && shift, "bn_finalize: bad arguments");
    RVT_CHECK(training ? (sum && sumsq && rows >= 1) : (running_mean && running_var), "bn_finalize: training needs sums, eval needs running statistics");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sum, sumsq, (float)rows, gamma, beta, eps, momentum,
                       running_mean, running_var, mean_out, rstd_out, scale, shift, C, training);
    return check_launch("bn_finalize");
}
// grid of the row-streaming element-wise kernels: 256 / Gp rows per workgroup pass, ~8 passes per workgroup, at most 4096 workgroups
static inline int bn_row_grid(int rows, int Gp) { const int nrl = 256 / Gp; return imin(4096, imax(1, (rows + nrl * 8 - 1) / (nrl * 8))); }
int rvt_bn_act_fwd(const void* x, const float* scale, const float* shift, void* y, int dtype, int rows, int C, int act, void* stream) {
    RVT_CHECK(C % 8 == 0 && C >= 8 && C <= 2048 && rows >= 1 && (act == 0 || act == 1), "bn_act_fwd: bad shape / activation");
    const int Gp = pow2_ge(C / 8);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_act_fwd_kernel<T>), dim3(bn_row_grid(rows, Gp)), dim3(256), 0, (hipStream_t)stream, (const T*)x, scale,
                                             shift, (T*)y, rows, C, Gp, act));
    return check_launch("bn_act_fwd");
}
int rvt_bn_train_act_fwd(const void* x, const float* sum, const float* sumsq, int count, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* mean_out, float* rstd_out, float* scale_out,
                         float* shift_out, void* y, int dtype, int rows, int C, int act, void* stream) {
    RVT_CHECK(C % 8 == 0 && C >= 8 && C <= 2048 && rows >= 1 && count >= 1 && (act == 0 || act == 1), "bn_train_act_fwd: bad shape / activation");
    RVT_CHECK(x && sum && sumsq && gamma && beta && mean_out && rstd_out && scale_out && shift_out && y, "bn_train_act_fwd: null argument");
    const int Gp = pow2_ge(C / 8);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_train_act_fwd_kernel<T>), dim3(bn_row_grid(rows, Gp)), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                                             sum, sumsq, (float)count, gamma, beta, eps, momentum, running_mean, running_var, mean_out, rstd_out,
                                             scale_out, shift_out, (T*)y, rows, C, Gp, act));
    return check_launch("bn_train_act_fwd");
}
int rvt_bn_act_bwd_stats(const void* dy, const void* x, const float* scale, const float* shift, const float* mean, const float* rstd,
                         float* dsum, float* dxsum, int dtype, int rows, int C, int act, void* stream) {
    RVT_CHECK(C % 8 == 0 && C >= 8 && C <= 1024 && rows >= 1, "bn_act_bwd_stats: C=%d must be a multiple of 8 in [8, 1024]", C);
    const int Gp = pow2_ge(C / 8), nrl = 256 / Gp;
    const int grid = imin(1024, imax(1, (rows + nrl * 8 - 1) / (nrl * 8)));
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_act_bwd_stats_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)x,
                                             scale, shift, mean, rstd, dsum, dxsum, rows, C, Gp, act));
    return check_launch("bn_act_bwd_stats");
}
int rvt_bn_act_bwd_apply(const void* dy, const void* x, const float* scale, const float* shift, const float* mean, const float* rstd,
                         const float* dsum, const float* dxsum, void* dx, int dtype, int rows, int C, int act, void* stream) {
    RVT_CHECK(C % 8 == 0 && C >= 8 && C <= 2048 && rows >= 1, "bn_act_bwd_apply: bad shape");
    const int Gp = pow2_ge(C / 8);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((bn_act_bwd_apply_kernel<T>), dim3(bn_row_grid(rows, Gp)), dim3(256), 0, (hipStream_t)stream, (const T*)dy,
                                             (const T*)x, scale, shift, mean, rstd, dsum, dxsum, (T*)dx, rows, C, Gp, 1.0f / (float)rows, act));
    return check_launch("bn_act_bwd_apply");
}

int rvt_state_reset_masked(void* st_, const unsigned char* mask, int dtype, int B, size_t per_sample, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int grid = grid_for((size_t)B * per_sample, 4096);
    DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((state_reset_kernel<T>), dim3(grid), dim3(256), 0, st, (T*)st_, mask, B,
                                             per_sample));
    return check_launch("state_reset_masked");
}

}  // extern "C"
