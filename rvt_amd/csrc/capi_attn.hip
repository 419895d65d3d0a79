// extern "C" entry points, part 6 of 8: partition attention (attn.hpp, attn_core2.hpp) and the fused attention half (attn_block.hpp).
#include "host.hpp"
#include "attn.hpp"
#include "attn_block.hpp"
#include "attn_core2.hpp"

using namespace rvt;

extern "C" {
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
        if (bwd) {
            if constexpr (sizeof(T) == 2 && (NB == 2 || NB == 3)) {
                const size_t qb = (size_t)g.F * g.H * g.W * 3 * g.C * 2, ob = qb / 3;
                if (g.dh == 32 && qb < 0x7fff0000ull && tuning().attn_staged != 0) {
                    hipLaunchKernelGGL((attn_core_bwd_staged_kernel<NB, HG>), grid, dim3(64 * HG), 0, st, (const bf16*)qkv, (const bf16*)dout, (bf16*)out, g,
                                       (unsigned)qb, (unsigned)ob);
                    return;
                }
            }
            hipLaunchKernelGGL((attn_core_bwd_kernel<T, NB, HG>), grid, dim3(64 * HG), 0, st, (const T*)qkv, (const T*)dout, (T*)out, g);
        } else {
            if constexpr (sizeof(T) == 2 && (NB == 2 || NB == 3)) {
                // rows staged through LDS (attn_core_fwd_staged_kernel): dim_head 32, tensors below 2 GiB; tuning.attn_staged = 0: the direct kernel
                const size_t qb = (size_t)g.F * g.H * g.W * 3 * g.C * 2, ob = qb / 3;
                if (g.dh == 32 && qb < 0x7fff0000ull && tuning().attn_staged != 0) {
                    hipLaunchKernelGGL((attn_core_fwd_staged_kernel<NB, HG>), grid, dim3(64 * HG), 0, st, (const bf16*)qkv, (bf16*)out, g, (unsigned)qb, (unsigned)ob);
                    return;
                }
            }
            hipLaunchKernelGGL((attn_core_fwd_kernel<T, NB, HG>), grid, dim3(64 * HG), 0, st, (const T*)qkv, (T*)out, g);
        }
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
    const int resident_override = tuning().attn_block_resident;
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
                       (T*)dqkv, (T*)u_out, ln_w, ln_b, (const T*)wqkv, bqkv, (const T*)wpg_t, dln_w, dln_b, g, eps, (const T*)nullptr);
}
template <class T>
static void launch_ab_bwd_pre(const void* x, const void* y0, const void* dxmid, void* dy0, void* dqkv, const float* ln_w, const void* wqkv,
                              const float* bqkv, const void* wpg_t, float* dln_w, float* dln_b, const AttnGeom& g, float eps, hipStream_t st) {
    constexpr int WPB = AbWaves<T, 2>::V;
    auto k = attn_block_bwd_kernel<T, 64, 2, false, WPB, true>;
    hipLaunchKernelGGL(k, dim3(ab_grid(k, 64 * WPB, g.F * g.P, WPB)), dim3(64 * WPB), 0, st, (const T*)x, (const T*)dxmid, (T*)dy0,
                       (T*)dqkv, (T*)nullptr, ln_w, (const float*)nullptr, (const T*)wqkv, bqkv, (const T*)wpg_t, dln_w, dln_b, g, eps,
                       (const T*)y0);
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

int rvt_attn_block_bwd_preln(const void* x, const void* y0, const void* dxmid, void* dy0, void* dqkv, const float* ln_w,
                             const void* wqkv, const float* bqkv, const void* wpg_t, float* dln_w, float* dln_b, int dtype, int F,
                             int H, int W, int C, int dim_head, int ph, int pw, int window, float eps, void* stream) {
    RVT_CHECK(rvt_attn_block_supported(dtype, C, dim_head, ph * pw) && ph * pw <= 64,
              "attn_block_bwd_preln: not built for dtype=%d C=%d dim_head=%d L=%d", dtype, C, dim_head, ph * pw);
    RVT_CHECK(y0 != nullptr && ln_w != nullptr && dln_w != nullptr && dln_b != nullptr, "attn_block_bwd_preln: y0, ln_w, dln_w, dln_b are required");
    AttnGeom g;
    if (make_attn_geom(g, F, H, W, C, dim_head, ph, pw, window)) return 1;
    DISPATCH_DTYPE(dtype, (launch_ab_bwd_pre<T>(x, y0, dxmid, dy0, dqkv, ln_w, wqkv, bqkv, wpg_t, dln_w, dln_b, g, eps, (hipStream_t)stream)));
    return check_launch("attn_block_bwd_preln");
}

}  // extern "C"
