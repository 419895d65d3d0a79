/* C ABI of librvt_hip.so — the MI355X (gfx950) kernels behind the RVT recurrent backbone.
 *
 * The reference (uzh-rpg/RVT) is pure PyTorch and has no FFI of its own; these entry points are the
 * operator boundary a maintainer would bind (ctypes stub: rvt_amd/_lib.py).  Each
 * one replaces the ATen op sequence of the cited reference lines.  Plain pointers and sizes only:
 *   - every pointer is a DEVICE pointer unless stated otherwise;
 *   - `dtype` selects the storage / MFMA input type of activations and weights:
 *         RVT_F32  (0)  float   — parity mode, exact f32 MFMA
 *         RVT_BF16 (1)  bfloat16 — performance mode
 *     accumulators, LayerNorm/softmax statistics, the LSTM cell state `c`, biases, LayerNorm
 *     affine parameters, LayerScale gammas and ALL gradient outputs of parameters are float32;
 *   - activations are channels-last, token-major: X[frame][y][x][c] with frame = t*B + b;
 *   - `stream` is a hipStream_t; calls are asynchronous, allocate nothing and are graph-capturable;
 *   - return value 0 = ok, non-zero = error (message via rvt_last_error(), thread-local).
 * Channel counts must be multiples of 8; dim_head a multiple of 8 and <= 32; partition size <= 96.
 *
 * TWO TIERS (the operator granularity grew with the tuning history).
 *   Integration surface — what a reference-side maintainer or a non-Python host binds:
 *       rvt_last_error, rvt_tuning_defaults / rvt_get_tuning / rvt_set_tuning          one record of launch geometry + routing
 *       rvt_stage_seq_fwd (+ _ws_bytes)                one call per backbone stage and sequence (no-grad forward)
 *       rvt_stacked_histogram                          event stream -> input tensor (row f4)
 *       rvt_yolox_decode / rvt_simota_loss (+ _ws_bytes) / rvt_yolox_decode_bwd        detection tail (row f3)
 *       rvt_pack_table                                 all kernel-side weight layouts of a module, one launch per step
 *   Operator level — every other entry below: one launch each, what the stage driver sequences and what the Python mirror
 *       (rvt_amd/stage.py, the training backward) calls directly.  Stable and tested one by one (tests/test_kernels.py), but a host
 *       that only runs the model needs none of them; the `*_supported` / `*_ws_*` queries say which fused forms exist for a shape.
 */
#ifndef RVT_HIP_H
#define RVT_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { RVT_F32 = 0, RVT_BF16 = 1 };

const char* rvt_last_error(void);
/* 1 if this library is the CPU SIMT emulator build used by the unit tests, 0 for the gfx950 build */
int rvt_is_emulator(void);

/* Launch geometry and kernel routing, ONE explicit record instead of environment variables (round 4).  The defaults are
 * the production route: what bench.py times and what a deployment runs; nothing in the library reads the environment.
 * The unit tests install a "test geometry" (tiny persistent grids, every kernel family forced on at test-size problems;
 * rvt_amd/tuning.py: TEST_GEOMETRY) through the same call and say so, and the production-route parity tests run on the
 * defaults.  The record is process-wide (not per stream / per thread): set it between launches, not concurrently with them.
 * 0 in a geometry field means "sized by the library" (occupancy query x 256 CUs). */
typedef struct RvtTuning {
    int struct_bytes;         /* sizeof(RvtTuning) of the caller; set / get reject a mismatch */
    /* ---- launch geometry ---- */
    int gemm_resident;        /* persistent workgroups of the 128-row GEMM engine, fused MLP (mlp.hpp) and ConvLSTM scan kernels */
    int gemm_xcd_panels;      /* 1: panels of long NT tile walks are dealt per XCD (gemm.hpp tile_of mode 3) */
    int wgrad_bn;             /* output-tile width of the split-K weight-gradient kernel: 0 = by shape, 64, 128 */
    int wgrad_blocks;         /* target workgroups of a split-K weight-gradient launch (512 = two per CU) */
    int wgrad_slice_tokens;   /* minimum tokens per K slice (8192; >= 64) */
    int ppgemm;               /* 1: bf16 products whose shape fits take the 256x256 LDS-DMA kernels (ppgemm.hpp, ppgemm_tn.hpp) */
    int ppgemm_min_m;         /* ... from this many token rows (4096) */
    int ppgemm_all;           /* 1: every epilogue flavour takes ppgemm whatever the contraction length (production: GELU-dual from K >= 512) */
    int ppgemm_grid;          /* workgroups of a ppgemm launch (0 = one per CU, sized by the fullest XCD) */
    int ppgemm_tn_items;      /* token slices per output tile of ppgemm_tn (0 = fill the chip) */
    int one_per_cu_grid;      /* grid cap of the one-workgroup-per-CU kernels: stem forward / weight gradient, chain-MLP weight gradient (0 = 256) */
    int stem;                 /* 1: uint8 planes take the stem kernels (stem.hpp) */
    int stem_depth;           /* software-pipeline depth of the stem forward's plane loads: 4 or 5 */
    int mlp_tm;               /* token-tile height of the LDS-staged fused MLP at bf16 C = 64: 0 / 64, or 128 */
    int mlp_chain;            /* 1: C = 64 MLP halves take the register-chained kernels (mlp_chain.hpp) */
    int mlp_chain_wgrad;      /* 1: ... including the weight-gradient half */
    int chain_resident;       /* persistent workgroups of the chain-MLP / dgrad_ln kernels */
    int attn_block_resident;  /* persistent workgroups of the fused attention half (attn_block.hpp) */
    int dgrad_ln;             /* 1: rvt_linear_dgrad_ln_supported may say yes */
    /* ---- routes taken by the host-side stage driver (rvt_amd/stage.py reads them back through rvt_get_tuning) ---- */
    int route_fused_mlp;      /* -1: by measurement (C = 64 all halves, C = 128 forward); 0: op-by-op; 1: every supported half */
    int route_mlp_bwd_fused;  /* 1: MLP backward recomputed on chip from (dxout, xmid) where built */
    int route_attn_block;     /* 1: fused attention half where built (C = 64) */
    int route_lstm_scan;      /* -1: time loop in the kernel where the weights stay on chip; 0: one launch per step; 1: every built width */
    int route_lstm_scan_wgrad;/* 1: ConvLSTM weight gradients inside the reverse scan where built */
    int route_conv_dgrad4;    /* 1: 3x3 / stride-2 conv input gradient as one gather GEMM */
    int route_wgrad_stream;   /* 1: weight-gradient launches on a second HIP stream */
    int lstm_scan_v2;         /* 1: bf16 C = 64 ConvLSTM scans take the T-form kernels of lstm_scan2.hpp (0: lstm_scan.hpp, A/B) */
    int route_stage_driver;   /* 1: the no-grad forward of rvt_amd takes rvt_stage_seq_fwd (one call per stage); 0: the Python host loop */
    int route_mlp_store_pre;  /* 1: the LDS-staged fused MLP forward (C = 128) saves the pre-activation h only, not GELU(h) and GELU'(h); default 0: measured slower (GELU on load costs more than the bytes it saves, profiles/r4/microbench_mlp128.txt) */
    int route_mlp_bwd_both;   /* 1: stage-1 MLP backward = ONE launch (rvt_mlp_bwd_recompute_both) instead of dgrad + wgrad */
    int mlp_stream;           /* 1 (round 5): C = 128 MLP halves take the streamed-weight chain kernels (mlp_stream.hpp): nothing-saved forward,
                                 recompute backward (input-gradient + weight-gradient launch); 0: LDS-staged forward that saves GELU / GELU' + op-by-op backward */
    int ln_linear;            /* 1 (round 5): LayerNorm + qkv projection of a C = 128 block in one launch where rvt_ln_linear_supported, and fc1 + GELU at K = 256, N = 1024 on the weight-stationary kernel (both ln_linear.hpp) */
    int conv_wgrad_tn;        /* 1 (round 5): conv weight gradients with Cout % 256 == 0, Cin % 64 == 0 and k*k*Cin >= 256 take ppgemm_tn.hpp (im2col gather by LDS-DMA; the last 256-wide k tile may lie partly beyond K) */
    int attn_staged;          /* 1 (round 5): the partition-attention core of stages 2-4 stages its rows through LDS (whole-line requests) where built */
    int lstm_scan3;           /* round 6: ConvLSTM (bf16, dws_conv False) with the time loop in the kernel, weights streamed from L2 in operand order, gates saved for the reverse scan (lstm_scan3.hpp): bit 0 = at C 256, bit 1 = at C 128 (instead of lstm_scan.hpp's register-resident weights) */
    int lstm_scan3_rb256;     /* 32-token blocks per workgroup tile of that forward at C = 256: 1 or 2 */
    int lstm_scan3_rb128;     /* ... at C = 128 */
    int route_stage_driver_train; /* 1 (round 6): the training forward / backward of a stage take rvt_stage_seq_train_fwd / rvt_stage_seq_bwd (one call per stage and direction) where covered; 0: the Python host loop */
    int route_attn_preln;     /* 1 (round 6): the backward of a stage's first block also carries the gradient through the down-sampling LayerNorm where no token mask sits between them: rvt_attn_block_bwd_preln on the fused attention half, rvt_linear_dgrad_preln on the op-by-op route (C <= 128); 0: separate rvt_layernorm_bwd */
    int conv_fwd_pp;          /* 1 (round 6): 3 x 3 / stride 2 / pad 1 convs with Cin % 64 == 0 and Cout % 256 == 0 take the 256-wide kernel with the im2col gather in its load stream (ppgemm.hpp GATHER = 2); 0: the 128-row engine */
} RvtTuning;
#define RVT_TUNING_DEFAULTS {(int)sizeof(RvtTuning), 0, 1, 0, 512, 8192, 1, 4096, 0, 0, 0, 0, 1, 4, 0, 1, 1, 0, 0, 1, -1, 1, 1, -1, 1, 1, 0, 1, 1, 0, 1, 1, 1, 1, 1, 3, 1, 1, 1, 1, 1}
void rvt_tuning_defaults(RvtTuning* t);        /* fills *t with the production defaults */
int rvt_get_tuning(RvtTuning* t);              /* t->struct_bytes must be set by the caller */
int rvt_set_tuning(const RvtTuning* t);

/* Measurement aid for bench.py (no reference counterpart): launches `workgroups` x 256 threads that issue nothing but
 * independent v_mfma_f32_32x32x16_bf16 (`iters` x 4 per wave) and returns the FLOPs the launch executes; timed with HIP
 * events it gives the bf16 MFMA rate this part SUSTAINS at its clock under load (the 2.5 PFLOP/s peak assumes 2.4 GHz).
 * scratch: >= workgroups * 256 floats (never written). */
double rvt_probe_mfma(float* scratch, int iters, int workgroups, void* stream);

/* Weight-gradient GEMMs (rvt_*_wgrad) cut the token contraction into K slices.  `ws` is their scratch for the
 * two-stage reduction: rvt_wgrad_workspace_floats(dtype, out_rows, out_cols, tokens, want_colsum) float32 elements
 * (out = dw's [rows][cols]; conv: [Cout][k*k*Cin]; lstm: [4C][2C]).  ws == NULL selects direct float atomics, which
 * are correct but slow on MI355X (device-scope atomics execute memory-side). */
size_t rvt_wgrad_workspace_floats(int dtype, int out_rows, int out_cols, int tokens, int want_colsum);

/* Event-tensor cast + zero pad (modules/detection.py:133-134, utils/padding.py:29-44) fused with the
 * NCHW -> channels-last repack: src [F][Cin][h][w] (uint8 if src_u8 else float32) ->
 * dst [F][H][W][Cp] (dtype), zero padded to H>=h, W>=w, Cp>=Cin. */
int rvt_prepack_input(const void* src, int src_u8, void* dst, int dtype, int F, int Cin, int h, int w,
                      int H, int W, int Cp, void* stream);

/* Down-sampling conv (maxvit.py:160-168,175; bias-free): in [F][H][W][Cin], w [Cout][k*k*Cin]
 * (tap-major, cin fastest) -> out [F][Ho][Wo][Cout], Ho=(H+2*pad-k)/stride+1.  bf16, k = 3, stride 2, pad 1, even H and W,
 * Cin % 64 == 0, Cout % 256 == 0 (stages 3-4 of RVT-Base): the 256-wide LDS-DMA kernel with the im2col gather in its load
 * stream (tuning.conv_fwd_pp; same products, fp32 summation order of that kernel); everything else: the 128-row engine. */
int rvt_conv_fwd(const void* in, const void* w, void* out, int dtype, int F, int H, int W, int Cin, int Cout,
                 int k, int stride, int pad, void* stream);
/* Input gradient of the 3x3 / stride-2 / pad-1 down-sampling conv of stages 2-4 (reference maxvit.py:160-168, autograd) as ONE
 * product over 2x2 blocks of input pixels (csrc/ppgemm.hpp, GATHER mode).  wd4: [4*Cin][4*Cout] weights in the block-sparse layout
 * of PACK_CONV_DGRAD4 (csrc/pack.hpp); add (nullable): cotangent already attached to the input, same shape as din.
 * rvt_conv_dgrad4_supported: bf16, k 3, stride 2, pad 1, even H and W, Cin and Cout multiples of 64, enough rows. */
int rvt_conv_dgrad4_supported(int dtype, int H, int W, int Cin, int Cout, int k, int stride, int pad, int F);
int rvt_conv_dgrad4(const void* dy, const void* wd4, const void* add, void* din, int dtype, int F, int H, int W, int Cin, int Cout,
                    void* stream);

/* Input gradient: din[F][H][W][Cin] = conv^T(dy) (+ add).  wd = class-packed transposed weights, see
 * rvt_conv_dgrad_weight_elems / rvt_amd.weights.pack_conv_dgrad: for each parity class (py,px) in
 * row-major order a [Cin][nky*nkx*Cout] matrix.  `add` (nullable) has din's shape. */
int rvt_conv_dgrad(const void* dy, const void* wd, const void* add, void* din, int dtype, int F, int H, int W,
                   int Cin, int Cout, int k, int stride, int pad, void* stream);
/* Weight gradient: dw[Cout][k*k*Cin] (float32) += dy^T im2col(in). */
int rvt_conv_wgrad(const void* in, const void* dy, float* dw, float* ws, int dtype, int F, int H, int W, int Cin,
                   int Cout, int k, int stride, int pad, void* stream);

/* The stem on the loader's uint8 planes (replaces rvt_prepack_input + rvt_conv_fwd + rvt_layernorm_fwd, and
 * rvt_conv_wgrad, for the first stage: maxvit.py:160-177 on the cast + padded input of modules/detection.py:133-134 and
 * utils/padding.py:29-44).  src [F][Cin][h][w] uint8, zero padded to H x W on the fly; w = the packed conv weight of
 * rvt_conv_fwd ([64][7*7*cp], tap-major, cin fastest, cp >= Cin); y0 = conv output, x = LayerNorm(y0), both
 * [F][Ho][Wo][64].  rvt_stem_supported: bf16, uint8 planes, 7x7 / stride 4 / pad 3, Cout = 64, Cin <= 20, w % 4 == 0. */
int rvt_stem_supported(int dtype, int src_u8, int Cin, int Cout, int k, int stride, int pad, int w);
int rvt_stem_fwd(const void* src, const void* w, const float* ln_w, const float* ln_b, void* y0, void* x, int dtype, int F,
                 int Cin, int cp, int h, int wd, int H, int W, float eps, void* stream);
/* dw[64][7*7*cp] (float32, the layout of rvt_conv_wgrad) += dy^T im2col(src); ws: rvt_stem_wgrad_ws_floats floats. */
size_t rvt_stem_wgrad_ws_floats(int Cin, int F, int H, int W);
int rvt_stem_wgrad(const void* src, const void* dy, float* dw, float* ws, int dtype, int F, int Cin, int cp, int h, int wd,
                   int H, int W, void* stream);

/* LayerNorm over channels (maxvit.py:172,177,229,241). */
int rvt_layernorm_fwd(const void* x, const float* w, const float* b, void* y, int dtype, int rows, int C,
                      float eps, void* stream);
/* dx = LN'(dy) (+ dres, nullable);  dw[C] += ..., db[C] += ... */
int rvt_layernorm_bwd(const void* x, const float* w, const void* dy, const void* dres, void* dx, float* dw,
                      float* db, int dtype, int rows, int C, float eps, void* stream);

/* y[M][N] = f(x)[M][K] W[N][K]^T + bias (nullable);  f = exact GELU if gelu_in else identity
 * (maxvit.py:347,353 and the MLP 100-118). */
int rvt_linear_fwd(const void* x, const void* w, const float* bias, void* y, int dtype, int M, int N, int K,
                   int gelu_in, void* stream);
/* MLP fc1 with its activation (maxvit.py:100-112): g = GELU(x W^T + bias) and, if gp != NULL, gp = GELU'(x W^T + bias)
 * (saved for backward so that no kernel re-evaluates erf).  Route note: bf16 at K = 256, N = 1024 (any M) takes the
 * weight-stationary kernel of ln_linear.hpp, whose GELU / GELU' come from the nearest-entry pair table of gelu_lut.hpp
 * (tests/test_kernels.py::test_linear_gelu_weight_stationary); every other shape evaluates erf. */
int rvt_linear_gelu_fwd(const void* x, const void* w, const float* bias, void* g, void* gp, int dtype, int M, int N, int K,
                        void* stream);
/* y = res + gamma * (f(x) W^T + bias)   — LayerScale + residual (maxvit.py:51-53,268-269). */
int rvt_linear_scale_res_fwd(const void* x, const void* w, const float* bias, const float* gamma, const void* res,
                             void* y, int dtype, int M, int N, int K, int gelu_in, void* stream);
/* dx[M][K] = dy[M][N] Wt[K][N]^T, then at most one of:  * gelu'(gelu_pre[M][K]),  + add[M][K],  * mul[M][K]
 * (all nullable). */
int rvt_linear_dgrad(const void* dy, const void* wt, const void* gelu_pre, const void* add, const void* mul, void* dx,
                     int dtype, int M, int N, int K, void* stream);
/* Input gradient of a linear layer whose input is a LayerNorm output, with the LayerNorm backward (+ residual cotangent) in
 * the epilogue (reference maxvit.py:229,347 norm1 -> qkv and :241,100-118 norm2 -> fc1, autograd; csrc/dgrad_ln.hpp):
 *   dx[M][C] = add + LN'(dy[M][K] w[K][C]; x[M][C]),  dln_w[C] += sum_tok (dy w) * xhat,  dln_b[C] += sum_tok (dy w).
 * w = the forward weight in its natural [K][C] layout (compute dtype), add nullable.
 * rvt_linear_dgrad_ln_supported: bf16, C in {64, 128}, K = 3C or 4C. */
int rvt_linear_dgrad_ln_supported(int dtype, int C, int K);
int rvt_linear_dgrad_ln(const void* dy, const void* w, const void* x, const void* add, void* dx, const float* ln_w,
                        float* dln_w, float* dln_b, int dtype, int M, int C, int K, float eps, void* stream);
/* The same launch for a stage's FIRST block (no norm1, maxvit_rnn.py:153), carried through the down-sampling norm in front of it
 * (maxvit.py:177): dy0 = LN'(dy w + add ; y0) - the added cotangent (the block's residual path) enters the norm.  Replaces
 * rvt_linear_dgrad(.., add) + rvt_layernorm_bwd(y0, ..).  Shapes as rvt_linear_dgrad_ln_supported; add is required. */
int rvt_linear_dgrad_preln(const void* dy, const void* w, const void* y0, const void* add, void* dy0, const float* ln_w,
                           float* dln_w, float* dln_b, int dtype, int M, int C, int K, float eps, void* stream);

/* u = LN(x; ln_w, ln_b), y = u W^T + bias in one launch (csrc/ln_linear.hpp; replaces rvt_layernorm_fwd + rvt_linear_fwd for
 * `self.qkv(self.norm1(x))`, reference maxvit.py:268 -> :347).  W [N][C] row-major, bias may be NULL, u may be NULL (no-grad
 * forward: the normalised rows are not kept); ln_w = ln_b = NULL: no LayerNorm (the first block behind a down-sampling conv,
 * maxvit.py:229-236), u is not written.  rvt_ln_linear_supported: bf16, C = 128, N = 384 (tuning.ln_linear = 0: no). */
int rvt_ln_linear_supported(int dtype, int C, int N);
int rvt_ln_linear_fwd(const void* x, const float* ln_w, const float* ln_b, const void* w, const float* bias, void* u, void* y,
                      int dtype, int M, int C, int N, float eps, void* stream);

/* dw[N][K] (float32) += dy[M][N]^T f(x)[M][K];  if dy_colsum != NULL also dy_colsum[N] += column sums of dy
 * (the bias gradient), computed from the tiles the kernel streams anyway. */
int rvt_linear_wgrad(const void* dy, const void* x, float* dw, float* dy_colsum, float* ws, int dtype, int M, int N,
                     int K, int gelu_in, void* stream);

/* Fused MLP half of a block (maxvit.py:269 + :100-118), built for the HBM-bound stages:
 * rvt_mlp_fused_supported(dtype, C) != 0  (bf16: C in {64,128}; f32: C == 64).
 *   rvt_mlp_fwd:  xout = xmid + gamma * (GELU(LN(xmid) W1^T + b1) W2^T + b2) in one pass; if g_out/gp_out are non-NULL
 *                 also g = GELU(h), gp = GELU'(h) [M][4C] for backward; g_out alone (gp_out NULL) receives the PRE-ACTIVATION
 *                 h = LN(xmid) W1^T + b1 instead (half the bytes; the backward applies GELU / GELU' on load through
 *                 rvt_linear_wgrad(gelu_in) and rvt_linear_dgrad(gelu_pre)); and if v2_out is non-NULL the LayerNorm output
 *                 LN(xmid) [M][C] (B operand of the fc1 weight gradient); nothing else of the chain reaches HBM.
 *   rvt_mlp_bwd_dgrad: dh = (dxout (W2*gamma)) * gp;  dxmid = dxout + LN'(dh W1; xmid);  dln_w/dln_b += LayerNorm
 *                 parameter gradients.  w2g_t = (W2*gamma[:,None])^T stored [4C][C]; w1_t = W1^T stored [C][4C].
 * w1 [4C][C], w2 [C][4C] (dtype); ln/bias/gamma and dln_* float32. */
int rvt_mlp_fused_supported(int dtype, int C);
int rvt_mlp_fwd(const void* xmid, void* xout, void* g_out, void* gp_out, void* v2_out, const float* ln_w, const float* ln_b,
                const void* w1, const float* b1, const void* w2, const float* b2, const float* gamma, int dtype, int M,
                int C, float eps, void* stream);
int rvt_mlp_bwd_dgrad(const void* dxout, const void* gp, const void* xmid, void* dh, void* dxmid, const float* ln_w,
                      const void* w2g_t, const void* w1_t, float* dln_w, float* dln_b, int dtype, int M, int C, float eps,
                      void* stream);

/* Backward of the MLP half with NOTHING but the block input saved (rvt_mlp_bwd_fused_supported: C == 64): LN2, fc1, GELU
 * and GELU' are recomputed on chip, and besides dxmid = dxout + LN2'(...) the kernel accumulates the weight gradients in
 * registers across its persistent tile walk:  dw1[4C][C] += dh^T LN2(xmid),  db1[4C] += colsum(dh),
 * s2[C][4C] += dxout^T GELU(h),  cs2[C] += colsum(dxout)  (s2 / cs2 are the RAW fc2 products: LayerScale is folded in by
 * rvt_layerscale_grad_table),  dln_w / dln_b += LayerNorm parameter gradients.  w1 [4C][C]; w2g_t = (W2*gamma[:,None])^T
 * [4C][C]; w1_t = W1^T [C][4C].  ws: rvt_mlp_bwd_fused_ws_floats(dtype, C, M) floats (per-workgroup partials). */
int rvt_mlp_bwd_fused_supported(int dtype, int C);
size_t rvt_mlp_bwd_fused_ws_floats(int dtype, int C, int M);
/* The same kernel cut in two along the critical path of backward (both recompute LN2 / fc1 / GELU from xmid):
 *   rvt_mlp_bwd_recompute_dgrad: dxmid and dln_w / dln_b only — no weight-gradient accumulators, several workgroups per CU;
 *   rvt_mlp_bwd_recompute_wgrad: dw1, db1, s2, cs2 only (two groups of hidden chunks per tile column, 64 accumulator
 *     registers each) — for the weight-gradient stream; reads (dxout, xmid) once per group instead of the 10 rows of C
 *     per token the two weight-gradient GEMMs re-read. */
int rvt_mlp_bwd_recompute_dgrad(const void* dxout, const void* xmid, void* dxmid, const float* ln_w, const float* ln_b,
                                const void* w1, const float* b1, const void* w2g_t, const void* w1_t, float* dln_w,
                                float* dln_b, int dtype, int M, int C, float eps, void* stream);
int rvt_mlp_bwd_recompute_wgrad(const void* dxout, const void* xmid, const float* ln_w, const float* ln_b, const void* w1,
                                const float* b1, const void* w2g_t, float* dw1, float* db1, float* s2, float* cs2, float* ws,
                                int dtype, int M, int C, float eps, void* stream);
/* Round 4: both of the above in ONE launch where rvt_mlp_bwd_both_supported (bf16, C = 64, RvtTuning.route_mlp_bwd_both): the
 * weight-gradient kernel's dh tile goes through LDS to two rotating waves that form dh W1, and the LayerNorm backward + residual
 * runs in the staging role of all threads — one recompute of fc1 / GELU / GELU' and one read of xmid / dxout instead of two. */
int rvt_mlp_bwd_both_supported(int dtype, int C);
int rvt_mlp_bwd_recompute_both(const void* dxout, const void* xmid, void* dxmid, const float* ln_w, const float* ln_b, const void* w1,
                               const float* b1, const void* w2g_t, const void* w1_t, float* dln_w, float* dln_b, float* dw1, float* db1,
                               float* s2, float* cs2, float* ws, int dtype, int M, int C, float eps, void* stream);

/* Partitioned multi-head attention core (maxvit.py:252-265,273-304,343-354 minus the two linears):
 * qkv [F*H*W][3C] in image token order, per-head layout [q|k|v]; out [F*H*W][C].  window=1: ph x pw
 * windows; window=0: dilated grid with grid size (ph,pw). */
int rvt_attn_fwd(const void* qkv, void* out, int dtype, int F, int H, int W, int C, int dim_head, int ph, int pw,
                 int window, void* stream);
int rvt_attn_bwd(const void* qkv, const void* dout, void* dqkv, int dtype, int F, int H, int W, int C, int dim_head,
                 int ph, int pw, int window, void* stream);

/* Fused attention half of a PartitionAttentionCl block (replaces maxvit.py:268 = norm1 :229, SelfAttentionCl :343-354 on
 * the partitions of :273-304, LayerScale :51-53 and the residual): one wave per partition, everything between the x rows
 * and the xmid rows on chip (csrc/attn_block.hpp).  rvt_attn_block_supported: C == 64, dim_head == 32, 32 < ph*pw <= 96.
 *   fwd: xmid = x + gamma * (attention(LN1(x) wqkv^T + bqkv) wp^T + bp); ln_w / ln_b NULL = no norm1 (first window block of a
 *        stage, maxvit_rnn.py:153); a_out (nullable) receives the attention output rows [F*H*W][C] (operand of the proj
 *        weight gradient).  wqkv [3C][C] (rows [head][q|k|v][dh]), wp [C][C].
 *   bwd: dx = dxmid + LN1'(dqkv wqkv), dqkv = attention backward of dxmid (gamma wp) with q / k / v / P recomputed from x;
 *        writes dqkv [F*H*W][3C] and (u_out nullable, LN only) LN1(x) for the qkv weight-gradient GEMM; dln_w / dln_b +=
 *        LayerNorm parameter gradients.  wpg_t = (wp * gamma[:,None])^T [C][C]. */
int rvt_attn_block_supported(int dtype, int C, int dim_head, int L);
int rvt_attn_block_fwd(const void* x, void* xmid, void* a_out, const float* ln_w, const float* ln_b, const void* wqkv,
                       const float* bqkv, const void* wp, const float* bp, const float* gamma, int dtype, int F, int H, int W,
                       int C, int dim_head, int ph, int pw, int window, float eps, void* stream);
int rvt_attn_block_bwd(const void* x, const void* dxmid, void* dx, void* dqkv, void* u_out, const float* ln_w,
                       const float* ln_b, const void* wqkv, const float* bqkv, const void* wpg_t, float* dln_w, float* dln_b,
                       int dtype, int F, int H, int W, int C, int dim_head, int ph, int pw, int window, float eps,
                       void* stream);
/* The backward of a stage's FIRST block (no norm1: its input x = LN_pre(y0) is the output of the down-sampling norm,
 * maxvit.py:177, maxvit_rnn.py:153) carried through that norm in the same launch (round 6): dy0 = LN_pre'(dxmid + dqkv wqkv ; y0)
 * is written instead of dx, ln_w is LN_pre's weight, dln_w / dln_b += its parameter gradients (autograd of maxvit.py:177).
 * Replaces rvt_attn_block_bwd(ln_w = NULL) followed by rvt_layernorm_bwd(y0, ..): same dqkv bits, one row per token less to
 * write and two less to read.  Partitions of at most 64 tokens (two 32-token blocks), like rvt_attn_block_bwd. */
int rvt_attn_block_bwd_preln(const void* x, const void* y0, const void* dxmid, void* dy0, void* dqkv, const float* ln_w,
                             const void* wqkv, const float* bqkv, const void* wpg_t, float* dln_w, float* dln_b, int dtype, int F,
                             int H, int W, int C, int dim_head, int ph, int pw, int window, float eps, void* stream);

/* ConvLSTM cell with 1x1 conv (rnn.py:52-67): mix = [x|h_prev] Wp^T + bp with gate-interleaved rows
 * (row n' = (c/8)*32 + gate*8 + c%8, gates f,i,o,g); writes h_out [M][C], c_out [M][C] (float32) and,
 * if gates != NULL, the activated gates [M][4C] in natural order [f|i|o|g]. */
int rvt_lstm_fwd(const void* x, const void* h_prev, const float* c_prev, const void* w_perm, const float* b_perm,
                 void* h_out, float* c_out, void* gates, int dtype, int M, int C, void* stream);
/* BPTT element-wise part: consumes dh_in (+ dh_rec nullable), updates dc_rec in place, writes dz [M][4C]. */
int rvt_lstm_gates_bwd(const void* dh_in, const void* dh_rec, float* dc_rec, const void* gates, const float* c_new,
                       const float* c_prev, void* dz, int dtype, int M, int C, void* stream);
/* [dx | dh_rec] = dz W : wt = W^T [2C][4C] natural gate order. */
int rvt_lstm_dgrad(const void* dz, const void* wt, void* dx, void* dh_rec, int dtype, int M, int C, void* stream);
/* dw[4C][2C] (float32) += dz^T [x | h_prev];  dz_colsum[4C] += column sums of dz if non-NULL (bias gradient). */
int rvt_lstm_wgrad(const void* dz, const void* x, const void* h_prev, float* dw, float* dz_colsum, float* ws, int dtype,
                   int M, int C, void* stream);

/* The same ConvLSTM cell (rnn.py:43-67) with the TIME LOOP INSIDE the kernel, for the 1x1-conv variant (dws_conv False: the
 * recurrence is independent per pixel): one launch per stage runs all T_steps of modules/detection.py:131-148 for that stage
 * with h / c on chip.  rvt_lstm_scan_supported(dtype, C) != 0 for C in {32, 64, 128}.
 *   fwd: x_all [T][M][C]; Hall [T+1][M][C], slot 0 = incoming h (caller-filled), slots 1..T written; c0 fp32 [M][C] or
 *        NULL (zeros); c_last fp32 [M][C]; Csave [T][M][C] (dtype; slot t = c_t, what BPTT needs) or NULL (inference);
 *        w [4C][2C] in the reference's NATURAL order (rows f,i,o,g; columns [x|h]); bias fp32 [4C].
 *   bwd: reverse scan with the gates recomputed; dH [T][M][C] cotangent of Hall[1..] (NULL = zeros), dc_last fp32 [M][C]
 *        (NULL = zeros); wt = w^T [2C][4C]; writes dx_all [T][M][C], dz_all [T][M][4C] (pre-activation gradients, natural
 *        gate order, operand of rvt_lstm_wgrad), dh0 [M][C], dc0 fp32 [M][C]. */
int rvt_lstm_scan_supported(int dtype, int C);
/*        bf16, C = 128 (rvt_lstm_scan_saves_gates): the weights live in the register file; with gates_out != NULL the forward
 *        also stores the activated gates [T][M][4C] (natural order f,i,o,g), and the backward given `gates` reads them instead
 *        of recomputing (W^T in registers; x_all / Hall unused; dz_all written for rvt_lstm_wgrad). */
int rvt_lstm_scan_saves_gates(int dtype, int C);
int rvt_lstm_scan_fwd(const void* x_all, void* Hall, const float* c0, float* c_last, void* Csave, const void* w,
                      const float* bias, void* gates_out, int dtype, int M, int C, int T_steps, void* stream);
/*        With dw != NULL (rvt_lstm_scan_bwd_ws_floats(dtype, C, M) > 0: bf16, C <= 64) the weight gradients are accumulated
 *        inside the kernel: dw [4C][2C] += dz^T [x | h_prev], db [4C] += colsum(dz) (natural gate order, fp32), through
 *        per-workgroup partial records in ws; dz_all is then neither written nor needed (may be NULL). */
size_t rvt_lstm_scan_bwd_ws_floats(int dtype, int C, int M);
int rvt_lstm_scan_bwd(const void* x_all, const void* Hall, const void* Csave, const float* c0, const void* dH,
                      const float* dc_last, const void* w, const void* wt, const float* bias, void* dx_all, void* dz_all,
                      void* dh0, float* dc0, float* dw, float* db, float* ws, const void* gates, int dtype, int M, int C,
                      int T_steps, void* stream);

/* ---- ConvLSTM of the WIDE stages with the time loop in the kernel (csrc/lstm_scan3.hpp; reference models/layers/rnn.py:43-67 over
 * the loop of modules/detection.py:131-148, and its BPTT).  bf16, dws_conv False, C = 128 / 256 (rvt_lstm_scan3_supported).  The weights do
 * not fit on chip: they are streamed from L2 every step in MFMA-operand order, which rvt_lstm_scan3_pack produces once per
 * optimizer step from the natural [4C][2C] matrix (gate order f,i,o,g, input order [x | h]: rnn.py:52-61):
 *   wp_fwd  [C/64][2C/16][8][64][8]   (wave, k-step, (channel block, gate), lane, element)    = 4C * 2C elements
 *   wtp_bwd [C/64][4C/16][4][64][8]   (wave, k-step, (channel block, x | h part), lane, element) = rows of W^T, same size
 * Forward: x_all [T][M][C], Hall [T+1][M][C] (slot 0 = incoming h, filled by the caller; slots 1.. written), c0 fp32 [M][C] or
 * NULL (zeros, rnn.py:43-47), c_last fp32 [M][C].  For BPTT it saves the ACTIVATED gates (gsave) and a bf16 copy of the cell
 * states (Csave) in a private register-dump order; both buffers hold T * rvt_lstm_scan3_rows(C, M) * 4C / * C elements and are
 * NULL together for a no-grad forward.  Backward: reads them back (same M, same tuning), dH [T][M][C] = cotangent of Hall[1..]
 * (NULL = zeros), dc_last fp32 [M][C] (NULL = zeros); writes dx_all [T][M][C], dz_all [T][M][4C] (natural gate order: the operand
 * of rvt_lstm_wgrad), dh0 [M][C], dc0 fp32 [M][C]. */
int rvt_lstm_scan3_supported(int dtype, int C);
int rvt_lstm_scan3_rows(int C, int M);
int rvt_lstm_scan3_pack(const void* w, void* wp_fwd, void* wtp_bwd, int C, void* stream);
int rvt_lstm_scan3_fwd(const void* x_all, void* Hall, const float* c0, float* c_last, void* Csave, const void* wp, const float* bias,
                       void* gsave, int dtype, int M, int C, int T_steps, void* stream);
int rvt_lstm_scan3_bwd(const void* gsave, const void* Csave, const float* c0, const void* dH, const float* dc_last, const void* wtp,
                       void* dx_all, void* dz_all, void* dh0, float* dc0, int dtype, int M, int C, int T_steps, void* stream);

/* ---- stage-major driver (rvt_stage_seq_fwd) -------------------------------------------------------------
 * One backbone stage (reference maxvit_rnn.py:169-182: down-sampling conv + LayerNorm, the window and grid attention blocks,
 * the ConvLSTM) over ALL T time steps of B sequences in ONE call, for the NO-GRAD forward: validation and streaming inference
 * (modules/detection.py:231-255 with T = 1 per call).  The kernel routing and the per-step ConvLSTM loop that rvt_amd/stage.py
 * runs in Python happen inside the library; it launches exactly the operators declared in this header, on `stream`.
 * (The training forward / backward keep their host loop in rvt_amd/stage.py: the saved-activation bookkeeping lives there.)
 * All pointers in the descriptors are DEVICE pointers except `blocks`, a HOST array of 2 * num_blocks records
 * (window block, grid block, window, grid, ...).  Unsupported here (call the operators instead): token masks, DWS-ConvLSTM. */
typedef struct RvtBlockWeights {          /* one PartitionAttentionCl block, maxvit.py:193-270 */
    const float *n1_w, *n1_b;             /* norm1 (NULL, NULL: Identity - the first window block of a stage) */
    const void* qkv_w; const float* qkv_b;      /* [3C][C] (dtype), [3C] */
    const void* proj_w; const float* proj_b;    /* [C][C], [C] */
    const float* g1;                      /* LayerScale 1 [C] */
    const float *n2_w, *n2_b;             /* norm2 */
    const void* fc1_w; const float* fc1_b;      /* [4C][C], [4C] */
    const void* fc2_w; const float* fc2_b;      /* [C][4C], [C] */
    const float* g2;                      /* LayerScale 2 [C] */
} RvtBlockWeights;
typedef struct RvtStageDesc {
    int struct_bytes;                     /* sizeof(RvtStageDesc) */
    int dtype, C, Cin, cin_pad;           /* cin_pad: channel count of the packed conv weight / prepacked input (multiple of 8) */
    int H_in, W_in, k, stride, pad;       /* conv geometry on the (padded) input resolution */
    int ph, pw, dim_head, num_blocks;
    float eps;
    int inp_u8, h_raw, w_raw;             /* stage 1 fed by the loader's uint8 planes [F][Cin][h_raw][w_raw] (cast + zero pad fused in) */
    const void* conv_w;                   /* [C][k*k*cin_pad] tap-major (rvt_conv_fwd layout) */
    const float *ln_w, *ln_b;             /* LayerNorm after the conv */
    const RvtBlockWeights* blocks;        /* HOST array, 2 * num_blocks entries */
    const void* lstm_w; const float* lstm_b;      /* gate-interleaved rows (rvt_lstm_fwd) */
    const void* lstm_wn; const float* lstm_bn;    /* natural row order (rvt_lstm_scan_fwd) */
} RvtStageDesc;
/* inp: [T*B][H_in][W_in][cin_pad] channels-last (dtype), or the uint8 planes when inp_u8;  h0 [B][H][W][C] (dtype) / c0 fp32,
 * both NULL = zero state;  Hall [T+1][B][H][W][C]: slot 0 is scratch for the incoming h, slots 1..T receive h_t (the stage's
 * output features = the next stage's input);  c_last fp32 [B][H][W][C];  ws: rvt_stage_seq_fwd_ws_bytes(desc, T, B) bytes. */
size_t rvt_stage_seq_fwd_ws_bytes(const RvtStageDesc* desc, int T, int B);
int rvt_stage_seq_fwd(const RvtStageDesc* desc, const void* inp, const void* h0, const float* c0, void* Hall, float* c_last,
                      void* ws, size_t ws_bytes, int T, int B, void* stream);

/* ---- training-side stage driver (rvt_stage_seq_bwd; round 6) ----------------------------------------------
 * The TRAINING forward and the BPTT backward of one stage (reference maxvit_rnn.py:169-182 under autograd, driven by
 * modules/detection.py:131-148) as ONE library call each: the per-operator launches that rvt_amd/stage.py issued from Python
 * (40 - 150 per stage and direction) are sequenced here.  Division of labour: the HOST decides the kernel routes (one place:
 * rvt_amd/stage.py) and owns every tensor that outlives the call - the activations kept for backward (RvtBlockSaved and the
 * stage-level pointers below) and the gradient buckets; the driver owns the order of launches and the backward's temporaries
 * (carved from `ws`).  Nothing is launched that the operator entry points above do not launch.  Not covered (the host keeps its
 * operator-by-operator loop): token masks, the DWS-ConvLSTM, the LDS-staged fused-MLP flavours that save GELU / GELU'. */
typedef struct RvtBlockSaved {            /* activations of one block kept for backward; NULL = not kept on the chosen route */
    const void* xin;                      /* block input [M][C] (= previous block's xout, or the LayerNorm output of the down-sampling) */
    void* u;                              /* norm1(xin) (op-by-op attention with a norm1) */
    void* qkv;                            /* [M][3C] (op-by-op attention) */
    void* a;                              /* attention output rows [M][C] (operand of the proj weight gradient) */
    void* xmid;                           /* [M][C] */
    void* v2;                             /* norm2(xmid) (op-by-op MLP) */
    void* hg; void* hgp;                  /* GELU(h), GELU'(h) [M][4C] (op-by-op MLP) */
    void* xout;                           /* [M][C] */
} RvtBlockSaved;
typedef struct RvtBlockTrain {            /* backward-side operands and fp32 gradient accumulators (+=) of one block */
    const void *qkv_wt, *proj_wt, *fc1_wt, *fc2_wt;      /* W^T copies; proj / fc2 with LayerScale folded in (rvt_amd/weights.py) */
    float *d_n1_w, *d_n1_b;               /* NULL without norm1 */
    float *d_qkv_w, *d_qkv_b, *d_S1, *d_cs1;             /* S1 = dxmid^T a, cs1 = colsum(dxmid): raw proj products (LayerScale fold later) */
    float *d_n2_w, *d_n2_b, *d_fc1_w, *d_fc1_b, *d_S2, *d_cs2;
} RvtBlockTrain;
typedef struct RvtStageTrain {
    int struct_bytes;                     /* sizeof(RvtStageTrain) */
    /* routes, decided by the host */
    int attn_block;                       /* 1: fused attention half (rvt_attn_block_fwd / _bwd) */
    int ln_linear;                        /* 1: norm1 + qkv in one launch (rvt_ln_linear_fwd) */
    int mlp_route;                        /* 0: LayerNorm, fc1 + GELU / GELU', fc2 as separate launches; 1: recompute route (rvt_mlp_fwd keeps nothing) */
    int mlp_bwd_both;                     /* route 1: rvt_mlp_bwd_recompute_both instead of _dgrad + _wgrad */
    int dgrad_ln_qkv, dgrad_ln_fc1;       /* 1: that input gradient + the LayerNorm backward behind it in one launch (rvt_linear_dgrad_ln) */
    int lstm_route;                       /* 0: one launch per step; 1: the rvt_lstm_scan_ kernels, gates recomputed; 2: the same with saved gates; 3: the rvt_lstm_scan3_ kernels */
    int lstm_scan_wgrad;                  /* routes 1: ConvLSTM weight gradients inside the reverse scan */
    int conv_dgrad4;                      /* 1: rvt_conv_dgrad4 for the input gradient of the down-sampling conv */
    int attn_preln;                       /* the first block's backward also carries the gradient through the down-sampling norm: rvt_attn_block_bwd_preln (attn_block = 1) or rvt_linear_dgrad_preln (op-by-op attention, where rvt_linear_dgrad_ln_supported(C, 3C)) */
    const RvtBlockSaved* saved;           /* HOST arrays, 2 * num_blocks entries each */
    const RvtBlockTrain* tb;
    void *y0, *x0;                        /* conv output, LayerNorm output (= saved[0].xin) */
    void* Hall;                           /* [T+1][B][H][W][C]; slot 0 = incoming h, filled by the host */
    float* c_last;                        /* [B][H][W][C] */
    void* Csave;                          /* scan routes: cell-state copies (layout of the scan kernel in use) */
    void* gates;                          /* routes 0, 2, 3: activated gates */
    float* Call;                          /* route 0: [T+1][B][H][W][C] fp32; slot 0 = incoming c, filled by the host */
    const float* c0_saved;                /* scan routes, backward: the incoming cell state as the forward saw it (NULL = zeros) */
    const void *lstm_wp3, *lstm_wtp3;     /* route 3: rvt_lstm_scan3_pack outputs */
    const void *lstm_wt, *conv_wd4, *conv_wd;
    float *d_lstm_w, *d_lstm_b, *d_ln_w, *d_ln_b, *d_raw_conv;
} RvtStageTrain;
/* inp / h0 / c0 as rvt_stage_seq_fwd.  No workspace: every buffer the forward writes is named in `tr`. */
int rvt_stage_seq_train_fwd(const RvtStageDesc* desc, const RvtStageTrain* tr, const void* inp, const float* c0, int T, int B,
                            void* stream);
/* dH [T][B][H][W][C] cotangent of Hall[1..] (route 0: must not be NULL), dc_last fp32 (NULL = zeros), prev_cot: cotangent already
 * attached to the stage's input frames (added to the conv input gradient; NULL = none), d_in [T*B][H_in][W_in][Cin] (NULL: no
 * input gradient), dh0 [B][H][W][C], dc0 fp32.  `inp` = the tensor the forward read (planes or channels-last frames).
 * ws: rvt_stage_seq_bwd_ws_bytes bytes.  Parameter gradients are ACCUMULATED (+=) into the pointers of `tr`. */
size_t rvt_stage_seq_bwd_ws_bytes(const RvtStageDesc* desc, const RvtStageTrain* tr, int T, int B);
int rvt_stage_seq_bwd(const RvtStageDesc* desc, const RvtStageTrain* tr, const void* inp, const void* dH, const float* dc_last,
                      const void* prev_cot, void* d_in, void* dh0, float* dc0, void* ws, size_t ws_bytes, int T, int B, void* stream);

/* Depth-wise k x k conv (k = 3; groups = channels, padding k/2, stride 1) of the DWS-ConvLSTM (rnn.py:25-29,50-54)
 * on channels-last maps: y[n][y][x][c] = b[c] + sum_taps w[c][ky][kx] x[...][c].  x / y rows have pitch ldx / ldy
 * elements (so a C-wide slice of a 2C-wide buffer can be addressed).  transpose=1 computes the input gradient
 * (mirrored taps, bias ignored).  w: float32 [C][k*k], b: float32 [C]. */
int rvt_dwconv_fwd(const void* x, int ldx, const float* w, const float* b, void* y, int ldy, int dtype, int N, int H,
                   int W, int C, int k, int transpose, void* stream);
/* dw[C][k*k] += correlate(x, dy), db[C] += sum dy  (float32). */
int rvt_dwconv_wgrad(const void* x, int ldx, const void* dy, int ldy, float* dw, float* db, int dtype, int N, int H,
                     int W, int C, int k, void* stream);

/* Token masking of stage 1 (maxvit_rnn.py:174-176): rows of x[M][C] with mask[m] != 0 are overwritten by the float32
 * mask token;  backward: dtoken[C] += sum of dx over masked rows, and those rows of dx are zeroed in place. */
int rvt_token_mask_fwd(void* x, const unsigned char* mask, const float* token, int dtype, int M, int C, void* stream);
int rvt_token_mask_bwd(void* dx, const unsigned char* mask, float* dtoken, int dtype, int M, int C, void* stream);

/* Labelled-frame gather of the training step (modules/utils/detection.py:32-46, BackboneFeatureSelector): dst[n] = src[idx[n]]
 * over whole frames of frame_bytes (a multiple of 16) for n < n_sel; idx: int32 frame indices (t*B + b), on the device.
 * scatter = 1 is the backward: dst[idx[n]] = src[n] (dst zero-filled by the caller; indices distinct). */
int rvt_gather_frames(const void* src, const int* idx, void* dst, int n_sel, size_t frame_bytes, int scatter, void* stream);

/* ---- YOLOX PAFPN building block (reference models/detection/yolox/models/network_blocks.py:29-53) ----
 * BaseConv = Conv2d(bias=False) -> BatchNorm2d(eps 1e-5, momentum 0.1) -> SiLU on channels-last maps x[rows = N*H*W][C].  The conv
 * is rvt_conv_fwd / rvt_conv_dgrad / rvt_conv_wgrad; these are the row-wise kernels around it (csrc/bnact.hpp).  act: 0 none, 1 SiLU.
 *   rvt_bn_stats:          sum[C] += column sums of x, sumsq[C] += column sums of x^2 (fp32; zero them first; under data parallelism
 *                          these two vectors are what SyncBatchNorm all-reduces, train.py:133)
 *   rvt_bn_finalize:       training: mean = sum / rows, var = sumsq / rows - mean^2 (biased), running statistics updated in place with
 *                          `momentum` (unbiased variance) when non-NULL; eval: mean / var = the running statistics.  Writes
 *                          scale = gamma * rstd, shift = beta - mean * scale and (nullable) mean_out / rstd_out for the backward.
 *   rvt_bn_act_fwd:        y = act(x * scale + shift)            (y may alias x)
 *   rvt_bn_act_bwd_stats:  dz = dy * act'(x * scale + shift);  dsum[C] += sum dz (= dbeta), dxsum[C] += sum dz * xhat (= dgamma)
 *   rvt_bn_act_bwd_apply:  dx = scale * (dz - dsum / rows - xhat * dxsum / rows)      (training-mode BatchNorm backward) */
/* Inference: Conv2d(bias=False) + BatchNorm2d (running statistics) + activation in ONE launch — scale / shift from rvt_bn_finalize
 * (training = 0) applied to the fp32 accumulator in the GEMM epilogue; act 0 = none, 1 = SiLU.  in / w / out as rvt_conv_fwd. */
int rvt_conv_bn_act_fwd(const void* in, const void* w, const float* scale, const float* shift, void* out, int dtype, int F, int H, int W,
                        int Cin, int Cout, int k, int stride, int pad, int act, void* stream);
int rvt_bn_stats(const void* x, float* sum, float* sumsq, int dtype, int rows, int C, void* stream);
int rvt_bn_finalize(const float* sum, const float* sumsq, int rows, const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var, float* mean_out, float* rstd_out, float* scale, float* shift, int C,
                    int training, void* stream);
int rvt_bn_act_fwd(const void* x, const float* scale, const float* shift, void* y, int dtype, int rows, int C, int act, void* stream);
/* rvt_bn_finalize (training) + rvt_bn_act_fwd in ONE launch: count = rows the sums cover (the global count under synchronised BatchNorm) */
int rvt_bn_train_act_fwd(const void* x, const float* sum, const float* sumsq, int count, const float* gamma, const float* beta, float eps,
                         float momentum, float* running_mean, float* running_var, float* mean_out, float* rstd_out, float* scale_out,
                         float* shift_out, void* y, int dtype, int rows, int C, int act, void* stream);
int rvt_bn_act_bwd_stats(const void* dy, const void* x, const float* scale, const float* shift, const float* mean, const float* rstd,
                         float* dsum, float* dxsum, int dtype, int rows, int C, int act, void* stream);
int rvt_bn_act_bwd_apply(const void* dy, const void* x, const float* scale, const float* shift, const float* mean, const float* rstd,
                         const float* dsum, const float* dxsum, void* dx, int dtype, int rows, int C, int act, void* stream);

/* YOLOX head tail on [B][A][5 + num_classes] fp32 prediction tensors (rvt_amd/csrc/simota.hpp; host mirror rvt_amd/head.py).
 * Replaces models/detection/yolox/models/yolo_head.py:236-290 (decode) and :291-606 (get_losses / get_assignments /
 * simota_matching: the per-image Python loop with its int(nlabel[b]) / .item() host syncs and per-ground-truth topk) by a fixed
 * sequence of launches with no host synchronisation.  Anchors are ordered level by level, row-major inside a level.
 *   rvt_yolox_decode:     one FPN level.  reg_obj [B*H*W][ld_ro] (columns 0-3 box regression, 4 objectness logit) and
 *                         cls [B*H*W][ld_cls] (class logits) of `dtype` -> rows anchor_offset.. of pred_train (decoded cx cy w h,
 *                         raw logits; NULL = skip) and pred_infer (decoded box, sigmoid scores; NULL = skip), both [B][A][5+nc].
 *   rvt_yolox_decode_bwd: d_reg_obj / d_cls = gradient of pred_train's rows of this level, columns scaled by
 *                         col_scale[0] (box) / [1] (objectness) / [2] (class) read from DEVICE memory; padding columns zeroed.
 *   rvt_simota_loss:      labels [B][G][5] fp32 rows (class, cx, cy, w, h), zero rows = padding AFTER the real ones (:309).
 *                         losses[5] (device) = loss, 5 * iou loss, objectness loss, class loss, num_fg / max(num_gts, 1);
 *                         g_pred (NULL = skip) = d(sum iou) / d box, d(sum obj) / d objectness, d(sum cls) / d class, each divided
 *                         by max(num_fg, 1): the caller scales the three column groups (5 * dloss + d(iou loss) ...);
 *                         match_out [B][A] (NULL = skip) = matched ground-truth row or -1, piou_out [B][A] = IoU with it.
 *                         use_l1 (an option no RVT config turns on) is not implemented. */
int rvt_yolox_decode(const void* reg_obj, const void* cls, int ld_ro, int ld_cls, int dtype, int B, int H, int W, int stride,
                     int num_classes, int anchor_offset, int A, float* pred_train, float* pred_infer, void* stream);
int rvt_yolox_decode_bwd(const float* g_pred, const float* pred_train, const float* col_scale, void* d_reg_obj, void* d_cls, int ld_ro,
                         int ld_cls, int dtype, int B, int H, int W, int stride, int num_classes, int anchor_offset, int A, void* stream);
size_t rvt_simota_ws_bytes(int B, int G, int A);
int rvt_simota_loss(const float* pred_train, const float* labels, const int* level_hw, const int* level_stride, int L, int B, int G, int A,
                    int num_classes, float* losses, float* g_pred, int* match_out, float* piou_out, void* ws, size_t ws_bytes,
                    void* stream);

/* Zero state rows of samples with mask[b] != 0 (modules/utils/detection.py:96-113).
 * st is [B][per_sample] of float32 (is_f32) or `dtype`. */
int rvt_state_reset_masked(void* st, const unsigned char* mask, int dtype, int B, size_t per_sample, void* stream);

/* Parameter-side tables (rvt_amd/csrc/pack.hpp; host mirror rvt_amd/weights.py).  One launch walks an array of
 * descriptors in DEVICE memory:
 *   rvt_pack_table: element-wise gathers fp32 parameter -> kernel-side layout (cast, transpose with LayerScale folded in,
 *     tap-major conv weights, stride-parity conv-dgrad panels, gate-interleaved ConvLSTM rows) and the accumulating unpack
 *     of the raw conv weight gradient — what the reference leaves to autograd / .to() / permute (maxvit.py:51-53,160-168,
 *     rnn.py:52-61).  total_blocks = sum over descriptors of ceil(n / 1024).
 *   rvt_layerscale_grad_table: dW += gamma*S, db += gamma*cs, dgamma += rowsum(W*S) + b*cs for the proj / fc2 linears
 *     whose input-gradient weights carry LayerScale.  total_blocks = sum of C (one block per output channel). */
int rvt_pack_table(const void* descs, int n_desc, int total_blocks, int dtype, void* stream);
int rvt_layerscale_grad_table(const void* descs, int n_desc, int total_blocks, void* stream);

/* Event stream -> stacked histogram, the uint8 event tensor the backbone consumes (data/utils/representations.py:76-117,
 * StackedHistogram.construct): x, y, pol (0/1), time are int64 [n_events], time sorted ascending.
 * out[(pol*bins + t_idx)][y][x] = min(count, count_cutoff) as uint8 [2*bins][H][W], with the reference's accumulator
 * wrap-around (uint8 if fastmode else int16).  scratch: 2*bins*H*W uint32 (zeroed here).  Bit-exact. */
int rvt_stacked_histogram(const long long* x, const long long* y, const long long* pol, const long long* time,
                          size_t n_events, int bins, int H, int W, int count_cutoff, int fastmode, unsigned* scratch,
                          unsigned char* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
