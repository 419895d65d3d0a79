// extern "C" entry points, part 11: the TRAINING-side stage driver (rvt_stage_seq_bwd; round 6) — the training
// forward and the BPTT backward of one backbone stage (reference maxvit_rnn.py:169-182 under autograd, over the loop of
// modules/detection.py:131-148) as one call each.  The launch sequences are the ones rvt_amd/stage.py (stage_seq_forward with
// save=True / stage_seq_backward) issues from Python, in the same order, with the same arguments: outputs are bit-identical
// (tests/test_stage_driver.py).  The host picks the routes and owns everything that outlives a call (saved activations, gradient
// buckets); this file owns the order of launches and the backward's temporaries.  Nothing is launched that the operator entry
// points do not launch.
#include "host.hpp"

using namespace rvt;

namespace {
struct TCarver {                      // bump allocator over the caller's workspace; pieces are 2-MiB aligned like the caching allocator's
    char* p; size_t left; bool ok = true;                 // large blocks (256-byte aligned pieces measured 0.15 ms per step slower at RVT-Base)
    static constexpr size_t ALIGN = (size_t)1 << 21;
    void* take(size_t bytes) {
        const size_t mis = (size_t)(reinterpret_cast<uintptr_t>(p) & (ALIGN - 1));
        if (mis) { const size_t skip = ALIGN - mis; if (skip > left) { ok = false; return nullptr; } p += skip; left -= skip; }
        bytes = (bytes + ALIGN - 1) & ~(ALIGN - 1);
        if (bytes > left) { ok = false; return nullptr; }
        void* r = p; p += bytes; left -= bytes;
        return r;
    }
};
static inline size_t t_elt(int dtype) { return dtype == RVT_F32 ? 4 : 2; }
static inline int t_conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }
static inline size_t zmax(size_t a, size_t b) { return a > b ? a : b; }

struct TrainSizes {
    int H, W, M; size_t e, act, state;
    size_t wgrad_floats, mlp_floats, scan_floats, stem_floats;
};
static TrainSizes train_sizes(const RvtStageDesc& d, const RvtStageTrain& t, int T, int B) {
    TrainSizes s;
    s.H = t_conv_out(d.H_in, d.k, d.stride, d.pad); s.W = t_conv_out(d.W_in, d.k, d.stride, d.pad);
    s.e = t_elt(d.dtype);
    const size_t tok = (size_t)T * B * s.H * s.W;
    s.M = (int)tok; s.act = tok * d.C * s.e; s.state = (size_t)B * s.H * s.W * d.C;
    const int C = d.C, M = s.M;
    size_t w = rvt_wgrad_workspace_floats(d.dtype, 4 * C, 2 * C, M, 1);
    w = zmax(w, rvt_wgrad_workspace_floats(d.dtype, C, 4 * C, M, 1));
    w = zmax(w, rvt_wgrad_workspace_floats(d.dtype, 4 * C, C, M, 1));
    w = zmax(w, rvt_wgrad_workspace_floats(d.dtype, C, C, M, 1));
    w = zmax(w, rvt_wgrad_workspace_floats(d.dtype, 3 * C, C, M, 1));
    w = zmax(w, rvt_wgrad_workspace_floats(d.dtype, C, d.k * d.k * d.cin_pad, M, 0));
    s.wgrad_floats = zmax(w, (size_t)1 << 20);
    s.mlp_floats = t.mlp_route == 1 ? rvt_mlp_bwd_fused_ws_floats(d.dtype, C, M) : 0;
    s.scan_floats = (t.lstm_route == 1 && t.lstm_scan_wgrad) ? rvt_lstm_scan_bwd_ws_floats(d.dtype, C, B * s.H * s.W) : 0;
    s.stem_floats = d.inp_u8 ? rvt_stem_wgrad_ws_floats(d.Cin, T * B, d.H_in, d.W_in) : 0;
    return s;
}
static size_t train_bwd_ws_bytes(const RvtStageDesc& d, const RvtStageTrain& t, int T, int B) {
    const TrainSizes s = train_sizes(d, t, T, B);
    const size_t pad = (size_t)2 << 21;            // (2-MiB aligned pieces: up to one alignment skip + one round-up each)
    size_t n = 3 * (s.act + pad);                  // dx ring (block cotangents)
    n += 4 * s.act + pad;                          // dz of the ConvLSTM, then dh of the op-by-op MLP
    n += 3 * s.act + pad;                          // dqkv
    n += 2 * (s.act + pad);                        // da / du / dv2, dy0
    n += s.state * s.e + pad;                      // per-step route: second dh buffer
    n += (s.wgrad_floats + s.mlp_floats + s.scan_floats + s.stem_floats) * 4 + 4 * pad;
    return n + pad;
}
static bool desc_ok(const RvtStageDesc* d, const RvtStageTrain* t) {
    return d != nullptr && t != nullptr && d->struct_bytes == (int)sizeof(RvtStageDesc) && t->struct_bytes == (int)sizeof(RvtStageTrain) &&
           d->blocks != nullptr && t->saved != nullptr && d->num_blocks >= 0;
}
}  // namespace

#define RVT_TRY(call) do { if ((call) != 0) return 1; } while (0)

extern "C" {

int rvt_stage_seq_train_fwd(const RvtStageDesc* dp, const RvtStageTrain* tp, const void* inp, const float* c0, int T, int B, void* stream) {
    RVT_CHECK(desc_ok(dp, tp), "stage_seq_train_fwd: bad descriptors (struct_bytes %d / %d expected)", (int)sizeof(RvtStageDesc), (int)sizeof(RvtStageTrain));
    const RvtStageDesc& d = *dp; const RvtStageTrain& t = *tp;
    RVT_CHECK(T >= 1 && B >= 1 && inp != nullptr && t.Hall != nullptr && t.c_last != nullptr && t.y0 != nullptr && t.x0 != nullptr,
              "stage_seq_train_fwd: bad arguments");
    const int C = d.C, F = T * B, dt = d.dtype;
    const TrainSizes s = train_sizes(d, t, T, B);
    const int H = s.H, W = s.W, M = s.M;
    RVT_CHECK(H % d.ph == 0 && W % d.pw == 0, "stage_seq_train_fwd: %dx%d not divisible by the partition %dx%d", H, W, d.ph, d.pw);
    RVT_CHECK((size_t)M * 4 * C < ((size_t)1 << 31), "stage_seq_train_fwd: %d token rows exceed the operators' 32-bit sizes", M);
    hipStream_t st = (hipStream_t)stream;

    // ---- down-sampling conv + LayerNorm (maxvit.py:174-178) ----
    if (d.inp_u8) {
        RVT_CHECK(rvt_stem_supported(dt, 1, d.Cin, C, d.k, d.stride, d.pad, d.w_raw), "stage_seq_train_fwd: uint8 planes need the stem kernels (the host prepacks otherwise)");
        RVT_TRY(rvt_stem_fwd(inp, d.conv_w, d.ln_w, d.ln_b, t.y0, t.x0, dt, F, d.Cin, d.cin_pad, d.h_raw, d.w_raw, d.H_in, d.W_in, d.eps, stream));
    } else {
        RVT_TRY(rvt_conv_fwd(inp, d.conv_w, t.y0, dt, F, d.H_in, d.W_in, d.cin_pad, C, d.k, d.stride, d.pad, stream));
        RVT_TRY(rvt_layernorm_fwd(t.y0, d.ln_w, d.ln_b, t.x0, dt, M, C, d.eps, stream));
    }
    // ---- attention blocks (maxvit.py:267-270): window, then grid ----
    const void* x = t.x0;
    for (int bi = 0; bi < 2 * d.num_blocks; bi++) {
        const RvtBlockWeights& bw = d.blocks[bi];
        const RvtBlockSaved& sv = t.saved[bi];
        const int window = (bi & 1) == 0;
        RVT_CHECK(sv.xin == x && sv.xmid != nullptr && sv.xout != nullptr && sv.a != nullptr, "stage_seq_train_fwd: block %d buffers inconsistent", bi);
        if (t.attn_block) {
            RVT_TRY(rvt_attn_block_fwd(x, sv.xmid, sv.a, bw.n1_w, bw.n1_b, bw.qkv_w, bw.qkv_b, bw.proj_w, bw.proj_b, bw.g1, dt, F, H, W, C,
                                       d.dim_head, d.ph, d.pw, window, d.eps, stream));
        } else {
            RVT_CHECK(sv.qkv != nullptr && (bw.n1_w == nullptr || sv.u != nullptr), "stage_seq_train_fwd: block %d misses qkv / u", bi);
            if (t.ln_linear) {
                RVT_TRY(rvt_ln_linear_fwd(x, bw.n1_w, bw.n1_b, bw.qkv_w, bw.qkv_b, bw.n1_w != nullptr ? sv.u : nullptr, sv.qkv, dt, M, C, 3 * C, d.eps, stream));
            } else {
                const void* uu = x;
                if (bw.n1_w != nullptr) { RVT_TRY(rvt_layernorm_fwd(x, bw.n1_w, bw.n1_b, sv.u, dt, M, C, d.eps, stream)); uu = sv.u; }
                RVT_TRY(rvt_linear_fwd(uu, bw.qkv_w, bw.qkv_b, sv.qkv, dt, M, 3 * C, C, 0, stream));
            }
            RVT_TRY(rvt_attn_fwd(sv.qkv, sv.a, dt, F, H, W, C, d.dim_head, d.ph, d.pw, window, stream));
            RVT_TRY(rvt_linear_scale_res_fwd(sv.a, bw.proj_w, bw.proj_b, bw.g1, x, sv.xmid, dt, M, C, C, 0, stream));
        }
        if (t.mlp_route == 1) {
            RVT_TRY(rvt_mlp_fwd(sv.xmid, sv.xout, nullptr, nullptr, nullptr, bw.n2_w, bw.n2_b, bw.fc1_w, bw.fc1_b, bw.fc2_w, bw.fc2_b, bw.g2, dt, M, C,
                                d.eps, stream));
        } else {
            RVT_CHECK(sv.v2 != nullptr && sv.hg != nullptr && sv.hgp != nullptr, "stage_seq_train_fwd: block %d misses v2 / hg / hgp", bi);
            RVT_TRY(rvt_layernorm_fwd(sv.xmid, bw.n2_w, bw.n2_b, sv.v2, dt, M, C, d.eps, stream));
            RVT_TRY(rvt_linear_gelu_fwd(sv.v2, bw.fc1_w, bw.fc1_b, sv.hg, sv.hgp, dt, M, 4 * C, C, stream));
            RVT_TRY(rvt_linear_scale_res_fwd(sv.hg, bw.fc2_w, bw.fc2_b, bw.g2, sv.xmid, sv.xout, dt, M, C, 4 * C, 0, stream));
        }
        x = sv.xout;
    }
    // ---- ConvLSTM over the T steps (rnn.py:43-67); Hall slot 0 = incoming h (host) ----
    const int Ms = B * H * W;
    char* const HallB = (char*)t.Hall;
    if (t.lstm_route == 3) {
        RVT_CHECK(t.lstm_wp3 != nullptr && t.Csave != nullptr && t.gates != nullptr, "stage_seq_train_fwd: lstm_scan3 buffers missing");
        RVT_TRY(rvt_lstm_scan3_fwd(x, t.Hall, c0, t.c_last, t.Csave, t.lstm_wp3, d.lstm_bn, t.gates, dt, Ms, C, T, stream));
    } else if (t.lstm_route == 1 || t.lstm_route == 2) {
        RVT_CHECK(t.Csave != nullptr && (t.lstm_route == 1 || t.gates != nullptr), "stage_seq_train_fwd: lstm_scan buffers missing");
        RVT_TRY(rvt_lstm_scan_fwd(x, t.Hall, c0, t.c_last, t.Csave, d.lstm_wn, d.lstm_bn, t.lstm_route == 2 ? t.gates : nullptr, dt, Ms, C, T, stream));
    } else {
        RVT_CHECK(t.Call != nullptr && t.gates != nullptr, "stage_seq_train_fwd: per-step ConvLSTM buffers missing");
        for (int ts = 0; ts < T; ts++)
            RVT_TRY(rvt_lstm_fwd((const char*)x + (size_t)ts * s.state * s.e, HallB + (size_t)ts * s.state * s.e, t.Call + (size_t)ts * s.state,
                                 d.lstm_w, d.lstm_b, HallB + (size_t)(ts + 1) * s.state * s.e, t.Call + (size_t)(ts + 1) * s.state,
                                 (char*)t.gates + (size_t)ts * s.state * 4 * s.e, dt, Ms, C, stream));
        if (hipMemcpyAsync(t.c_last, t.Call + (size_t)T * s.state, s.state * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) {
            set_last_error("stage_seq_train_fwd: state copy failed"); return 1;
        }
    }
    return check_launch("stage_seq_train_fwd");
}

size_t rvt_stage_seq_bwd_ws_bytes(const RvtStageDesc* d, const RvtStageTrain* t, int T, int B) {
    if (!desc_ok(d, t) || T < 1 || B < 1) return 0;
    return train_bwd_ws_bytes(*d, *t, T, B);
}

int rvt_stage_seq_bwd(const RvtStageDesc* dp, const RvtStageTrain* tp, const void* inp, const void* dH, const float* dc_last,
                      const void* prev_cot, void* d_in, void* dh0, float* dc0, void* ws, size_t ws_bytes, int T, int B, void* stream) {
    RVT_CHECK(desc_ok(dp, tp) && tp->tb != nullptr, "stage_seq_bwd: bad descriptors");
    const RvtStageDesc& d = *dp; const RvtStageTrain& t = *tp;
    RVT_CHECK(T >= 1 && B >= 1 && inp != nullptr && dh0 != nullptr && dc0 != nullptr, "stage_seq_bwd: bad arguments");
    RVT_CHECK(ws != nullptr && ws_bytes >= train_bwd_ws_bytes(d, t, T, B), "stage_seq_bwd: workspace of %zu bytes < rvt_stage_seq_bwd_ws_bytes = %zu",
              ws_bytes, train_bwd_ws_bytes(d, t, T, B));
    const int C = d.C, F = T * B, dt = d.dtype;
    const TrainSizes s = train_sizes(d, t, T, B);
    const int H = s.H, W = s.W, M = s.M, Ms = B * H * W;
    hipStream_t st = (hipStream_t)stream;
    TCarver cv{(char*)ws, ws_bytes};
    void* ring[3] = {cv.take(s.act), cv.take(s.act), cv.take(s.act)};
    void* big4 = cv.take(4 * s.act);
    void* dqkv = cv.take(3 * s.act);
    void* t1 = cv.take(s.act);
    void* dy0 = cv.take(s.act);
    void* dh_b = cv.take(s.state * s.e);
    float* ws_wgrad = (float*)cv.take(s.wgrad_floats * 4);
    float* ws_mlp = s.mlp_floats ? (float*)cv.take(s.mlp_floats * 4) : nullptr;
    float* ws_scan = s.scan_floats ? (float*)cv.take(s.scan_floats * 4) : nullptr;
    float* ws_stem = s.stem_floats ? (float*)cv.take(s.stem_floats * 4) : nullptr;
    RVT_CHECK(cv.ok, "stage_seq_bwd: workspace carving overflow");
    const int nb = 2 * d.num_blocks;
    const void* x_last = nb > 0 ? t.saved[nb - 1].xout : t.x0;       // the ConvLSTM's input frames
    const char* const HallB = (const char*)t.Hall;

    // ---- ConvLSTM BPTT ----
    void* dx = ring[0];
    void* dz = big4;
    bool lstm_wgrad_done = false;
    if (t.lstm_route == 3) {
        RVT_TRY(rvt_lstm_scan3_bwd(t.gates, t.Csave, t.c0_saved, dH, dc_last, t.lstm_wtp3, dx, dz, dh0, dc0, dt, Ms, C, T, stream));
    } else if (t.lstm_route == 1 || t.lstm_route == 2) {
        lstm_wgrad_done = t.lstm_route == 1 && t.lstm_scan_wgrad != 0;
        RVT_TRY(rvt_lstm_scan_bwd(x_last, t.Hall, t.Csave, t.c0_saved, dH, dc_last, d.lstm_wn, t.lstm_wt, d.lstm_bn, dx, lstm_wgrad_done ? nullptr : dz,
                                  dh0, dc0, lstm_wgrad_done ? t.d_lstm_w : nullptr, lstm_wgrad_done ? t.d_lstm_b : nullptr, ws_scan,
                                  t.lstm_route == 2 ? t.gates : nullptr, dt, Ms, C, T, stream));
    } else {
        RVT_CHECK(dH != nullptr, "stage_seq_bwd: the per-step route needs dH (zeros, not NULL)");
        if (dc_last != nullptr) { if (hipMemcpyAsync(dc0, dc_last, s.state * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) { set_last_error("stage_seq_bwd: copy failed"); return 1; } }
        else if (hipMemsetAsync(dc0, 0, s.state * 4, st) != hipSuccess) { set_last_error("stage_seq_bwd: memset failed"); return 1; }
        void* dh_buf[2] = {dh0, dh_b};                   // step t writes dh_buf[t & 1]: the last one (t = 0) lands in dh0
        const void* dh_rec = nullptr;
        for (int ts = T - 1; ts >= 0; ts--) {
            char* const dz_t = (char*)dz + (size_t)ts * s.state * 4 * s.e;
            RVT_TRY(rvt_lstm_gates_bwd((const char*)dH + (size_t)ts * s.state * s.e, dh_rec, dc0, (const char*)t.gates + (size_t)ts * s.state * 4 * s.e,
                                       t.Call + (size_t)(ts + 1) * s.state, t.Call + (size_t)ts * s.state, dz_t, dt, Ms, C, stream));
            void* nxt = dh_buf[ts & 1];
            RVT_TRY(rvt_lstm_dgrad(dz_t, t.lstm_wt, (char*)dx + (size_t)ts * s.state * s.e, nxt, dt, Ms, C, stream));
            dh_rec = nxt;
        }
    }
    if (!lstm_wgrad_done)
        RVT_TRY(rvt_lstm_wgrad(dz, x_last, HallB, t.d_lstm_w, t.d_lstm_b, ws_wgrad, dt, M, C, stream));

    // ---- attention blocks, reversed ----
    bool preln_done = false;                                 // the first block's launch already wrote dy0 (rvt_attn_block_bwd_preln)
    int ri = 0;                                              // ring[ri] = the cotangent of the current block's output
    for (int bi = nb - 1; bi >= 0; bi--) {
        const RvtBlockWeights& bw = d.blocks[bi];
        const RvtBlockSaved& sv = t.saved[bi];
        const RvtBlockTrain& tb = t.tb[bi];
        const int window = (bi & 1) == 0;
        void* const dxo = ring[ri];
        void* const dxmid = ring[(ri + 1) % 3];
        void* const dxi = ring[(ri + 2) % 3];
        // MLP branch: xout = xmid + g2 * (gelu(h) W2^T + b2)
        if (t.mlp_route == 1) {
            if (t.mlp_bwd_both) {
                RVT_TRY(rvt_mlp_bwd_recompute_both(dxo, sv.xmid, dxmid, bw.n2_w, bw.n2_b, bw.fc1_w, bw.fc1_b, tb.fc2_wt, tb.fc1_wt, tb.d_n2_w, tb.d_n2_b,
                                                   tb.d_fc1_w, tb.d_fc1_b, tb.d_S2, tb.d_cs2, ws_mlp, dt, M, C, d.eps, stream));
            } else {
                RVT_TRY(rvt_mlp_bwd_recompute_wgrad(dxo, sv.xmid, bw.n2_w, bw.n2_b, bw.fc1_w, bw.fc1_b, tb.fc2_wt, tb.d_fc1_w, tb.d_fc1_b, tb.d_S2, tb.d_cs2,
                                                    ws_mlp, dt, M, C, d.eps, stream));
                RVT_TRY(rvt_mlp_bwd_recompute_dgrad(dxo, sv.xmid, dxmid, bw.n2_w, bw.n2_b, bw.fc1_w, bw.fc1_b, tb.fc2_wt, tb.fc1_wt, tb.d_n2_w, tb.d_n2_b,
                                                    dt, M, C, d.eps, stream));
            }
        } else {
            void* const dhd = big4;                          // (dz is consumed: rvt_lstm_wgrad is enqueued ahead on this stream)
            RVT_TRY(rvt_linear_wgrad(dxo, sv.hg, tb.d_S2, tb.d_cs2, ws_wgrad, dt, M, C, 4 * C, 0, stream));
            RVT_TRY(rvt_linear_dgrad(dxo, tb.fc2_wt, nullptr, nullptr, sv.hgp, dhd, dt, M, C, 4 * C, stream));
            RVT_TRY(rvt_linear_wgrad(dhd, sv.v2, tb.d_fc1_w, tb.d_fc1_b, ws_wgrad, dt, M, 4 * C, C, 0, stream));
            if (t.dgrad_ln_fc1) {
                RVT_TRY(rvt_linear_dgrad_ln(dhd, bw.fc1_w, sv.xmid, dxo, dxmid, bw.n2_w, tb.d_n2_w, tb.d_n2_b, dt, M, C, 4 * C, d.eps, stream));
            } else {
                RVT_TRY(rvt_linear_dgrad(dhd, tb.fc1_wt, nullptr, nullptr, nullptr, t1, dt, M, 4 * C, C, stream));
                RVT_TRY(rvt_layernorm_bwd(sv.xmid, bw.n2_w, t1, dxo, dxmid, tb.d_n2_w, tb.d_n2_b, dt, M, C, d.eps, stream));
            }
        }
        // attention branch: xmid = xin + g1 * (a Wp^T + bp)
        RVT_TRY(rvt_linear_wgrad(dxmid, sv.a, tb.d_S1, tb.d_cs1, ws_wgrad, dt, M, C, C, 0, stream));
        if (t.attn_block) {
            void* const u_out = bw.n1_w != nullptr ? t1 : nullptr;
            if (t.attn_preln && bi == 0 && bw.n1_w == nullptr) {
                // the stage's first block: the same launch carries the gradient through the down-sampling norm in front of it
                RVT_TRY(rvt_attn_block_bwd_preln(sv.xin, t.y0, dxmid, dy0, dqkv, d.ln_w, bw.qkv_w, bw.qkv_b, tb.proj_wt, t.d_ln_w, t.d_ln_b, dt, F, H, W,
                                                 C, d.dim_head, d.ph, d.pw, window, d.eps, stream));
                preln_done = true;
            } else {
                RVT_TRY(rvt_attn_block_bwd(sv.xin, dxmid, dxi, dqkv, u_out, bw.n1_w, bw.n1_b, bw.qkv_w, bw.qkv_b, tb.proj_wt, tb.d_n1_w, tb.d_n1_b, dt, F, H, W,
                                           C, d.dim_head, d.ph, d.pw, window, d.eps, stream));
            }
            RVT_TRY(rvt_linear_wgrad(dqkv, u_out != nullptr ? u_out : sv.xin, tb.d_qkv_w, tb.d_qkv_b, ws_wgrad, dt, M, 3 * C, C, 0, stream));
        } else {
            RVT_TRY(rvt_linear_dgrad(dxmid, tb.proj_wt, nullptr, nullptr, nullptr, t1, dt, M, C, C, stream));          // da
            RVT_TRY(rvt_attn_bwd(sv.qkv, t1, dqkv, dt, F, H, W, C, d.dim_head, d.ph, d.pw, window, stream));
            RVT_TRY(rvt_linear_wgrad(dqkv, bw.n1_w != nullptr ? sv.u : sv.xin, tb.d_qkv_w, tb.d_qkv_b, ws_wgrad, dt, M, 3 * C, C, 0, stream));
            if (bw.n1_w == nullptr && bi == 0 && t.attn_preln) {
                // the stage's first block: qkv input gradient + residual cotangent carried through the down-sampling norm (one launch)
                RVT_TRY(rvt_linear_dgrad_preln(dqkv, bw.qkv_w, t.y0, dxmid, dy0, d.ln_w, t.d_ln_w, t.d_ln_b, dt, M, C, 3 * C, d.eps, stream));
                preln_done = true;
            } else if (bw.n1_w == nullptr) {
                RVT_TRY(rvt_linear_dgrad(dqkv, tb.qkv_wt, nullptr, dxmid, nullptr, dxi, dt, M, 3 * C, C, stream));
            } else if (t.dgrad_ln_qkv) {
                RVT_TRY(rvt_linear_dgrad_ln(dqkv, bw.qkv_w, sv.xin, dxmid, dxi, bw.n1_w, tb.d_n1_w, tb.d_n1_b, dt, M, C, 3 * C, d.eps, stream));
            } else {
                RVT_TRY(rvt_linear_dgrad(dqkv, tb.qkv_wt, nullptr, nullptr, nullptr, t1, dt, M, 3 * C, C, stream));   // du
                RVT_TRY(rvt_layernorm_bwd(sv.xin, bw.n1_w, t1, dxmid, dxi, tb.d_n1_w, tb.d_n1_b, dt, M, C, d.eps, stream));
            }
        }
        ri = (ri + 2) % 3;
    }
    // ---- down-sampling LayerNorm + conv ----
    if (!preln_done) RVT_TRY(rvt_layernorm_bwd(t.y0, d.ln_w, ring[ri], nullptr, dy0, t.d_ln_w, t.d_ln_b, dt, M, C, d.eps, stream));
    if (d.inp_u8) {
        RVT_TRY(rvt_stem_wgrad(inp, dy0, t.d_raw_conv, ws_stem, dt, F, d.Cin, d.cin_pad, d.h_raw, d.w_raw, d.H_in, d.W_in, stream));
    } else {
        RVT_TRY(rvt_conv_wgrad(inp, dy0, t.d_raw_conv, ws_wgrad, dt, F, d.H_in, d.W_in, d.cin_pad, C, d.k, d.stride, d.pad, stream));
    }
    if (d_in != nullptr) {
        if (t.conv_dgrad4) {
            RVT_TRY(rvt_conv_dgrad4(dy0, t.conv_wd4, prev_cot, d_in, dt, F, d.H_in, d.W_in, d.cin_pad, C, stream));
        } else {
            RVT_TRY(rvt_conv_dgrad(dy0, t.conv_wd, prev_cot, d_in, dt, F, d.H_in, d.W_in, d.cin_pad, C, d.k, d.stride, d.pad, stream));
        }
    }
    return check_launch("stage_seq_bwd");
}

}  // extern "C"
#undef RVT_TRY
