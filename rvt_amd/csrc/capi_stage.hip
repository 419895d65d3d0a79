// extern "C" entry points, part 9: the stage-major driver of ONE backbone stage over a whole sequence (SURVEY.md §8b:
// rvt_stage_seq_fwd).  Round 4: the no-grad forward — validation and streaming inference (reference
// modules/detection.py:231-255 calling maxvit_rnn.py:93-105,169-182 once per time step).  The host loop of rvt_amd/stage.py
// (kernel routing, workspace carving, the per-step ConvLSTM launches) runs here in C++: one C call per stage instead of
// 20-60 Python-level operator calls (the streaming step of RVT-Base, B = 64, was 4.1 ms of host enqueue for 0.9 ms of kernels).
// Nothing is launched that the operator entry points do not launch; this file only sequences them.
#include "host.hpp"

using namespace rvt;

namespace {
struct Carver {                       // bump allocator over the caller's workspace (256-byte aligned pieces)
    char* p; size_t left; bool ok = true;
    void* take(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > left) { ok = false; return nullptr; }
        void* r = p; p += bytes; left -= bytes;
        return r;
    }
};
static inline size_t elt_bytes(int dtype) { return dtype == RVT_F32 ? 4 : 2; }
static inline int conv_out(int x, int k, int s, int p) { return (x + 2 * p - k) / s + 1; }

// the routes of rvt_amd/stage.py (use_attn_block / use_fused_mlp / use_lstm_scan) for a forward that keeps nothing
static bool route_attn_block(const RvtStageDesc& d) {
    return tuning().route_attn_block != 0 && rvt_attn_block_supported(d.dtype, d.C, d.dim_head, d.ph * d.pw);
}
static bool route_fused_mlp_infer(const RvtStageDesc& d) {
    const int mode = tuning().route_fused_mlp;
    // the widths whose backward recomputes from the block input run the same nothing-saved forward when training
    if (mode != 0 && tuning().route_mlp_bwd_fused != 0 && rvt_mlp_bwd_fused_supported(d.dtype, d.C)) return true;
    if (mode == 0 || !rvt_mlp_fused_supported(d.dtype, d.C)) return false;
    return mode == 1 || d.C == 64 || d.C == 128;
}
static bool route_lstm_scan(const RvtStageDesc& d, int T) {
    const int mode = tuning().route_lstm_scan;
    if (mode == 0 || !rvt_lstm_scan_supported(d.dtype, d.C)) return false;
    if (mode == -1 && T == 1) return false;      // one step (streaming inference): the per-step GEMM beats staging the scan's weights (3.69 vs 3.81 ms per step at B = 64)
    return mode == 1 || d.C <= 64 || rvt_lstm_scan_saves_gates(d.dtype, d.C);
}
// (rvt_amd/stage.py: use_lstm_scan3) wide stages / long stage-2 scans with streamed weights; one step keeps the per-step GEMM
static bool route_lstm_scan3(const RvtStageDesc& d, int T, int Ms) {
    if (T <= 1 || !rvt_lstm_scan3_supported(d.dtype, d.C)) return false;
    return !(d.C == 128 && Ms < 16384 && tuning().route_lstm_scan != 0);
}
static size_t stage_ws_bytes(const RvtStageDesc& d, int T, int B) {
    const int H = conv_out(d.H_in, d.k, d.stride, d.pad), W = conv_out(d.W_in, d.k, d.stride, d.pad);
    const size_t tok = (size_t)T * B * H * W, e = elt_bytes(d.dtype), pad = 256;
    size_t n = 3 * (tok * d.C * e + pad);                                   // activation ping-pong (y0 / x / xmid / xout)
    if (!route_attn_block(d)) n += tok * d.C * e * 5 + 3 * pad;             // u, qkv (3C), a
    if (!route_fused_mlp_infer(d)) n += tok * d.C * e * 5 + 2 * pad;        // v2, GELU(h) (4C)
    if (!d.inp_u8 || !rvt_stem_supported(d.dtype, 1, d.Cin, d.C, d.k, d.stride, d.pad, d.w_raw))
        n += d.inp_u8 ? (size_t)T * B * d.H_in * d.W_in * d.cin_pad * e + pad : 0;      // prepacked input
    n += 2 * ((size_t)B * H * W * d.C * 4 + pad);                           // cell-state ping-pong of the per-step route
    n += (size_t)8 * d.C * d.C * e + pad;                                   // ConvLSTM weights in operand order (lstm_scan3 route)
    return n + 4096;
}
}  // namespace

extern "C" {

size_t rvt_stage_seq_fwd_ws_bytes(const RvtStageDesc* d, int T, int B) {
    if (d == nullptr || d->struct_bytes != (int)sizeof(RvtStageDesc) || T < 1 || B < 1) return 0;
    return stage_ws_bytes(*d, T, B);
}

int rvt_stage_seq_fwd(const RvtStageDesc* dp, const void* inp, const void* h0, const float* c0, void* Hall, float* c_last,
                      void* ws, size_t ws_bytes, int T, int B, void* stream) {
    RVT_CHECK(dp != nullptr && dp->struct_bytes == (int)sizeof(RvtStageDesc), "stage_seq_fwd: struct_bytes must be sizeof(RvtStageDesc) = %d",
              (int)sizeof(RvtStageDesc));
    const RvtStageDesc& d = *dp;
    RVT_CHECK(T >= 1 && B >= 1 && d.num_blocks >= 0 && d.blocks != nullptr && inp != nullptr && Hall != nullptr && c_last != nullptr,
              "stage_seq_fwd: bad arguments");
    RVT_CHECK((h0 == nullptr) == (c0 == nullptr), "stage_seq_fwd: h0 and c0 go together");
    RVT_CHECK(ws != nullptr && ws_bytes >= stage_ws_bytes(d, T, B), "stage_seq_fwd: workspace of %zu bytes < rvt_stage_seq_fwd_ws_bytes = %zu",
              ws_bytes, stage_ws_bytes(d, T, B));
    const int C = d.C, F = T * B, dt = d.dtype;
    const int H = conv_out(d.H_in, d.k, d.stride, d.pad), W = conv_out(d.W_in, d.k, d.stride, d.pad);
    RVT_CHECK(H % d.ph == 0 && W % d.pw == 0, "stage_seq_fwd: %dx%d not divisible by the partition %dx%d", H, W, d.ph, d.pw);
    const size_t e = elt_bytes(dt), tok = (size_t)F * H * W, act = tok * C * e;
    RVT_CHECK(tok * 4 * C < ((size_t)1 << 31), "stage_seq_fwd: %zu token rows exceed the operators' 32-bit sizes", tok);
    const int M = (int)tok;
    Carver cv{(char*)ws, ws_bytes};
    void* bufs[3] = {cv.take(act), cv.take(act), cv.take(act)};
    hipStream_t st = (hipStream_t)stream;
#define RVT_TRY(call) do { if ((call) != 0) return 1; } while (0)

    // ---- down-sampling conv + LayerNorm (maxvit.py:174-178) ----
    void* x = bufs[1];
    if (d.inp_u8 && rvt_stem_supported(dt, 1, d.Cin, C, d.k, d.stride, d.pad, d.w_raw)) {
        RVT_TRY(rvt_stem_fwd(inp, d.conv_w, d.ln_w, d.ln_b, bufs[0], x, dt, F, d.Cin, d.cin_pad, d.h_raw, d.w_raw, d.H_in, d.W_in, d.eps, stream));
    } else {
        const void* cin = inp;
        if (d.inp_u8) {                                   // loader planes, no stem kernel for this shape: cast + pad + repack first
            void* pk = cv.take((size_t)F * d.H_in * d.W_in * d.cin_pad * e);
            RVT_TRY(rvt_prepack_input(inp, 1, pk, dt, F, d.Cin, d.h_raw, d.w_raw, d.H_in, d.W_in, d.cin_pad, stream));
            cin = pk;
        }
        RVT_TRY(rvt_conv_fwd(cin, d.conv_w, bufs[0], dt, F, d.H_in, d.W_in, d.cin_pad, C, d.k, d.stride, d.pad, stream));
        RVT_TRY(rvt_layernorm_fwd(bufs[0], d.ln_w, d.ln_b, x, dt, M, C, d.eps, stream));
    }
    int xi = 1;                                           // bufs[xi] = the running activation
    auto other = [&](int a, int b) { return 3 - a - b; };

    // ---- attention blocks (maxvit.py:267-270): window, then grid ----
    const bool fused_attn = route_attn_block(d), fused_mlp = route_fused_mlp_infer(d);
    void *u = nullptr, *qkv = nullptr, *a_ = nullptr, *v2 = nullptr, *hg = nullptr;
    if (!fused_attn) { u = cv.take(act); qkv = cv.take(3 * act); a_ = cv.take(act); }
    if (!fused_mlp) { v2 = cv.take(act); hg = cv.take(4 * act); }
    RVT_CHECK(cv.ok, "stage_seq_fwd: workspace carving overflow");
    for (int bi = 0; bi < 2 * d.num_blocks; bi++) {
        const RvtBlockWeights& bw = d.blocks[bi];
        const int window = (bi & 1) == 0;
        const int mi = (xi + 1) % 3, oi = other(xi, mi);
        void* xin = bufs[xi]; void* xmid = bufs[mi]; void* xout = bufs[oi];
        if (fused_attn) {
            RVT_TRY(rvt_attn_block_fwd(xin, xmid, nullptr, bw.n1_w, bw.n1_b, bw.qkv_w, bw.qkv_b, bw.proj_w, bw.proj_b, bw.g1, dt, F, H, W, C,
                                       d.dim_head, d.ph, d.pw, window, d.eps, stream));
        } else {
            const void* uu = xin;
            if (rvt_ln_linear_supported(dt, C, 3 * C)) {
                RVT_TRY(rvt_ln_linear_fwd(xin, bw.n1_w, bw.n1_b, bw.qkv_w, bw.qkv_b, nullptr, qkv, dt, M, C, 3 * C, d.eps, stream));
            } else {
                if (bw.n1_w != nullptr) { RVT_TRY(rvt_layernorm_fwd(xin, bw.n1_w, bw.n1_b, u, dt, M, C, d.eps, stream)); uu = u; }
                RVT_TRY(rvt_linear_fwd(uu, bw.qkv_w, bw.qkv_b, qkv, dt, M, 3 * C, C, 0, stream));
            }
            RVT_TRY(rvt_attn_fwd(qkv, a_, dt, F, H, W, C, d.dim_head, d.ph, d.pw, window, stream));
            RVT_TRY(rvt_linear_scale_res_fwd(a_, bw.proj_w, bw.proj_b, bw.g1, xin, xmid, dt, M, C, C, 0, stream));
        }
        if (fused_mlp) {
            RVT_TRY(rvt_mlp_fwd(xmid, xout, nullptr, nullptr, nullptr, bw.n2_w, bw.n2_b, bw.fc1_w, bw.fc1_b, bw.fc2_w, bw.fc2_b, bw.g2, dt, M, C,
                                d.eps, stream));
        } else {
            RVT_TRY(rvt_layernorm_fwd(xmid, bw.n2_w, bw.n2_b, v2, dt, M, C, d.eps, stream));
            RVT_TRY(rvt_linear_gelu_fwd(v2, bw.fc1_w, bw.fc1_b, hg, nullptr, dt, M, 4 * C, C, stream));
            RVT_TRY(rvt_linear_scale_res_fwd(hg, bw.fc2_w, bw.fc2_b, bw.g2, xmid, xout, dt, M, C, 4 * C, 0, stream));
        }
        xi = oi;
    }
    x = bufs[xi];

    // ---- ConvLSTM over the T steps (rnn.py:43-67); Hall slot 0 = incoming h, slots 1..T = the stage's output features ----
    const size_t sN = (size_t)B * H * W * C;             // elements of one state
    const int Ms = B * H * W;
    char* const HallB = (char*)Hall;
    if (route_lstm_scan3(d, T, Ms)) {
        void* wp = cv.take((size_t)8 * C * C * e);
        RVT_CHECK(cv.ok, "stage_seq_fwd: workspace carving overflow");
        if (h0 != nullptr) { if (hipMemcpyAsync(HallB, h0, sN * e, hipMemcpyDeviceToDevice, st) != hipSuccess) { set_last_error("stage_seq_fwd: state copy failed"); return 1; } }
        else if (hipMemsetAsync(HallB, 0, sN * e, st) != hipSuccess) { set_last_error("stage_seq_fwd: memset failed"); return 1; }
        RVT_TRY(rvt_lstm_scan3_pack(d.lstm_wn, wp, nullptr, C, stream));
        RVT_TRY(rvt_lstm_scan3_fwd(x, Hall, c0, c_last, nullptr, wp, d.lstm_bn, nullptr, dt, Ms, C, T, stream));
    } else if (route_lstm_scan(d, T)) {
        if (h0 != nullptr) { if (hipMemcpyAsync(HallB, h0, sN * e, hipMemcpyDeviceToDevice, st) != hipSuccess) { set_last_error("stage_seq_fwd: state copy failed"); return 1; } }
        else if (hipMemsetAsync(HallB, 0, sN * e, st) != hipSuccess) { set_last_error("stage_seq_fwd: memset failed"); return 1; }
        RVT_TRY(rvt_lstm_scan_fwd(x, Hall, c0, c_last, nullptr, d.lstm_wn, d.lstm_bn, nullptr, dt, Ms, C, T, stream));
    } else {
        float* cbuf[2] = {(float*)cv.take(sN * 4), (float*)cv.take(sN * 4)};
        RVT_CHECK(cv.ok, "stage_seq_fwd: workspace carving overflow");
        const void* h_prev = h0;
        const float* c_prev = c0;
        if (h0 == nullptr) {                              // None state -> zeros (rnn.py:43-47)
            if (hipMemsetAsync(HallB, 0, sN * e, st) != hipSuccess || hipMemsetAsync(cbuf[1], 0, sN * 4, st) != hipSuccess) {
                set_last_error("stage_seq_fwd: memset failed"); return 1;
            }
            h_prev = HallB; c_prev = cbuf[1];
        }
        for (int t = 0; t < T; t++) {
            float* c_out = t + 1 == T ? c_last : cbuf[t & 1];
            RVT_TRY(rvt_lstm_fwd((const char*)x + (size_t)t * sN * e, h_prev, c_prev, d.lstm_w, d.lstm_b, HallB + (size_t)(t + 1) * sN * e, c_out,
                                 nullptr, dt, Ms, C, stream));
            h_prev = HallB + (size_t)(t + 1) * sN * e;
            c_prev = c_out;
        }
    }
#undef RVT_TRY
    return check_launch("stage_seq_fwd");
}

}  // extern "C"
