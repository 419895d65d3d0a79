"""One GEMM-family op in isolation, for `rocprofv3 --pmc` passes (see profiles/pmc_probe.sh).  usage: pmc_probe.py <op>"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops
if '--lib' in sys.argv:
    from rvt_amd import _lib
    _lib._install_test_library(_lib.load_library(os.path.abspath(sys.argv[sys.argv.index('--lib') + 1])))

dev, dt = torch.device('cuda', 0), torch.bfloat16
op = sys.argv[1] if len(sys.argv) > 1 else 'wgrad'
M, C = 7741440, 64
rnd = lambda *s: torch.randn(*s, device=dev).to(dt)
x, dy4 = rnd(M, C), rnd(M, 4 * C)
if op == 'wgrad':           # dW[256][64] = dy^T x  (+ bias gradient)
    dw, cs = torch.zeros(4 * C, C, device=dev), torch.zeros(4 * C, device=dev)
    fn = lambda: ops.linear_wgrad(dy4, x, dw, colsum_out=cs)
elif op == 'fwd':           # y[M][256] = x W^T + b
    w, b, y = rnd(4 * C, C), torch.zeros(4 * C, device=dev), torch.empty(M, 4 * C, device=dev, dtype=dt)
    fn = lambda: ops.linear_fwd(x, w, b, out=y)
elif op == 'dgrad':         # dx[M][64] = dy W
    wt, dx = rnd(C, 4 * C), torch.empty(M, C, device=dev, dtype=dt)
    fn = lambda: ops.linear_dgrad(dy4, wt, out=dx)
elif op in ('attn_fwd', 'attn_bwd'):    # stage-1 window attention core: 504 frames of 96x160 tokens, C=64, 2 heads of 32
    F_, H, W = 504, 96, 160
    qkv = rnd(F_ * H * W, 3 * C)
    if op == 'attn_fwd':
        fn = lambda: ops.attn_fwd(qkv, F_, H, W, C, 32, 6, 10, True)
    else:
        do = rnd(F_ * H * W, C)
        fn = lambda: ops.attn_bwd(qkv, do, F_, H, W, C, 32, 6, 10, True)
elif op == 'dgrad_k1024':   # stage-3 fc1 input gradient: dx[M][256] = dh[M][1024] W   (MFMA-heavier K loop)
    M3 = 483840
    dh, wt, dx = rnd(M3, 1024), rnd(256, 1024), torch.empty(M3, 256, device=dev, dtype=dt)
    fn = lambda: ops.linear_dgrad(dh, wt, out=dx)
elif op == 'wgrad_s3':       # stage-3 fc1 weight gradient: dW[1024][256] = dh[M][1024]^T v[M][256]
    M3 = 483840
    dh3, v3 = rnd(M3, 1024), rnd(M3, 256)
    dw3, cs3 = torch.zeros(1024, 256, device=dev), torch.zeros(1024, device=dev)
    fn = lambda: ops.linear_wgrad(dh3, v3, dw3, colsum_out=cs3)
elif op == 'lstm_fwd_s3':    # one ConvLSTM step of stage 3: 23040 pixels, C = 256
    Mp, Cc = 23040, 256
    xs, hs, cs_ = rnd(Mp, Cc), rnd(Mp, Cc), torch.randn(Mp, Cc, device=dev)
    wl, bl = rnd(4 * Cc, 2 * Cc) * 0.05, torch.zeros(4 * Cc, device=dev)
    ho, co, go = torch.empty(Mp, Cc, device=dev, dtype=dt), torch.empty(Mp, Cc, device=dev), torch.empty(Mp, 4 * Cc, device=dev, dtype=dt)
    fn = lambda: ops.lstm_fwd(xs, hs, cs_, wl, bl, ho, co, go)
elif op == 'fwd_k512':      # stage-4 fc1: y[M][2048] = x[M][512] W^T
    M4 = 120960
    x4, w4, b4 = rnd(M4, 512), rnd(2048, 512), torch.zeros(2048, device=dev)
    y4 = torch.empty(M4, 2048, device=dev, dtype=dt)
    fn = lambda: ops.linear_fwd(x4, w4, b4, out=y4)
elif op in ('mlp_fwd', 'mlp_bwd'):
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, w2 = rnd(4 * C, C) * 0.1, rnd(C, 4 * C) * 0.1
    b1, b2, gam = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    if op == 'mlp_fwd':
        fn = lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=True)
    else:
        dy, dlw, dlb = rnd(M, C), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        w2gt, w1t = w2.t().contiguous(), w1.t().contiguous()
        fn = lambda: ops.mlp_bwd_dgrad(dy, dy4, x, lw, w2gt, w1t, dlw, dlb, 1e-5)
elif op in ('mlpc_fwd', 'mlpc_dgrad', 'mlpc_wgrad', 'mlpc_both'):      # recompute MLP route (csrc/mlp_chain.hpp unless RVT_MLP_CHAIN=0)
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    w1, w2 = rnd(4 * C, C) * 0.1, rnd(C, 4 * C) * 0.1
    b1, b2, gam = torch.zeros(4 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    dy = rnd(M, C)
    w2gt, w1t = w2.t().contiguous(), w1.t().contiguous()
    z = lambda *s: torch.zeros(*s, device=dev)
    dlw, dlb, dw1, db1, s2, cs2 = z(C), z(C), z(4 * C, C), z(4 * C), z(C, 4 * C), z(C)
    del dy4
    if op == 'mlpc_fwd':
        fn = lambda: ops.mlp_fwd(x, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=False)
    elif op == 'mlpc_dgrad':
        fn = lambda: ops.mlp_bwd_recompute_dgrad(dy, x, lw, lb, w1, b1, w2gt, w1t, dlw, dlb, 1e-5)
    elif op == 'mlpc_both':
        fn = lambda: ops.mlp_bwd_recompute_both(dy, x, lw, lb, w1, b1, w2gt, w1t, dlw, dlb, dw1, db1, s2, cs2, 1e-5)
    else:
        fn = lambda: ops.mlp_bwd_recompute_wgrad(dy, x, lw, lb, w1, b1, w2gt, dw1, db1, s2, cs2, 1e-5)
elif op in ('ab_fwd', 'ab_bwd', 'ab_fwd_ln', 'ab_bwd_ln'):      # fused attention half (csrc/attn_block.hpp), stage-1 window block
    F_, H, W = 504, 96, 160
    del dy4
    ln = op.endswith('_ln')
    lw, lb = (torch.ones(C, device=dev), torch.zeros(C, device=dev)) if ln else (None, None)
    wqkv, wp = rnd(3 * C, C) * 0.125, rnd(C, C) * 0.125
    bqkv, bp, gam = torch.zeros(3 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
    x4 = x.view(F_, H, W, C)
    if op.startswith('ab_fwd'):
        fn = lambda: ops.attn_block_fwd(x4, lw, lb, wqkv, bqkv, wp, bp, gam, F_, H, W, C, 32, 6, 10, True, 1e-5, True)
    else:
        dxm = rnd(F_, H, W, C)
        dlw, dlb = (torch.zeros(C, device=dev), torch.zeros(C, device=dev)) if ln else (None, None)
        fn = lambda: ops.attn_block_bwd(x4, dxm, lw, lb, wqkv, bqkv, wp.t().contiguous(), dlw, dlb, F_, H, W, C, 32, 6, 10, True, 1e-5)
elif op in ('conv_fwd_stem', 'conv_wgrad_stem'):      # stem: 7x7 stride-4 conv, 24 (20 padded) -> 64 channels, 504 frames of 384x640
    del x, dy4
    F_, Hi, Wi, Cp, Co, k, st, pd = 504, 384, 640, 24, 64, 7, 4, 3
    inp = rnd(F_, Hi, Wi, Cp)
    w = rnd(Co, k * k * Cp) * 0.05
    if op == 'conv_fwd_stem':
        fn = lambda: ops.conv_fwd(inp, w, k, st, pd)
    else:
        dy = rnd(F_, Hi // st, Wi // st, Co)
        dw = torch.zeros(Co, k * k * Cp, device=dev)
        fn = lambda: ops.conv_wgrad(inp, dy, dw, k, st, pd)
elif op in ('stem_fwd', 'stem_wgrad'):                # the same stem on the uint8 planes (csrc/stem.hpp)
    del x, dy4
    from rvt_amd import weights
    F_, Cin, h, w_, H, W = 504, 20, 360, 640, 384, 640
    src = torch.randint(0, 11, (F_, Cin, h, w_), dtype=torch.uint8, device=dev)
    wp = weights.pack_conv_fwd(torch.randn(64, Cin, 7, 7, device=dev) * 0.05, 24, dt)
    lw, lb = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    if op == 'stem_fwd':
        fn = lambda: ops.stem_fwd(src, wp, lw, lb, H, W, 1e-5)
    else:
        dy = rnd(F_, 96, 160, 64)
        dw = torch.zeros(64, 49 * 24, device=dev)
        fn = lambda: ops.stem_wgrad(src, dy, dw, H, W)
elif op in ('lstm_scan_fwd', 'lstm_scan_bwd'):         # stage-1 ConvLSTM scans: 368640 pixels x 21 steps, C = 64
    del x, dy4
    T_, Mp, Cc = 21, 368640, 64
    xa = rnd(T_, Mp, Cc)
    Hall = rnd(T_ + 1, Mp, Cc) * 0.5
    Cs = rnd(T_, Mp, Cc)
    cl = torch.empty(Mp, Cc, device=dev)
    w = rnd(4 * Cc, 2 * Cc) * 0.1
    b = torch.zeros(4 * Cc, device=dev)
    if op == 'lstm_scan_fwd':
        fn = lambda: ops.lstm_scan_fwd(xa, Hall, None, cl, Cs, w, b)
    else:
        dH, dxa = rnd(T_, Mp, Cc), torch.empty(T_, Mp, Cc, device=dev, dtype=dt)
        dh0, dc0 = torch.empty(Mp, Cc, device=dev, dtype=dt), torch.empty(Mp, Cc, device=dev)
        dw, db = torch.zeros(4 * Cc, 2 * Cc, device=dev), torch.zeros(4 * Cc, device=dev)
        wt = w.t().contiguous()
        fn = lambda: ops.lstm_scan_bwd(xa, Hall, Cs, None, dH, None, w, wt, b, dxa, None, dh0, dc0, dw=dw, db=db)
elif op in ('pp_fwd_s4', 'pp_dgrad_s4', 'pp_scale_res_s4', 'pp_wgrad_s4', 'pp_fwd_s3', 'pp_wgrad_s3'):
    # round 3: the 256 x 256 LDS-DMA GEMMs (csrc/ppgemm.hpp, ppgemm_tn.hpp) at the stage-4 / stage-3 MLP shapes
    Ms, Cs_ = (120960, 512) if op.endswith('s4') else (483840, 256)
    xs, hs = rnd(Ms, Cs_), rnd(Ms, 4 * Cs_)
    if op.startswith('pp_fwd'):
        w, b, y = rnd(4 * Cs_, Cs_) * 0.05, torch.zeros(4 * Cs_, device=dev), torch.empty(Ms, 4 * Cs_, device=dev, dtype=dt)
        fn = lambda: ops.linear_fwd(xs, w, b, out=y)
    elif op.startswith('pp_dgrad'):
        wt, dx = rnd(Cs_, 4 * Cs_) * 0.05, torch.empty(Ms, Cs_, device=dev, dtype=dt)
        fn = lambda: ops.linear_dgrad(hs, wt, out=dx)
    elif op.startswith('pp_scale_res'):
        w, b, gam = rnd(Cs_, 4 * Cs_) * 0.05, torch.zeros(Cs_, device=dev), torch.ones(Cs_, device=dev)
        y = torch.empty(Ms, Cs_, device=dev, dtype=dt)
        fn = lambda: ops.linear_scale_res_fwd(hs, w, b, gam, xs, out=y)
    else:
        dw, cs = torch.zeros(4 * Cs_, Cs_, device=dev), torch.zeros(4 * Cs_, device=dev)
        fn = lambda: ops.linear_wgrad(hs, xs, dw, colsum_out=cs)
elif op == 'conv_dgrad4_s3':   # stage-3 conv input gradient as one gather GEMM: 504 frames, 48 x 80 x 128 <- 24 x 40 x 256
    from rvt_amd import weights as Wt
    F_, H, W, Cin, Co = 504, 48, 80, 128, 256
    dyc = rnd(F_, H // 2, W // 2, Co)
    wd4 = Wt.pack_conv_dgrad4(torch.randn(Co, Cin, 3, 3, device=dev) * 0.05, dt)
    fn = lambda: ops.conv_dgrad4(dyc, wd4, None, H, W, Cin)
elif op in ('dgrad_ln_k512', 'dgrad_ln_k384'):    # stage-2 fc1 / qkv input gradient with the LayerNorm backward in the epilogue (csrc/dgrad_ln.hpp)
    Ms, Cs_, Ks = 1935360, 128, 512 if op.endswith('512') else 384
    xs, dys, dres = rnd(Ms, Cs_), rnd(Ms, Ks), rnd(Ms, Cs_)
    w, lw = rnd(Ks, Cs_) * 0.1, torch.rand(Cs_, device=dev) + 0.5
    dw, db, out = torch.zeros(Cs_, device=dev), torch.zeros(Cs_, device=dev), torch.empty(Ms, Cs_, device=dev, dtype=dt)
    fn = lambda: ops.linear_dgrad_ln(dys, w, xs, dres, lw, dw, db, 1e-5, out=out)
elif op in ('scan_fwd_s1', 'scan_bwd_s1', 'scan_fwd_s1_v1', 'scan_bwd_s1_v1'):     # stage-1 ConvLSTM scans, 368640 pixels x 21 steps, C = 64 (lstm_scan2.hpp / lstm_scan.hpp)
    from rvt_amd import tuning
    tuning.use(lstm_scan_v2=0 if op.endswith('_v1') else 1)
    T_, Mp, Cc = 21, 368640, 64
    xa, Hall, Cs = rnd(T_, Mp, Cc), rnd(T_ + 1, Mp, Cc) * 0.5, rnd(T_, Mp, Cc)
    wl, bl = rnd(4 * Cc, 2 * Cc) * 0.1, torch.zeros(4 * Cc, device=dev)
    if op.startswith('scan_fwd'):
        c_last = torch.empty(Mp, Cc, device=dev)
        fn = lambda: ops.lstm_scan_fwd(xa, Hall, None, c_last, Cs, wl, bl)
    else:
        dH, dxa = rnd(T_, Mp, Cc), torch.empty(T_, Mp, Cc, device=dev, dtype=dt)
        dh0, dc0 = torch.empty(Mp, Cc, device=dev, dtype=dt), torch.empty(Mp, Cc, device=dev)
        wt = wl.t().contiguous()
        dw, db = torch.zeros(4 * Cc, 2 * Cc, device=dev), torch.zeros(4 * Cc, device=dev)
        fn = lambda: ops.lstm_scan_bwd(xa, Hall, Cs, None, dH, None, wl, wt, bl, dxa, None, dh0, dc0, dw=dw, db=db)
elif op in ('mlps_fwd', 'mlps_dgrad', 'mlps_wgrad'):       # round 5: streamed-weight MLP kernels at the stage-2 shape (csrc/mlp_stream.hpp)
    del x, dy4
    Ms, Cs_ = 1935360, 128
    xs, dys = rnd(Ms, Cs_), rnd(Ms, Cs_)
    lw, lb = torch.ones(Cs_, device=dev), torch.zeros(Cs_, device=dev)
    w1, w2 = rnd(4 * Cs_, Cs_) * 0.1, rnd(Cs_, 4 * Cs_) * 0.1
    b1, b2, gam = torch.zeros(4 * Cs_, device=dev), torch.zeros(Cs_, device=dev), torch.ones(Cs_, device=dev)
    w2gt, w1t = w2.t().contiguous(), w1.t().contiguous()
    z = lambda *s: torch.zeros(*s, device=dev)
    dlw, dlb, dw1, db1, s2, cs2 = z(Cs_), z(Cs_), z(4 * Cs_, Cs_), z(4 * Cs_), z(Cs_, 4 * Cs_), z(Cs_)
    if op == 'mlps_fwd':
        fn = lambda: ops.mlp_fwd(xs, lw, lb, w1, b1, w2, b2, gam, 1e-5, want_grad=False)
    elif op == 'mlps_dgrad':
        fn = lambda: ops.mlp_bwd_recompute_dgrad(dys, xs, lw, lb, w1, b1, w2gt, w1t, dlw, dlb, 1e-5)
    else:
        fn = lambda: ops.mlp_bwd_recompute_wgrad(dys, xs, lw, lb, w1, b1, w2gt, dw1, db1, s2, cs2, 1e-5)
if op.startswith('scan3_'):       # round 6: wide / long ConvLSTM scans with streamed weights (csrc/lstm_scan3.hpp): scan3_{fwd,bwd}_{256,128}
    Cc = int(op.split('_')[2]); Mp, T_ = (23040, 21) if Cc == 256 else (92160, 21)
    xa = rnd(T_, Mp, Cc)
    w = rnd(4 * Cc, 2 * Cc) * 0.05
    b = torch.zeros(4 * Cc, device=dev)
    wp, wtp = ops.lstm_scan3_pack(w)
    rows = ops.lstm_scan3_rows(Cc, Mp)
    Hall = torch.zeros(T_ + 1, Mp, Cc, device=dev, dtype=dt)
    Cs, gs = torch.empty(T_, rows, Cc, device=dev, dtype=dt), torch.empty(T_, rows, 4 * Cc, device=dev, dtype=dt)
    cl = torch.empty(Mp, Cc, device=dev)
    ops.lstm_scan3_fwd(xa, Hall, None, cl, Cs, wp, b, gs)
    if op.startswith('scan3_fwd'):
        fn = lambda: ops.lstm_scan3_fwd(xa, Hall, None, cl, Cs, wp, b, gs)
    else:
        dH, dcl = rnd(T_, Mp, Cc), torch.randn(Mp, Cc, device=dev)
        dxa, dz = torch.empty(T_, Mp, Cc, device=dev, dtype=dt), torch.empty(T_, Mp, 4 * Cc, device=dev, dtype=dt)
        dh0, dc0 = torch.empty(Mp, Cc, device=dev, dtype=dt), torch.empty(Mp, Cc, device=dev)
        fn = lambda: ops.lstm_scan3_bwd(gs, Cs, None, dH, dcl, wtp, dxa, dz, dh0, dc0)
for _ in range(3):
    fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
print(f'[{op}] one launch: {e0.elapsed_time(e1):.3f} ms')
