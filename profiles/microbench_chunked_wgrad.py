"""Does the qkv weight gradient of stage 1 get cheaper when it reads dqkv right behind the kernel that wrote it, chunk by chunk
(the chunk staying in the 256-MB memory-side cache)?  attn_block_bwd + linear_wgrad over all 504 frames at once, against the same
two kernels over chunks of frames with ONE reused dqkv buffer.  usage: python profiles/microbench_chunked_wgrad.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from rvt_amd import ops, _lib as L
dev, dt = torch.device('cuda', 0), torch.bfloat16
F_, H, W, C, dh, ph, pw = 504, 96, 160, 64, 32, 6, 10
rnd = lambda *s: torch.randn(*s, device=dev).to(dt)
x, dxmid = rnd(F_, H, W, C), rnd(F_, H, W, C)
wqkv, bqkv, wpt = rnd(3 * C, C) * 0.1, torch.zeros(3 * C, device=dev), rnd(C, C) * 0.1
lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
dlw, dlb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
dW, db = torch.zeros(3 * C, C, device=dev), torch.zeros(3 * C, device=dev)
dx = torch.empty_like(x)
def call_bwd(f0, f1, dqkv, u):
    n = f1 - f0
    L.call('rvt_attn_block_bwd', L.ptr(x[f0:f1]), L.ptr(dxmid[f0:f1]), L.ptr(dx[f0:f1]), L.ptr(dqkv), L.ptr(u), L.ptr(lnw), L.ptr(lnb), L.ptr(wqkv),
           L.ptr(bqkv), L.ptr(wpt), L.ptr(dlw), L.ptr(dlb), L.dtype_code(dt), n, H, W, C, dh, ph, pw, 0, 1e-5, L.stream_of(x))
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
dq_full, u_full = torch.empty(F_, H, W, 3 * C, device=dev, dtype=dt), torch.empty(F_, H, W, C, device=dev, dtype=dt)
def whole():
    call_bwd(0, F_, dq_full, u_full)
    ops.linear_wgrad(dq_full, u_full, dW, colsum_out=db)
print(f'whole: attn_block_bwd + qkv wgrad = {timeit(whole):.3f} ms', flush=True)
t_b = timeit(lambda: call_bwd(0, F_, dq_full, u_full)); print(f'   attn_block_bwd alone {t_b:.3f} ms', flush=True)
for nch in (8, 16, 32, 63):
    fc = (F_ + nch - 1) // nch
    dq_c, u_c = torch.empty(fc, H, W, 3 * C, device=dev, dtype=dt), torch.empty(fc, H, W, C, device=dev, dtype=dt)
    def chunked():
        for f0 in range(0, F_, fc):
            f1 = min(F_, f0 + fc)
            call_bwd(f0, f1, dq_c, u_c)
            ops.linear_wgrad(dq_c[:f1 - f0], u_c[:f1 - f0], dW, colsum_out=db)
    def chunked_bwd_only():
        for f0 in range(0, F_, fc):
            call_bwd(f0, min(F_, f0 + fc), dq_c, u_c)
    print(f'{nch:3d} chunks of {fc} frames ({fc * H * W * 3 * C * 2 / 1e6:.0f} MB of dqkv): both {timeit(chunked):.3f} ms, attn_block_bwd alone {timeit(chunked_bwd_only):.3f} ms', flush=True)
