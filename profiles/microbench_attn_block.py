"""GPU micro-benchmark: fused attention half (csrc/attn_block.hpp) vs the op-by-op chain at the RVT-Base 1Mpx stage-1 shape
(F = 504 frames of 96x160 tokens, C = 64, 6x10 partitions), window and grid, with and without norm1."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops

dev = torch.device('cuda', 0)
dt = torch.bfloat16
F_, H, W, C, dh, ph, pw, eps = 504, 96, 160, 64, 32, 6, 10, 1e-5
if '--small' in sys.argv:
    F_ = 48
if '--gen1' in sys.argv:      # RVT-Base on Gen1 (config/experiment/gen1/base.yaml): 64 x 80 tokens at stage 1, 8 x 10 = 80-token partitions
    F_, H, W, ph, pw = 504, 64, 80, 8, 10


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


g = torch.Generator(device=dev).manual_seed(0)
M = F_ * H * W
x = torch.randn(F_, H, W, C, device=dev, generator=g).to(dt)
dxm = torch.randn(F_, H, W, C, device=dev, generator=g).to(dt)
lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
wqkv = (torch.randn(3 * C, C, device=dev, generator=g) * 0.125).to(dt)
wp = (torch.randn(C, C, device=dev, generator=g) * 0.125).to(dt)
bqkv, bp, gam = torch.zeros(3 * C, device=dev), torch.zeros(C, device=dev), torch.ones(C, device=dev)
wpgt, wqkvt = wp.t().contiguous(), wqkv.t().contiguous()
dlw, dlb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
print(f'M = {M} tokens ({2 * M * C / 1e9:.2f} GB per row of C bf16)')
for window in (True, False):
    for ln in (False, True):
        a_w, a_b = (lw, lb) if ln else (None, None)
        t_f = timeit(lambda: ops.attn_block_fwd(x, a_w, a_b, wqkv, bqkv, wp, bp, gam, F_, H, W, C, dh, ph, pw, window, eps, True))
        t_fi = timeit(lambda: ops.attn_block_fwd(x, a_w, a_b, wqkv, bqkv, wp, bp, gam, F_, H, W, C, dh, ph, pw, window, eps, False))
        t_b = timeit(lambda: ops.attn_block_bwd(x, dxm, a_w, a_b, wqkv, bqkv, wpgt, dlw if ln else None, dlb if ln else None,
                                                F_, H, W, C, dh, ph, pw, window, eps))

        def chain_f():
            u = ops.layernorm_fwd(x, lw, lb, eps) if ln else x
            qkv = ops.linear_fwd(u, wqkv, bqkv)
            a = ops.attn_fwd(qkv, F_, H, W, C, dh, ph, pw, window)
            return ops.linear_scale_res_fwd(a, wp, bp, gam, x), qkv
        t_c = timeit(chain_f)
        _, qkv = chain_f()

        def chain_b():
            da = ops.linear_dgrad(dxm, wpgt)
            dqkv = ops.attn_bwd(qkv, da, F_, H, W, C, dh, ph, pw, window)
            if ln:
                du = ops.linear_dgrad(dqkv, wqkvt)
                return ops.layernorm_bwd(x, lw, du, dxm, dlw, dlb, eps)
            return ops.linear_dgrad(dqkv, wqkvt, add=dxm)
        t_cb = timeit(chain_b)
        del qkv
        print(f'window={int(window)} norm1={int(ln)}: fwd fused(train) {t_f:.3f} ms  fused(infer) {t_fi:.3f} ms  chain {t_c:.3f} ms | '
              f'bwd(dgrad) fused {t_b:.3f} ms  chain {t_cb:.3f} ms', flush=True)
