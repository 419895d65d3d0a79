"""GPU micro-benchmark: the stem of RVT-Base at the 1 Mpx bench shape (504 frames of 20 x 360 x 640 uint8, padded to 384 x 640):
prepack + im2col GEMM + LayerNorm (and the im2col^T weight-gradient GEMM) against the stem kernels on the uint8 planes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rvt_amd import ops, weights

dev, dt = torch.device('cuda', 0), torch.bfloat16


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


F_, Cin, h, w, H, W = int(os.environ.get('STEM_F', 504)), 20, 360, 640, 384, 640
g = torch.Generator().manual_seed(0)
src = torch.randint(0, 11, (F_, Cin, h, w), generator=g, dtype=torch.uint8).to(dev)
wt = (torch.randn(64, Cin, 7, 7, generator=g) * 0.05)
wp = weights.pack_conv_fwd(wt.to(dev), 24, dt)
lw, lb = torch.ones(64, device=dev), torch.zeros(64, device=dev)

t_pre = timeit(lambda: ops.prepack_input(src, H, W, 24, dt))
inp = ops.prepack_input(src, H, W, 24, dt)
t_conv = timeit(lambda: ops.conv_fwd(inp, wp, 7, 4, 3))
y_ref = ops.conv_fwd(inp, wp, 7, 4, 3)
t_ln = timeit(lambda: ops.layernorm_fwd(y_ref, lw, lb, 1e-5))
x_ref = ops.layernorm_fwd(y_ref, lw, lb, 1e-5)
t_stem = timeit(lambda: ops.stem_fwd(src, wp, lw, lb, H, W, 1e-5))
y0, x = ops.stem_fwd(src, wp, lw, lb, H, W, 1e-5)
print(f'forward : prepack {t_pre:.3f} + conv {t_conv:.3f} + LN {t_ln:.3f} = {t_pre + t_conv + t_ln:.3f} ms   |   stem_fwd {t_stem:.3f} ms', flush=True)
print(f'          max |y0 - conv_fwd| = {(y0.float() - y_ref.float()).abs().max().item():.4f} (max |y| {y_ref.float().abs().max().item():.2f}),'
      f' max |x - LN| = {(x.float() - x_ref.float()).abs().max().item():.4f}', flush=True)
flops = 2.0 * F_ * 96 * 160 * 64 * Cin * 49
print(f'          stem_fwd {flops / t_stem / 1e9:.0f} TFLOP/s of real taps; reads {src.numel() / 1e9:.2f} GB, writes {2 * y0.numel() * 2 / 1e9:.2f} GB'
      f' -> {(src.numel() + 4 * y0.numel()) / t_stem / 1e6:.0f} GB/s', flush=True)

dy = torch.randn(F_, 96, 160, 64, device=dev).to(dt)
dw_a, dw_b = torch.zeros(64, 49 * 24, device=dev), torch.zeros(64, 49 * 24, device=dev)
t_wg = timeit(lambda: ops.conv_wgrad(inp, dy, dw_a, 7, 4, 3))
t_swg = timeit(lambda: ops.stem_wgrad(src, dy, dw_b, H, W))
dw_a.zero_(); dw_b.zero_()
ops.conv_wgrad(inp, dy, dw_a, 7, 4, 3)
ops.stem_wgrad(src, dy, dw_b, H, W)
torch.cuda.synchronize()
print(f'wgrad   : conv_wgrad {t_wg:.3f} ms   |   stem_wgrad {t_swg:.3f} ms   ({flops / t_swg / 1e9:.0f} TFLOP/s);'
      f' max |diff| {(dw_a - dw_b).abs().max().item():.3f} of max {dw_a.abs().max().item():.1f}', flush=True)
