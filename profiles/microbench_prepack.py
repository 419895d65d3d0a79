import sys, os, torch
sys.path.insert(0, '/root/repo')
from rvt_amd import _lib, ops
if len(sys.argv) > 1:
    _lib._install_test_library(_lib.load_library(os.path.abspath(sys.argv[1])))
src = torch.randint(0, 11, (504, 20, 360, 640), dtype=torch.uint8, device='cuda')
out = torch.empty(504, 384, 640, 24, dtype=torch.bfloat16, device='cuda')
for _ in range(3): ops.prepack_input(src, 384, 640, 24, torch.bfloat16, out=out)
torch.cuda.synchronize()
ts=[]
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.prepack_input(src, 384, 640, 24, torch.bfloat16, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[3]
print(sys.argv[1:] or 'default', f'{ms:.3f} ms', f'{(src.numel() + out.numel()*2)/ms/1e9:.2f} TB/s')
