import sys, torch
sys.path.insert(0, '/root/repo')
from tests.test_backbone import run_hip_case
from tests.harness import compare, load_golden
for name in ['micro', 'tiny_gen1_gamma', 'base_qvga', 'base_1mpx']:
    got = run_hip_case(name, torch.device('cuda', 0), torch.bfloat16, with_batch2=False)
    for rt, grt in ((4e-2, 8e-2),):
        try:
            w = compare(got, load_golden(name), rtol=rt, what=name, grad_rtol=grt)
            print(name, 'worst err/tol at (4e-2, 8e-2):', round(w, 3))
        except AssertionError as e:
            print(name, 'FAIL', str(e)[:200])
    for rt, grt in ((2.5e-2, 4e-2),):
        try:
            w = compare(got, load_golden(name), rtol=rt, what=name, grad_rtol=grt)
            print(name, 'worst err/tol at (2.5e-2, 4e-2):', round(w, 3))
        except AssertionError as e:
            print(name, 'FAIL at tighter', str(e)[:300])
