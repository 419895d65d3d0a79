"""Golden fixtures of the YOLOX PAFPN recorded from the UNMODIFIED reference (imported from /root/reference).

TEST INFRASTRUCTURE; runs only in the authoring container.  Usage: python oracle/make_golden_fpn.py
For every case of tests/casegen_fpn.py: build the reference YOLOPAFPN (yolox_extension/models/yolo_pafpn.py), assert its
state_dict names / shapes equal rvt_amd.fpn.YOLOPAFPN's, load the numpy-seeded parameters, record
  * the eval-mode forward (running statistics),
  * the training-mode forward + backward of  L = sum_i <out_i, cot_i>: outputs, every parameter gradient, the input gradients, and
    the BatchNorm running statistics after the step,
in tests/golden/<case>.npz.  Only numerical outputs are stored."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, '_stubs'))
sys.path.insert(1, '/root/reference')
sys.path.insert(2, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests import casegen_fpn as cg  # noqa: E402


def main():
    from models.detection.yolox_extension.models.yolo_pafpn import YOLOPAFPN as RefFPN
    from rvt_amd.fpn import YOLOPAFPN as OurFPN
    torch.set_num_threads(4)
    for name, c in cg.CASES.items():
        ref = RefFPN(depth=c['depth'], in_stages=(2, 3, 4), in_channels=c['in_channels'])
        ours = OurFPN(depth=c['depth'], in_stages=(2, 3, 4), in_channels=c['in_channels'])
        shapes = [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
        assert shapes == [(k, tuple(v.shape)) for k, v in ours.state_dict().items()], 'state_dict mismatch'
        params = cg.make_params(name, shapes)
        ref.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        xs = {s: torch.from_numpy(a) for s, a in cg.make_inputs(name).items()}
        cots = [torch.from_numpy(a) for a in cg.make_cotangents(name)]
        out = {'names': np.array([k for k, _ in shapes])}
        ref.eval()
        with torch.no_grad():
            for i, o in enumerate(ref(xs)):
                out[f'eval_out{i}'] = o.numpy()
        ref.train()
        xg = {s: x.clone().requires_grad_(True) for s, x in xs.items()}
        outs = ref(xg)
        loss = sum((o * ct).sum() for o, ct in zip(outs, cots))
        loss.backward()
        out['loss'] = np.array(float(loss))
        for i, o in enumerate(outs):
            out[f'train_out{i}'] = o.detach().numpy()
        for s in (2, 3, 4):
            out[f'dx{s}'] = xg[s].grad.numpy()
        for k, p in ref.named_parameters():
            g = p.grad.double().numpy().reshape(-1)
            if g.size <= 4096:
                out[f'grad/{k}'] = p.grad.numpy()                         # small tensors element for element
            else:                                                           # large ones: l2 norm, sum, 512 evenly spaced samples
                out[f'gradstat/{k}'] = np.array([np.sqrt((g * g).sum()), g.sum()])
                out[f'gradsamp/{k}'] = g[np.linspace(0, g.size - 1, 512).astype(np.int64)].astype(np.float32)
        for k, b in ref.named_buffers():
            if not k.endswith('num_batches_tracked'):
                out[f'buf/{k}'] = b.numpy()
        np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', f'{name}.npz'), **out)
        print(name, 'loss', float(loss), 'params', sum(p.numel() for p in ref.parameters()))


if __name__ == '__main__':
    main()
