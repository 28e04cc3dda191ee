"""fp64 run of the CPU oracle's train step -> tests/golden/oracle_fp64_grads.npz.

The reference's own fp32 CPU gradients of the slot-encoder side deviate from the exact (fp64)
gradients by 0.1-0.5 % (ill-conditioned eps-renormalised attention); this fixture lets the GPU
tests also check the HIP backward against the exact values.  Produced by OUR oracle (already pinned
against the reference in tests/test_oracle_golden.py), not by the reference.  Lives under tests/
because it executes the oracle (test infrastructure: only tests/ may).

    python tests/gen_oracle_fp64_grads.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import slotdiff_oracle as O           # noqa: E402
from slotdiffusion_amd import spec               # noqa: E402
from tests import common as C                    # noqa: E402

torch.set_num_threads(8)
_te = O.timestep_embedding
O.timestep_embedding = lambda t, dim, mp=10000: _te(t, dim, mp).double()
cfg, G = C.clevrtex_cfg(), C.load_golden()
img = C.make_inputs(2)[0]
keys = C.load_keys()['img_based/SADiffusion/clevrtex-7slot']
frozen = set(keys['frozen'])
train = [k for k, _ in keys['params'] if k not in frozen]
W = {k: (v.double() if v.is_floating_point() else v) for k, v in C.oracle_weights(cfg).items()}
for k in train:
    W[k].requires_grad_(True)
slots, _ = O.sa_encode(W, img.double(), spec.resnet18_plan(False), 3, training=True)
loss, _, _ = O.ldm_loss(W, spec.unet_plan(cfg['dec_dict']['unet_dict']),
                        cfg['dec_dict']['vae_dict']['enc_dec_dict'], img.double(), slots, G['t'],
                        G['noise'].double())
loss.backward()
out = {'loss': np.float64(loss.item())}
for k in G:
    if k.startswith('grad:'):
        out[k] = W[k[5:]].grad.float().numpy()
out['grad_norms'] = np.array([float(W[str(n)].grad.norm()) for n in G['grad_norms_names']],
                             dtype=np.float32)
np.savez_compressed(os.path.join(C.GOLD, 'oracle_fp64_grads.npz'), **out)
print({k: getattr(v, 'shape', v) for k, v in out.items()})
