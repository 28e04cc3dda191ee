"""Which Python lines launch which libsdmi entry points during one eager train step (dev tool).
    python tools/call_sites.py [entry-name filter]"""
import collections
import os
import sys
import traceback
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from slotdiffusion_amd import _lib

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ''
hits = collections.Counter()
orig = _lib._call
ON = [False]


def spy(fname, stream, **kw):
    if ON[0] and flt in fname:
        st = [f for f in traceback.extract_stack()[:-1] if f.filename.startswith(root) and 'tools/' not in f.filename
              and not f.filename.endswith(('_lib.py', 'ops.py'))]
        hits[(fname, ' < '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(st[-3:])))] += 1
    return orig(fname, stream, **kw)


_lib._call = spy
model, cfg, _ = bench.build_model(torch.bfloat16)
model = model.cuda().train()
model.use_graph = False
img = bench.synth_batch(64, 0, 'cuda')


def step():
    model.grad_arena().zero_()
    out = model(dict(img=img))
    model.calc_train_loss(dict(img=img), out)['denoise_loss'].backward()


step()
ON[0] = True
step()
torch.cuda.synchronize()
for (f, s), n in sorted(hits.items(), key=lambda kv: -kv[1])[:40]:
    print(f'{n:4d}  {f:24s} {s}')
