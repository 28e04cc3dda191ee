"""Which Python lines of the train step run ATen (non-libsdmi) device work?  (GPU box, dev tool)
    python tools/find_aten.py
Python-level wrappers around the usual suspects (zeros / zero_ / fill_ / copying contiguous() /
casts / arithmetic / RNG / indexing) count calls per call site inside this repository during one
eager step; accumulations done by the autograd engine itself are what remains of the profiler's
ATen kernel count."""
import collections
import os
import sys
import traceback
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hits = collections.Counter()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.startswith(root) and 'tools/find_aten' not in fr.filename:
            return f'{fr.filename.replace(root + "/", "")}:{fr.lineno}'
    return 'outside'


def wrap(obj, name, cond=None, tag=None):
    orig = getattr(obj, name)

    def f(*a, **k):
        if ON[0] and (cond is None or cond(*a, **k)):
            hits[(tag or name, site())] += 1
        return orig(*a, **k)
    setattr(obj, name, f)


T = torch.Tensor
wrap(torch, 'zeros'); wrap(torch, 'randn'); wrap(torch, 'randint'); wrap(torch, 'full'); wrap(torch, 'ones')
wrap(torch, 'stack'); wrap(torch, 'cat'); wrap(torch, 'tensor')
for n in ('zero_', 'fill_', 'add_', 'mul_', 'copy_', 'clone', 'float', 'long', 'to', '__add__', '__mul__',
          '__sub__', '__neg__', '__getitem__', 'sum', 'expand', 'repeat'):
    c = None
    if n in ('float', 'long', 'to'):
        c = lambda self, *a, **k: self.is_cuda
    if n == '__getitem__':
        c = lambda self, idx: self.is_cuda and torch.is_tensor(idx)
    if n in ('expand',):
        c = lambda *a, **k: False
    wrap(T, n, c)
wrap(T, 'contiguous', lambda self, *a, **k: self.is_cuda and not self.is_contiguous(), 'contiguous(copy)')

model, cfg, _ = bench.build_model(torch.bfloat16)
model = model.cuda().train()
model.use_graph = False
from slotdiffusion_amd.optim import FusedAdam
opt = FusedAdam(model, lr=1e-4, dec_lr=2e-4, clip_grad=1.0, total_steps=1000)
img = bench.synth_batch(16, 0, 'cuda')


def step():
    opt.zero_grad()
    out = model(dict(img=img))
    loss = model.calc_train_loss(dict(img=img), out)['denoise_loss']
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
ON[0] = True
step()
ON[0] = False
torch.cuda.synchronize()
for (name, s), n in sorted(hits.items(), key=lambda kv: -kv[1]):
    print(f'{n:4d}  {name:18s} {s}')
