"""Isolated timing of the non-GEMM launches of one train step (GPU box): each distinct
(function, integer arguments) is replayed back-to-back REPS times between two events."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from slotdiffusion_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SKIP = ('sdmi_igemm', 'sdmi_wgrad', 'sdmi_pack_dgrad', 'sdmi_pack_dgrad_batch')
REPS = 20
model, cfg, _ = bench.build_model(torch.bfloat16)
model = model.cuda()
model.train()
model.bank().overlap_wgrad = False
img = bench.synth_batch(B, 0, 'cuda')
recs = []
orig = _lib._call


def rec(fname, stream, **kw):
    if fname not in SKIP:
        recs.append((fname, dict(kw)))
    orig(fname, stream, **kw)


def step():
    model.grad_arena().zero_()
    out = model(dict(img=img))
    loss = model.calc_train_loss(dict(img=img), out)['denoise_loss']
    loss.backward()


step()
_lib._call = rec
step()
_lib._call = orig
torch.cuda.synchronize()
agg = {}
for fname, kw in recs:
    key = (fname,) + tuple((k, v) for k, v in sorted(kw.items())
                           if isinstance(v, int) and v < (1 << 24) and k not in ('seed',))
    agg.setdefault(key, [0, kw])[0] += 1
rows = []
st = torch.cuda.current_stream().cuda_stream
for key, (cnt, kw) in agg.items():
    for _ in range(2):
        orig(key[0], st, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(REPS):
        orig(key[0], st, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / REPS
    rows.append((us * cnt, cnt, us, key))
rows.sort(reverse=True)
print(f'non-GEMM total {sum(r[0] for r in rows) / 1e3:.3f} ms over {sum(r[1] for r in rows)} launches')
for r in rows[:70]:
    dims = ' '.join(f'{k}={v}' for k, v in r[3][1:] if k not in ('dtype', 'accumulate'))
    print(f'{r[0] / 1e3:7.3f} ms n={r[1]:3d} {r[2]:8.1f} us  {r[3][0][5:]:18s} {dims}')
