"""Isolated per-shape timing of the GEMM-family launches of one train step / UNet eval (GPU box).
Each distinct (kernel, shape) is replayed back-to-back REPS times between two events."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from slotdiffusion_amd import _lib, ops

mode = sys.argv[1] if len(sys.argv) > 1 else 'unet'
dtype = torch.bfloat16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
REPS = 20
model, cfg, _ = bench.build_model(dtype)
model = model.cuda()
model.use_graph = False
recs = []
orig = _lib._call


def rec(fname, stream, **kw):
    if fname in ('sdmi_igemm', 'sdmi_wgrad'):
        recs.append((fname, dict(kw)))
    orig(fname, stream, **kw)


img = bench.synth_batch(B, 0, 'cuda')
keep = []
if mode == 'unet':
    model.eval()
    with torch.no_grad():
        slots, _ = model.encode(img)
        x = ops.nchw_to_nhwc(torch.randn(B, 3, 32, 32).cuda(), torch.float32, 4)
        t = torch.full((B,), 500., device='cuda')
        model._unet_eps(x, t, slots)
        _lib._call = rec
        keep.append(model._unet_eps(x, t, slots))
        _lib._call = orig
else:
    model.train()
    model.grad_arena()
    out = model(dict(img=img))
    loss = model.calc_train_loss(dict(img=img), out)['denoise_loss']
    loss.backward()
    _lib._call = rec
    out = model(dict(img=img))
    loss = model.calc_train_loss(dict(img=img), out)['denoise_loss']
    loss.backward()
    _lib._call = orig
torch.cuda.synchronize()
# NOTE: buffers referenced by the recorded pointers may have been freed; re-running only reads /
# writes inside torch's cached blocks (still mapped) -- fine for timing, results are discarded.
agg = {}
for fname, kw in recs:
    key = (fname, kw['M'], kw['N'], kw['K'], kw['KH'], kw['stride'], kw.get('ups', 0), kw.get('zins', 0),
           kw.get('batch', 1), kw.get('splits', 0))
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
    fl = 2.0 * kw['M'] * kw['N'] * kw['K'] * max(1, kw.get('batch', 1))
    rows.append((us * cnt, cnt, us, fl / (us * 1e-6) / 1e12, key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f'{mode} B={B}: GEMM-family total {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} launches')
for r in rows[:60]:
    print(f'{r[0] / 1e3:8.3f} ms  n={r[1]:3d}  {r[2]:8.1f} us  {r[3]:7.1f} TF/s  {r[4]}')
