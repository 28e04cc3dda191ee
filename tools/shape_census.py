"""Census of every libsdmi launch of one train step / UNet evaluation (GPU box, dev tool): each
distinct (entry point, shape) is replayed back-to-back REPS times between two events, so the table
shows where the time of a step goes per shape, isolated from launch gaps.

    python tools/shape_census.py train|unet [B] [top]
"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from slotdiffusion_amd import _lib, ops

mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
TOP = int(sys.argv[3]) if len(sys.argv) > 3 else 90
REPS = 20
dtype = torch.bfloat16
model, cfg, _ = bench.build_model(dtype)
model = model.cuda()
model.use_graph = False
model.bank().overlap_wgrad = False
recs = []
orig = _lib._call
PTR = {n: {f for f, t in fl if t is _lib.ctypes.c_void_p} for n, fl in _lib.STRUCTS.items()}


def rec(fname, stream, **kw):
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
    for it in range(2):
        if it == 1:
            _lib._call = rec
        out = model(dict(img=img))
        loss = model.calc_train_loss(dict(img=img), out)['denoise_loss']
        loss.backward()
    _lib._call = orig
torch.cuda.synchronize()
agg = {}
for fname, kw in recs:
    ptrs = PTR[_lib.FUNCS[fname]]
    key = (fname,) + tuple((k, v) for k, v in sorted(kw.items())
                           if k not in ptrs and isinstance(v, int) and k not in ('seed',))
    key += tuple((k, 1) for k in sorted(kw) if k in ptrs and kw[k] and k in
                 ('residual', 'rowvec', 'bias', 'dbias', 'workspace'))
    agg.setdefault(key, [0, kw])[0] += 1
rows = []
st = torch.cuda.current_stream().cuda_stream
for key, (cnt, kw) in agg.items():
    try:
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
    except Exception as ex:                      # noqa: BLE001
        us = float('nan')
    fl = 0.0
    if key[0] in ('sdmi_igemm', 'sdmi_wgrad'):
        fl = 2.0 * kw['M'] * kw['N'] * kw['K'] * max(1, kw.get('batch', 1))
    rows.append((us * cnt, cnt, us, fl / (us * 1e-6) / 1e12 if us == us and us > 0 else 0.0, key))
rows.sort(key=lambda r: -(r[0] if r[0] == r[0] else 0))
tot = sum(r[0] for r in rows if r[0] == r[0])
print(f'{mode} B={B}: total {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} launches, '
      f'{len(rows)} distinct')
fam = {}
for r in rows:
    f = fam.setdefault(r[4][0], [0.0, 0])
    f[0] += r[0] if r[0] == r[0] else 0
    f[1] += r[1]
for k, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f'  family {t / 1e3:8.3f} ms  n={c:4d}  {k}')
SHOW = ('M', 'N', 'K', 'KH', 'stride', 'ups', 'zins', 'batch', 'splits', 'split_k', 'osy', 'B', 'HW', 'C',
        'H', 'W', 'Cin', 'rows', 'n', 'heads', 'Sq', 'Skv', 'act', 'dtype', 'out_dtype', 'groups', 'rows_per',
        'R', 'D', 'residual', 'rowvec', 'bias', 'Ca', 'Cb', 'cols')
for r in rows[:TOP]:
    d = dict(r[4][1:])
    s = ' '.join(f'{k}={d[k]}' for k in SHOW if k in d)
    print(f'{r[0] / 1e3:8.3f} ms  n={r[1]:3d}  {r[2]:8.1f} us  {r[3]:7.1f} TF/s  {r[4][0][5:]}  {s}')
