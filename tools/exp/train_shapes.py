"""Per-shape timing of the GEMM-shaped launches of one eager train step (GPU box): sdmi_igemm, sdmi_bwd_pair
(dgrad + wgrad shapes read back from the nested argument structs), sdmi_wgrad.  Dev tool."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from slotdiffusion_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model, cfg, _ = bench.build_model(torch.bfloat16)
model = model.cuda().train()
model.use_graph = False
img = bench.synth_batch(B, 0, 'cuda')
recs = []
orig = _lib._call
G = _lib.CSTRUCT['SdmiGemmArgs']
W = _lib.CSTRUCT['SdmiWgradArgs']


def timed(fname, stream, **kw):
    if fname == 'sdmi_bwd_pair':
        d = ctypes.cast(kw['dgrad'], ctypes.POINTER(G)).contents
        w = ctypes.cast(kw['wgrad'], ctypes.POINTER(W)).contents
        key = (fname, d.M, d.N, d.K, d.KH, d.stride, 'w', w.M, w.N, w.K, w.splits, kw['wgrad_tile'], kw['dgrad_cap'])
        fl = 2.0 * d.M * d.N * d.K + 2.0 * w.M * w.N * w.K
    elif fname == 'sdmi_igemm':
        key = (fname, kw['M'], kw['N'], kw['K'], kw['KH'], kw['stride'], kw.get('ups', 0), kw.get('batch', 1),
               'a2' if kw.get('a2') else '', 'sk%d' % kw.get('split_k', 0))
        fl = 2.0 * kw['M'] * kw['N'] * kw['K'] * max(1, kw.get('batch', 1))
    elif fname == 'sdmi_wgrad':
        key = (fname, kw['M'], kw['N'], kw['K'], kw['KH'], kw['stride'], kw.get('splits', 1))
        fl = 2.0 * kw['M'] * kw['N'] * kw['K']
    else:
        key, fl = (fname,), 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(fname, stream, **kw)
    e1.record()
    recs.append((key, fl, e0, e1))


def step():
    model.grad_arena().zero_()
    out = model(dict(img=img))
    model.calc_train_loss(dict(img=img), out)['denoise_loss'].backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
_lib._call = timed
R = 3
for _ in range(R):
    step()
torch.cuda.synchronize()
_lib._call = orig
agg = {}
for key, fl, e0, e1 in recs:
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1)
    d[2] += fl
tot = sum(d[1] for d in agg.values())
print(f'total {tot / R:.3f} ms per train step (B={B}, eager, event-timed: launch gaps included)')
by = {}
for key, d in agg.items():
    by[key[0]] = by.get(key[0], 0.0) + d[1] / R
print('  '.join(f'{k} {v:.2f}' for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]))
for key, d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 90]:
    tf = d[2] / (d[1] * 1e-3) / 1e12 if d[1] > 0 else 0
    print(f'{d[1] / R:8.3f} ms  n={d[0] // R:3d}  {d[1] / d[0] * 1e3:7.1f} us  {tf:7.1f} TF/s  {key}')
