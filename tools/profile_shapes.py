"""Per-shape timing of the igemm launches in one UNet evaluation (GPU box).  Dev tool."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from slotdiffusion_amd import _lib, ops

dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model, cfg, _ = bench.build_model(dtype)
model = model.cuda().eval()
model.use_graph = False
recs = []
orig = _lib._call


def timed(fname, stream, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(fname, stream, **kw)
    e1.record()
    recs.append((fname, kw, e0, e1))


TRAIN = len(sys.argv) > 3 and sys.argv[3] == 'train'
if TRAIN:
    model.train()
    model.bank().overlap_wgrad = False
    img = bench.synth_batch(B, 0, 'cuda')

    def step():
        model.grad_arena().zero_()
        out = model(dict(img=img))
        model.calc_train_loss(dict(img=img), out)['denoise_loss'].backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    _lib._call = timed
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib._call = orig
with torch.no_grad():
    if TRAIN:
        sys.argv = sys.argv[:1]
    img = bench.synth_batch(B, 0, 'cuda')
    slots, _ = model.encode(img)
    x = ops.nchw_to_nhwc(torch.randn(B, 3, 32, 32).cuda(), torch.float32, 4)
    t = torch.full((B,), 500., device='cuda')
    for _ in range(0 if TRAIN else 2):
        model._unet_eps(x, t, slots)
    torch.cuda.synchronize()
    _lib._call = timed
    for _ in range(0 if TRAIN else 3):
        model._unet_eps(x, t, slots)
    torch.cuda.synchronize()
    _lib._call = orig
agg = {}
for fname, kw, e0, e1 in recs:
    if fname == 'sdmi_igemm':
        key = (fname, kw['M'], kw['N'], kw['K'], kw['KH'], kw['stride'], kw.get('ups', 0), kw.get('batch', 1))
        fl = 2.0 * kw['M'] * kw['N'] * kw['K'] * max(1, kw.get('batch', 1))
    elif fname == 'sdmi_wgrad':
        key = (fname, kw['M'], kw['N'], kw['K'], kw['KH'], kw['stride'], kw.get('splits', 1))
        fl = 2.0 * kw['M'] * kw['N'] * kw['K']
    else:
        key = (fname,)
        fl = 0
    d = agg.setdefault(key, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1)
    d[2] += fl
tot = sum(d[1] for d in agg.values())
print(f'total {tot / 3:.3f} ms per {"train step" if TRAIN else "UNet eval"} (B={B}, {dtype})')
for key, d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    tf = d[2] / (d[1] * 1e-3) / 1e12 if d[1] > 0 else 0
    print(f'{d[1] / 3:8.3f} ms  n={d[0] // 3:3d}  {tf:7.1f} TF/s  {key}')
