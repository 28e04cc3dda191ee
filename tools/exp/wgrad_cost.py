"""What the weight gradients cost inside the train step: the measurement-only switches that
used to live in kern.py (SDMI_EXP_SKIP_WGRAD / SDMI_EXP_WGRAD_TINY) as a monkeypatch of the launch function, so the
product path carries no switch that skips work.  usage: python tools/exp/wgrad_cost.py [skip|tiny]
(gradients are WRONG in both modes; timing only)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from slotdiffusion_amd import kern, _lib

mode = sys.argv[1] if len(sys.argv) > 1 else 'skip'
real_call = kern.call


def patched(fname, stream, **kw):
    if fname == 'sdmi_wgrad':
        if mode == 'skip':
            return
        if kw.get('KH', 1) > 1:                       # one m-step of work, same launch
            kw.update(B=1, Ho=min(kw['Ho'], max(1, 64 // kw['Wo'])))
            kw['M'] = kw['Ho'] * kw['Wo']
        else:
            kw['M'] = kw['B'] = min(kw['M'], 64)
        kw['splits'] = 1
    return real_call(fname, stream, **kw)


for tag, fn in (('baseline', real_call), (mode, patched)):
    kern.call = fn
    m, cfg, _ = bench.build_model(torch.bfloat16)
    m = m.cuda().train()
    m.bank().pair_bwd = False                         # weight gradients as their own launches
    from slotdiffusion_amd.optim import FusedAdam, GraphedTrainStep
    img = bench.synth_batch(64, 0, 'cuda')
    step = GraphedTrainStep(m, FusedAdam(m, lr=1e-4, dec_lr=2e-4, clip_grad=1.0, total_steps=1000), dict(img=img))
    for _ in range(3):
        step(dict(img=img))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        step(dict(img=img))
    torch.cuda.synchronize()
    print(f'{tag}: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms per train step', flush=True)
