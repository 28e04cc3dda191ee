"""Which Python lines of the train step launch ATen (non-libsdmi) device kernels?  (GPU box, dev tool)
    python tools/find_aten.py
Eager step under torch.profiler with stacks; kernels are attributed to the innermost frame inside
this repository."""
import collections
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from torch.profiler import ProfilerActivity, profile

model, cfg = bench.build_model(torch.bfloat16)
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith('aten::') and ev.kernels:
        site = 'unknown'
        for fr in ev.stack:
            if root in fr and 'tools/find_aten' not in fr:
                site = fr.replace(root + '/', '')
                break
        agg[(ev.name, site[:110])] += len(ev.kernels)
for (name, site), n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f'{n:4d}  {name:28s} {site}')
