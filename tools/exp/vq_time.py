"""sdmi_vq_nearest at the sampler's shape ([64][32][32] latents x 4096 codes): dependent-chain time per launch."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from slotdiffusion_amd import ops
z = torch.randn(64, 32, 32, 4, device='cuda')
cb = torch.randn(4096, 3, device='cuda') / 3 ** 0.5
ops.vq_nearest(z, cb)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ops.vq_nearest(z, cb)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.vq_nearest(z, cb, want_idx=False)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
print(f'SDMI_VQ_REG={os.environ.get("SDMI_VQ_REG", "1")}: {e0.elapsed_time(e1) * 1e3 / 100:.1f} us per launch')
