import torch, sys
sys.path.insert(0, '/root/repo')
from slotdiffusion_amd import ops
z = torch.randn(64, 32, 32, 4, device='cuda'); cb = torch.randn(4096, 3, device='cuda')
for _ in range(3): ops.vq_nearest(z, cb, scale=1.0, want_idx=False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): ops.vq_nearest(z, cb, scale=1.0, want_idx=False)
e1.record(); torch.cuda.synchronize(); print('vq us', e0.elapsed_time(e1) * 50)
