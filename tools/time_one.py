"""Time ONE implicit-GEMM conv shape (events around 20 back-to-back launches)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from slotdiffusion_amd import ops

B, H, C, N = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 32, 256, 256)))
k = int(sys.argv[5]) if len(sys.argv) > 5 else 3
x = torch.randn(B, H, H, C, device='cuda').bfloat16()
w = (torch.randn(N, k * k * C, device='cuda') / (k * k * C) ** 0.5).bfloat16()
b = torch.zeros(N, device='cuda')
for _ in range(3):
    y = ops.conv2d(x, w, b, kh=k, kw=k, stride=1, pad=(k // 2,) * 4)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(20):
    y = ops.conv2d(x, w, b, kh=k, kw=k, stride=1, pad=(k // 2,) * 4)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 50
fl = 2.0 * B * H * H * N * k * k * C
print(f'{os.environ.get("TAG", "")} conv B={B} H={H} C={C} N={N} k={k}: {us:.1f} us  {fl / us / 1e6:.1f} TF/s')
