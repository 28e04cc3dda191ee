"""Launch ONE implicit-GEMM shape a few times (for rocprofv3 --pmc studies of the igemm kernel)."""
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
for _ in range(12):
    y = ops.conv2d(x, w, b, kh=k, kw=k, stride=1, pad=(k // 2,) * 4)
torch.cuda.synchronize()
print('ok', tuple(y.shape))
