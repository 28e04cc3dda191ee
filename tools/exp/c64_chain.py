"""Dependent chain of the 64 -> 64 channel 3x3 convolution at 128^2 (slot encoder / VQ-VAE, B = 64): us per launch.
SDMI_IGEMM_HALO_C64=0: conv3x3_c64_kernel; 1: the halo-staged kernel's 256 x 64 tiles (igemm_halo.h)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from slotdiffusion_amd import ops
import torch.nn.functional as F
dev = 'cuda'
for B, H, C in ((64, 128, 64), (64, 64, 64)):
    w = (torch.randn(C, 3, 3, C, device=dev) / (9 * C) ** 0.5).bfloat16()
    bias = torch.randn(C, device=dev)
    xs = [torch.randn(B, H, H, C, device=dev).bfloat16(), torch.empty(B, H, H, C, device=dev, dtype=torch.bfloat16)]
    def fn():
        a, b = xs
        for _ in range(10):
            ops.conv2d(a, w, bias, out=b)
            a, b = b, a
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    fl = 2.0 * B * H * H * C * 9 * C
    x = torch.randn(2, H, H, C, device=dev).bfloat16()
    print(f'C64={os.environ.get("SDMI_IGEMM_HALO_C64", "1")} conv3x3 B={B} H={H} C={C}: {us:7.2f} us {fl / us / 1e6:7.1f} TF/s', flush=True)
