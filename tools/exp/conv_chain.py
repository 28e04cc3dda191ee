"""Dependent chains (HIP graph replay) of the UNet's 3x3 convolution shapes, us per launch.
SDMI_IGEMM_DMA=0/1/2 selects the LDS-DMA kernels (igemm.hip dispatch)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from slotdiffusion_amd import _lib
from slotdiffusion_amd.ops import _p

dev = 'cuda'
CH = 30


def chain_time(fn, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / CH


WS = torch.empty(16 * 4096 * 512, device=dev)


def conv(x, w, y, B, H, C, N, k=3):
    auto = B * H * H <= 4096
    _lib.call('sdmi_igemm', torch.cuda.current_stream().cuda_stream, a=_p(x), w=_p(w), out=_p(y), dtype=_lib.BF16,
              workspace=(_p(WS) if auto else 0),
              out_dtype=_lib.BF16, M=B * H * H, N=N, K=k * k * C, lda=C, ldw=k * k * C, ldc=N, B=B, H=H, W=H, Cin=C,
              Ho=H, Wo=H, KH=k, KW=k, stride=1, pad_t=k // 2, pad_l=k // 2, act=0, alpha=1.0,
              split_k=(0 if auto else 1), batch=1)


tag = os.environ.get('SDMI_IGEMM_DMA', '3') + ' LW=' + os.environ.get('SDMI_IGEMM_DMA_LW', '8') + ' ALL=' + os.environ.get('SDMI_IGEMM_DMA_ALL', '0')
SHAPES = [(64, 32, 128, 128), (64, 32, 256, 128), (64, 32, 256, 256), (64, 16, 256, 256), (64, 16, 512, 256),
          (64, 8, 384, 384), (64, 64, 128, 128), (256, 16, 256, 256)]
if os.environ.get('SHAPES') == 'low':      # the 8^2 / 4^2 levels (64 x 64 tiles, split-K)
    SHAPES = [(64, 8, 384, 384), (64, 8, 768, 384), (64, 8, 640, 384), (64, 4, 512, 512), (64, 4, 1024, 512),
              (64, 4, 896, 512), (64, 8, 256, 384), (16, 16, 256, 256)]
    tag += ' BK256=' + os.environ.get('SDMI_IGEMM_BK256', '0') + ' SPLIT=' + os.environ.get('SDMI_IGEMM_SPLIT_TARGET', '384')
for B, H, C, N in SHAPES:
    w = (torch.randn(N, 9 * C, device=dev) / (9 * C) ** 0.5).bfloat16()
    xs = [torch.randn(B, H, H, C, device=dev).bfloat16(), torch.empty(B, H, H, N, device=dev, dtype=torch.bfloat16)]
    if C == N:
        def fn():
            a, b = xs
            for _ in range(CH):
                conv(a, w, b, B, H, C, N)
                a, b = b, a
    else:
        def fn():
            for _ in range(CH):
                conv(xs[0], w, xs[1], B, H, C, N)
    us = chain_time(fn)
    fl = 2.0 * B * H * H * N * 9 * C
    print(f'DMA={tag} conv3x3 B={B:3d} H={H:3d} C={C:4d} N={N:4d}: {us:7.2f} us  {fl / us / 1e6:7.1f} TF/s', flush=True)

# correctness of whatever kernel the dispatch picked: against torch's fp32 convolution on the bf16-rounded operands
import torch.nn.functional as F
for B, H, C, N in [(64, 16, 256, 256), (64, 32, 128, 128), (64, 8, 384, 384), (64, 4, 1024, 512)]:
    g = torch.Generator(device=dev).manual_seed(B + H + C)
    w = (torch.randn(N, 3, 3, C, device=dev, generator=g) / (9 * C) ** 0.5).bfloat16()
    x = torch.randn(B, H, H, C, device=dev, generator=g).bfloat16()
    y = torch.empty(B, H, H, N, device=dev, dtype=torch.bfloat16)
    conv(x, w, y, B, H, C, N)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    err = float((y.float() - ref).norm() / ref.norm())
    print(f'check B={B} H={H} C={C} N={N}: rel-L2 {err:.2e}', 'OK' if err < 5e-3 else 'WRONG', flush=True)
