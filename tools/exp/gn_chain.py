"""GroupNorm (+SiLU) in dependent chains replayed from a HIP graph, us per launch, next to an elementwise copy
(sdmi_act, no activation) of the same tensor: how far is the single-pass kernel from a pure load -> store pass?"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from slotdiffusion_amd import ops

dev = 'cuda'
CH = 40


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


for B, HW, C in [(64, 16, 512), (64, 16, 1024), (64, 64, 384), (64, 64, 768), (64, 256, 256), (64, 256, 512),
                 (64, 1024, 128), (64, 1024, 256)]:
    x = torch.randn(B, HW, C, device=dev).bfloat16()
    y = torch.empty_like(x)
    ga, be = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    bufs = [x, y]

    def gn():
        a, b = bufs
        for _ in range(CH):
            ops.group_norm(a, ga, be, eps=1e-5, act='silu', out=b)
            a, b = b, a

    def cp():
        a, b = bufs
        for _ in range(CH):
            b = ops.act(a, None)
            a = b
    t_gn, t_cp = chain_time(gn), chain_time(cp)
    mb = 2 * x.numel() * 2 / 1e6
    print(f'B={B} HW={HW:5d} C={C:5d} ({mb:6.1f} MB in+out): GroupNorm+SiLU {t_gn:6.2f} us ({mb / t_gn / 1e3:5.2f} TB/s) | '
          f'copy {t_cp:6.2f} us ({mb / t_cp / 1e3:5.2f} TB/s)', flush=True)
