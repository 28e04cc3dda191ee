"""Graph-replayed dependent chains of one linear layer (x <- x W^T, N = K) and of a trivial kernel:
us per launch inside a HIP graph.  Dev tool for the small-GEMM latency study."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from slotdiffusion_amd import ops, _lib

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


cnt = torch.zeros(4, dtype=torch.int64, device=dev)


def trivial():
    for _ in range(CH):
        _lib.call('sdmi_counters_inc', torch.cuda.current_stream().cuda_stream, step=0, seed=cnt.data_ptr())


print(f'trivial kernel chain: {chain_time(trivial):.2f} us per launch')
for M, K, N in [(16384, 256, 256), (8192, 256, 256), (4096, 256, 256), (32768, 256, 256), (16384, 128, 256),
                (16384, 512, 256), (16384, 256, 512), (1024, 512, 512), (4096, 384, 384), (65536, 256, 256),
                (16384, 256, 1024), (16384, 1024, 256)]:
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    xs = [torch.randn(M, K, device=dev).bfloat16(), torch.empty(M, N, device=dev, dtype=torch.bfloat16)]
    if N == K:
        def fn():
            a, b = xs
            for _ in range(CH):
                ops.linear(a, w, out=b)
                a, b = b, a
    else:
        def fn():
            for _ in range(CH):
                ops.linear(xs[0], w, out=xs[1])
    us = chain_time(fn)
    print(f'linear M={M:6d} K={K:5d} N={N:5d}: {us:7.2f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s  '
          f'{(M * K + M * N + N * K) * 2 / us / 1e6:6.2f} TB/s algorithmic')
