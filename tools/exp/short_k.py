"""Where do the short-K transformer GEMMs spend their time?  Dependent chains replayed from a HIP graph
(us per launch): a trivial kernel, plain linears of the UNet's transformer shapes, and the FF1 layer
(LayerNorm fold + GEGLU) with each epilogue feature switched on separately."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
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


print(f'trivial kernel chain: {chain_time(trivial):.2f} us per launch', flush=True)
for M, K, N in [(16384, 256, 256), (16384, 64, 256), (16384, 256, 768), (16384, 256, 2048), (16384, 1024, 256),
                (4096, 384, 384), (4096, 384, 3072), (1024, 512, 512), (1024, 512, 4096), (65536, 128, 128)]:
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def fn():
        for _ in range(CH):
            ops.linear(x, w, out=y)
    us = chain_time(fn)
    print(f'linear M={M:6d} K={K:5d} N={N:5d}: {us:7.2f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s', flush=True)

# FF1: out[M, N] = geglu(LN(x) W^T), W [2N, K]
for M, K, N in [(16384, 256, 1024), (4096, 384, 1536), (1024, 512, 2048)]:
    w = (torch.randn(2 * N, K, device=dev) / K ** 0.5).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    colsum = w.float().sum(1).contiguous()
    bias = torch.zeros(2 * N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    y2 = torch.empty(M, 2 * N, device=dev, dtype=torch.bfloat16)
    for tag, kw, out in (('plain 2N', dict(), y2), ('LN fold, 2N', dict(ln_colsum=colsum, ln_eps=1e-5), y2),
                         ('GEGLU', dict(geglu=True), y), ('LN fold + GEGLU', dict(ln_colsum=colsum, ln_eps=1e-5, geglu=True), y)):
        def fn():
            for _ in range(CH):
                ops.linear(x, w, bias, out=out, n=(N if kw.get('geglu') else 2 * N), **kw)
        us = chain_time(fn)
        print(f'FF1 M={M:6d} K={K:4d} N={N:5d} {tag:16s}: {us:7.2f} us  {4.0 * M * N * K / us / 1e6:7.1f} TF/s', flush=True)
    # the two-kernel alternative: LayerNorm kernel + plain / GEGLU GEMM
    g_, b_ = torch.ones(K, device=dev), torch.zeros(K, device=dev)

    def fn2():
        for _ in range(CH):
            xn = ops.layer_norm(x, g_, b_, eps=1e-5)
            ops.linear(xn, w, bias, out=y, n=N, geglu=True)
    try:
        us = chain_time(fn2)
        print(f'FF1 M={M:6d} K={K:4d} N={N:5d} LN kernel + GEGLU GEMM: {us * 2:7.2f} us per pair', flush=True)
    except Exception as e:                     # noqa: BLE001
        print('LN + GEGLU variant failed:', e)
