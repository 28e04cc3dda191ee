"""Would two half-batch sampling chains on two streams beat one B = 64 chain?  Two model instances (separate caches),
each with its own captured 20-NFE graph at B / 2, replayed concurrently; against the one-graph pass at B.
usage: python tools/exp/two_stream.py [B]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from slotdiffusion_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = 'cuda'


def prep(n):
    m, cfg, _ = bench.build_model(torch.bfloat16)
    m = m.to(dev).eval()
    img = bench.synth_batch(n, 0, dev)
    with torch.no_grad():
        slots, _ = m.encode(img)
    x_T = ops.nchw_to_nhwc(torch.randn(n, 3, 32, 32, device=dev), torch.float32, 4)
    return m, x_T, slots


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    m, x, s = prep(B)
    one = timeit(lambda: m._dpm_sample(x, s))
    print(f'one chain, B = {B}: {one:.2f} ms', flush=True)
    ma, xa, sa = prep(B // 2)
    mb, xb, sb = prep(B // 2)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    half = timeit(lambda: ma._dpm_sample(xa, sa))
    mb._dpm_sample(xb, sb)
    print(f'one chain, B = {B // 2}: {half:.2f} ms', flush=True)

    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            ma._dpm_sample(xa, sa)
        with torch.cuda.stream(s2):
            mb._dpm_sample(xb, sb)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    two = timeit(both)
    print(f'two chains of B = {B // 2} on two streams: {two:.2f} ms  (vs one chain {one:.2f})', flush=True)
