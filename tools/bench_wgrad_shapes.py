"""Graph-replayed sdmi_wgrad (+ its fold) per UNet shape for several M splits: us per launch pair.
Dev tool for the weight-gradient study."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from slotdiffusion_amd import _lib
from slotdiffusion_amd.kern import _DT

CH = 20
dt = torch.bfloat16


def run(M, N, K, kh, splits, B=64):
    Cin = K // (kh * kh)
    HW = M // B
    H = int(round(HW ** 0.5))
    x = torch.randn(B, H, H, Cin, device='cuda').to(dt)
    dy = torch.randn(M, N, device='cuda').to(dt)
    ws = torch.empty(max(1, splits) * (N * K + N) + 64, device='cuda')
    dw = torch.zeros(N, K, device='cuda')
    kw = dict(a=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(), dbias=0, workspace=ws.data_ptr(),
              dtype=_DT[dt], M=M, N=N, K=K, lda=Cin, ldy=N, B=B, H=H, W=H, Cin=Cin, Ho=H, Wo=H, KH=kh, KW=kh,
              stride=1, pad_t=kh // 2, pad_l=kh // 2, ups=0, splits=splits, accumulate=1)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        st = s.cuda_stream
        _lib.call('sdmi_wgrad', st, **kw)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(CH):
                _lib.call('sdmi_wgrad', torch.cuda.current_stream().cuda_stream, **kw)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 5 / CH


shapes = [(1024, 512, 512, 1), (4096, 384, 384, 1), (16384, 256, 256, 1), (16384, 2048, 256, 1), (1024, 512, 4608, 3),
          (4096, 384, 3456, 3), (16384, 256, 2304, 3), (65536, 256, 2304, 3), (65536, 128, 1152, 3), (262144, 128, 1152, 3)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for M, N, K, kh in shapes:
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    cur = max(1, min((192 + tiles - 1) // tiles, M // 512, 512))
    if M <= 1024:
        cur = 1
    row = []
    for sp in sorted({1, 2, 4, 8, 16, 32, 64, cur}):
        if M // sp < 64:
            continue
        us = run(M, N, K, kh, sp)
        row.append(f'{sp}{"*" if sp == cur else ""}:{us:.1f}')
    print(f'M={M:7d} N={N:5d} K={K:5d} k={kh}  tiles={tiles:3d}  ' + '  '.join(row) +
          f'   [GF {2.0 * M * N * K / 1e9:.1f}]', flush=True)
