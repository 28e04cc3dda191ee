"""Latency model of sdmi_wgrad: time vs contraction length at fixed N, K (GPU box)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from slotdiffusion_amd import _lib
from slotdiffusion_amd.kern import _DT

st = torch.cuda.current_stream().cuda_stream
for dtype in (torch.bfloat16, torch.float32):
    for (N, K) in ((512, 512), (128, 128)):
        for M in (128, 256, 1024, 4096, 16384):
            for splits in (1, 4):
                for bias in (0, 1):
                    x = torch.randn(M, K, device='cuda').to(dtype)
                    dy = torch.randn(M, N, device='cuda').to(dtype)
                    ws = torch.empty(splits * (N * K + N), device='cuda')
                    dw = torch.zeros(N, K, device='cuda')
                    db = torch.zeros(N, device='cuda')
                    kw = dict(a=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(),
                              dbias=(db.data_ptr() if bias else 0), workspace=ws.data_ptr(),
                              dtype=_DT[dtype], M=M, N=N, K=K, lda=K, ldy=N, B=M, H=1, W=1, Cin=K,
                              Ho=1, Wo=1, KH=1, KW=1, stride=1, pad_t=0, pad_l=0, ups=0,
                              splits=splits, accumulate=1)
                    for _ in range(3):
                        _lib.call('sdmi_wgrad', st, **kw)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(20):
                        _lib.call('sdmi_wgrad', st, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    print(f'{str(dtype)[6:]:9s} N={N} K={K} M={M:6d} splits={splits} bias={bias}: '
                          f'{e0.elapsed_time(e1) * 50:.1f} us')
