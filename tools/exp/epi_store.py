"""EXPERIMENT (apply tools/exp/igemm_ablation.patch to csrc/igemm_body.h and rebuild first): where an igemm
launch spends its time.  Run-time switches ride in the `act` bits: 0x100 one store per tile, 0x800 no fragment
reads / MFMAs, 0x1000 no global loads, 0x2000 no LDS writes, 0x4000 no epilogue.  Results are wrong on purpose;
the numbers of round 3 are in profiles/r03_igemm_ablation.txt."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from slotdiffusion_amd import _lib
from slotdiffusion_amd.ops import _p

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


def lin(x, w, y, act, M, N, K):
    _lib.call('sdmi_igemm', torch.cuda.current_stream().cuda_stream, a=_p(x), w=_p(w), out=_p(y), dtype=_lib.BF16,
              out_dtype=_lib.BF16, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, B=M, H=1, W=1, Cin=K, Ho=1, Wo=1, KH=1, KW=1,
              stride=1, act=act, alpha=1.0, split_k=1, batch=1)


def conv(x, w, y, act, B, H, C, N):
    _lib.call('sdmi_igemm', torch.cuda.current_stream().cuda_stream, a=_p(x), w=_p(w), out=_p(y), dtype=_lib.BF16,
              out_dtype=_lib.BF16, M=B * H * H, N=N, K=9 * C, lda=C, ldw=9 * C, ldc=N, B=B, H=H, W=H, Cin=C, Ho=H,
              Wo=H, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, act=act, alpha=1.0, split_k=1, batch=1)


for M, K, N in [(16384, 256, 256), (16384, 256, 768), (16384, 256, 2048), (16384, 1024, 256), (4096, 384, 384),
                (4096, 384, 3072), (1024, 512, 512), (1024, 512, 4096), (65536, 128, 128)]:
    w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = []
    for act in (0, 0x100, 0x4000, 0x3900, 0x7900, 0x7100):
        def fn():
            for _ in range(CH):
                lin(x, w, y, act, M, N, K)
        res.append(chain_time(fn))
    print(f'linear M={M:6d} K={K:5d} N={N:5d}: full {res[0]:7.2f} us | no stores {res[1]:7.2f} | no epilogue {res[2]:7.2f} | '
          f'all off but epilogue math {res[3]:7.2f} | all off, no epilogue {res[4]:7.2f} | only MFMA+frag reads {res[5]:7.2f}', flush=True)
for B, H, C, N in [(64, 32, 128, 128), (64, 16, 256, 256), (64, 8, 384, 384), (64, 4, 512, 512), (64, 32, 256, 256)]:
    w = (torch.randn(N, 9 * C, device=dev) / (9 * C) ** 0.5).bfloat16()
    x = torch.randn(B, H, H, C, device=dev).bfloat16()
    y = torch.empty(B, H, H, N, device=dev, dtype=torch.bfloat16)
    res = []
    for act in (0, 0x100, 0x4000, 0x3900, 0x7900, 0x7100):
        def fn():
            for _ in range(CH):
                conv(x, w, y, act, B, H, C, N)
        res.append(chain_time(fn))
    print(f'conv3x3 B={B} H={H:3d} C={C:4d} N={N:4d}: full {res[0]:7.2f} us | no stores {res[1]:7.2f} | no epilogue {res[2]:7.2f} | '
          f'all off but epilogue math {res[3]:7.2f} | all off, no epilogue {res[4]:7.2f} | only MFMA+frag reads {res[5]:7.2f}', flush=True)
