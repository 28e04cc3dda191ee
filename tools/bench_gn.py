"""GroupNorm forward / backward timing over the shapes of the train step (GPU box, dev tool).
SDMI_GN_T / SDMI_GN_S force the single-pass geometry (one process per setting).

    python tools/bench_gn.py [check]
"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from slotdiffusion_amd import _lib, ops
from slotdiffusion_amd._lib import call

SHAPES = [(64, 1024, 128), (64, 1024, 256), (64, 1024, 384), (64, 256, 256), (64, 256, 384),
          (64, 256, 512), (64, 256, 640), (64, 64, 384), (64, 64, 768), (64, 16, 512), (64, 16, 1024),
          (64, 4096, 128), (64, 16384, 64)]
REPS = 30
tag = f"T={os.environ.get('SDMI_GN_T', '-')} S={os.environ.get('SDMI_GN_S', '-')}"
check = len(sys.argv) > 1 and sys.argv[1] == 'check'


def p(t):
    return 0 if t is None else t.data_ptr()


for dt in (torch.bfloat16,):
    for B, HW, C in SHAPES:
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(B, HW, C, generator=g) * 1.3 + 0.2).to(dt).cuda()
        dy = torch.randn(B, HW, C, generator=g).to(dt).cuda()
        gamma = (1 + 0.1 * torch.randn(C, generator=g)).cuda()
        beta = (0.1 * torch.randn(C, generator=g)).cuda()
        st = torch.cuda.current_stream().cuda_stream
        y, stats = ops.group_norm(x, gamma, beta, eps=1e-5, act='silu', return_stats=True)
        dx = torch.empty_like(x)
        dg, db = torch.zeros(C, device='cuda'), torch.zeros(C, device='cuda')
        nsplit = max(1, min(16, HW // 64))
        partial = torch.empty((B * nsplit * C * 2 + B * 32 * 2,), dtype=torch.float32, device='cuda')
        kw = dict(x=p(x), dy=p(dy), dx=p(dx), gamma=p(gamma), beta=p(beta), stats=p(stats), dgamma=p(dg),
                  dbeta=p(db), partial=p(partial), dtype=_lib.BF16 if dt == torch.bfloat16 else _lib.F32,
                  B=B, HW=HW, C=C, groups=32, act=_lib.ACT['silu'], nsplit=nsplit, residual=0,
                  dresidual=0, accumulate=0)
        call('sdmi_groupnorm_bwd', st, **kw)
        if check:
            xr = x.float().cpu().requires_grad_(True)
            yr = torch.nn.functional.silu(torch.nn.functional.group_norm(
                xr.transpose(1, 2), 32, gamma.cpu(), beta.cpu(), 1e-5)).transpose(1, 2)
            yr.backward(dy.float().cpu())
            e1 = float((y.float().cpu() - yr).abs().max())
            e2 = float((dx.float().cpu() - xr.grad).abs().max() / xr.grad.abs().max())
            print(f'  check B={B} HW={HW} C={C}: y err {e1:.3e}  dx rel err {e2:.3e}', flush=True)
        res = []
        out = torch.empty_like(x)
        part_f = torch.empty((B * nsplit * 32 * 2,), dtype=torch.float32, device='cuda')
        kwf = dict(x=p(x), y=p(out), gamma=p(gamma), beta=p(beta), stats=p(stats), partial=p(part_f),
                   dtype=kw['dtype'], B=B, HW=HW, C=C, groups=32, eps=1e-5, act=_lib.ACT['silu'],
                   nsplit=nsplit, residual=0)
        for which in ('fwd', 'bwd'):
            fn = (lambda: call('sdmi_groupnorm', st2, **kwf)) if which == 'fwd' else \
                (lambda: call('sdmi_groupnorm_bwd', st2, **kw))
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                st2 = side.cuda_stream
                for _ in range(3):
                    fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                st2 = torch.cuda.current_stream().cuda_stream
                for _ in range(REPS):
                    fn()
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / REPS / 3)
        mb = B * HW * C * 2 / 1e6
        print(f'{tag}  B={B} HW={HW:5d} C={C:4d}  fwd {res[0]:7.1f} us ({2 * mb / res[0]:5.2f} TB/s)  '
              f'bwd {res[1]:7.1f} us ({3 * mb / res[1]:5.2f} TB/s)', flush=True)
