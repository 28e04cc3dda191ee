"""Debug / timing of the direct 3x3 c64 weight-gradient kernel against torch (GPU box)."""
import math, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from slotdiffusion_amd import _lib
B, H, W = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (3, 64, 64))]
splits = int(sys.argv[4]) if len(sys.argv) > 4 else 4
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 64, H, W, generator=g).bfloat16().float().requires_grad_(True)
w = (torch.randn(64, 64, 3, 3, generator=g) / 24).requires_grad_(True)
y = F.conv2d(x, w, None, padding=1)
dy = torch.randn(y.shape, generator=g).bfloat16().float()
y.backward(dy)
ref = w.grad.permute(0, 2, 3, 1).reshape(64, 576)
xd = x.detach().permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
dyd = dy.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
M, K = B * H * W, 576


def run(sp):
    ws = torch.zeros(sp * (64 * K + 64), device='cuda')
    dw = torch.zeros(64, K, device='cuda')
    db = torch.zeros(64, device='cuda')
    _lib.call('sdmi_wgrad', torch.cuda.current_stream().cuda_stream, a=xd.data_ptr(), dy=dyd.data_ptr(),
              dw=dw.data_ptr(), dbias=db.data_ptr(), workspace=ws.data_ptr(), dtype=_lib.BF16, M=M, N=64, K=K,
              lda=64, ldy=64, B=B, H=H, W=W, Cin=64, Ho=H, Wo=W, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, ups=0,
              splits=sp, accumulate=0)
    torch.cuda.synchronize()
    return dw.cpu(), db.cpu()


dw, db = run(splits)
dw1, db1 = run(1)
print('regular vs ref', float((dw1 - ref).norm() / ref.norm()))
print('direct  vs ref', float((dw - ref).norm() / ref.norm()))
print('bias direct vs ref', float((db - dy.sum((0, 2, 3))).norm() / dy.sum((0, 2, 3)).norm()))
e = (dw - ref).view(64, 9, 64)
print('err by tap', [round(float(e[:, t].norm() / ref.view(64, 9, 64)[:, t].norm()), 4) for t in range(9)])
print('err by cout half', [round(float(e[h * 32:(h + 1) * 32].norm()), 4) for h in range(2)])
print('err by cin half', [round(float(e[:, :, h * 32:(h + 1) * 32].norm()), 4) for h in range(2)])
if len(sys.argv) > 5:
    import time
    for sp in (splits,):
        for _ in range(3):
            run(sp)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ws = torch.zeros(sp * (64 * K + 64), device='cuda'); dwt = torch.zeros(64, K, device='cuda'); dbt = torch.zeros(64, device='cuda')
        e0.record()
        for _ in range(10):
            _lib.call('sdmi_wgrad', torch.cuda.current_stream().cuda_stream, a=xd.data_ptr(), dy=dyd.data_ptr(),
                      dw=dwt.data_ptr(), dbias=dbt.data_ptr(), workspace=ws.data_ptr(), dtype=_lib.BF16, M=M, N=64, K=K,
                      lda=64, ldy=64, B=B, H=H, W=W, Cin=64, Ho=H, Wo=W, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, ups=0,
                      splits=sp, accumulate=0)
        e1.record(); torch.cuda.synchronize()
        print(f'splits {sp}: {e0.elapsed_time(e1) * 100:.1f} us per launch (+fold)')
