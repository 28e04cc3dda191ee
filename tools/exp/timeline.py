import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from slotdiffusion_amd import _lib
B, H, C, N, k = 64, 32, 256, 256, 3
x = torch.randn(B, H, H, C, device='cuda').bfloat16()
w = (torch.randn(N, k * k * C, device='cuda') / 48).bfloat16()
y = torch.empty(B, H, H, N, device='cuda', dtype=torch.bfloat16)
dbg = torch.zeros(64 * 8, dtype=torch.int64, device='cuda')
st = torch.cuda.current_stream().cuda_stream
kw = dict(a=x.data_ptr(), w=w.data_ptr(), out=y.data_ptr(), dtype=_lib.BF16, out_dtype=_lib.BF16, M=B*H*H, N=N, K=k*k*C,
          lda=C, ldw=k*k*C, ldc=N, B=B, H=H, W=H, Cin=C, Ho=H, Wo=H, KH=k, KW=k, stride=1, pad_t=1, pad_l=1, alpha=1.0,
          split_k=1, batch=1, workspace=dbg.data_ptr())
for _ in range(3):
    _lib.call('sdmi_igemm', st, **kw)
torch.cuda.synchronize()
d = dbg.cpu().view(64, 8)
t0 = int(d[2, 0])
print('g: loader[start, loads-landed, stored, issued] | mfma[at-barrier, past-barrier, done]   (cycles rel., s_memtime ticks)')
for g in range(2, 40):
    r = [int(v) - t0 for v in d[g, :7]]
    print(g, r[:4], '|', r[4:7], ' period', int(d[g, 5]) - int(d[g - 1, 5]))
