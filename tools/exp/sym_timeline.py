"""Where a K tile of igemm_sym_kernel goes (library built with -DSDMI_SYM_TIMELINE, SDMI_LIBPATH): s_memtime sums
of wave 0 of every workgroup -- counted vmcnt wait, barrier, DMA issue, fragment reads + MFMAs, epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from slotdiffusion_amd import _lib
from slotdiffusion_amd.ops import _p
dev = 'cuda'
os.environ.setdefault('SDMI_IGEMM_SYM', '4')
for B, H, C, N in ((64, 32, 256, 256), (64, 32, 128, 128), (64, 16, 256, 256)):
    w = (torch.randn(N, 9 * C, device=dev) / (9 * C) ** 0.5).bfloat16()
    x = torch.randn(B, H, H, C, device=dev).bfloat16()
    y = torch.empty(B, H, H, N, device=dev, dtype=torch.bfloat16)
    ws = torch.zeros(8 * 1024, dtype=torch.int64, device=dev)
    for it in range(3):
        _lib.call('sdmi_igemm', torch.cuda.current_stream().cuda_stream, a=_p(x), w=_p(w), out=_p(y), dtype=_lib.BF16,
                  workspace=_p(ws), out_dtype=_lib.BF16, M=B * H * H, N=N, K=9 * C, lda=C, ldw=9 * C, ldc=N, B=B, H=H, W=H,
                  Cin=C, Ho=H, Wo=H, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, act=0, alpha=1.0, split_k=1, batch=1)
    torch.cuda.synchronize()
    t = ws.view(-1, 8)[:, :6].double().cpu()
    t = t[t.sum(1) > 0]
    tiles = (B * H * H // 128) * (N // 128)
    nkt = 9 * C // 64
    per_wg_tiles = tiles / t.shape[0]
    m = t.median(0).values
    tot = float(m.sum())
    print(f'SYM={os.environ["SDMI_IGEMM_SYM"]} conv3x3 B={B} H={H} C={C} N={N}: {t.shape[0]} workgroups x {per_wg_tiles:.1f} tiles x {nkt} K tiles; '
          f'shader cycles of wave 0 (median workgroup, total {tot:.0f}):')
    for nm, v in zip(('vmcnt wait', 'barrier', 'DMA issue', 'fragment reads + MFMA', 'epilogue', 'prologue'), m.tolist()):
        print(f'    {nm:22s} {v:10.0f}  {100 * v / tot:5.1f} %   {v / (per_wg_tiles * nkt):7.1f} per K tile')
