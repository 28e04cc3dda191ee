"""Where a K tile of igemm_pp_kernel goes (library built with -DSDMI_PP_TIMELINE: `bash tools/exp/build_variant.sh
igemm.hip pp_tl -DSDMI_PP_TIMELINE`, run with SDMI_LIBPATH=tools/exp/libsdmi_pp_tl.so): s_memtime sums of wave 0
(group 0) and wave 4 (group 1) of every workgroup."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from slotdiffusion_amd import _lib
from slotdiffusion_amd.ops import _p
dev = 'cuda'
NAMES = ('L: frag-read + DMA issue', 'lgkmcnt(0) wait', 'barrier', 'MFMA segment', 'vmcnt(6) wait', 'un-stagger barrier',
         'epilogue', 'zero acc + stagger barrier')
for B, H, C, N in ((64, 32, 256, 256), (64, 32, 128, 128), (64, 64, 128, 128)):
    w = (torch.randn(N, 9 * C, device=dev) / (9 * C) ** 0.5).bfloat16()
    x = torch.randn(B, H, H, C, device=dev).bfloat16()
    y = torch.empty(B, H, H, N, device=dev, dtype=torch.bfloat16)
    ws = torch.zeros(8 * 1024, dtype=torch.int64, device=dev)
    for it in range(3):
        _lib.call('sdmi_igemm', torch.cuda.current_stream().cuda_stream, a=_p(x), w=_p(w), out=_p(y), dtype=_lib.BF16,
                  workspace=_p(ws), out_dtype=_lib.BF16, M=B * H * H, N=N, K=9 * C, lda=C, ldw=9 * C, ldc=N, B=B, H=H, W=H,
                  Cin=C, Ho=H, Wo=H, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, act=0, alpha=1.0, split_k=1, batch=1)
    torch.cuda.synchronize()
    t = ws.view(-1, 2, 8).double().cpu()
    t = t[t.sum((1, 2)) > 0]
    tiles = (B * H * H // 256) * (N // 128)
    nkt = 9 * C // 64
    per = tiles / t.shape[0]
    if os.environ.get('COMPACT'):
        m = t[:, 0].median(0).values / (per * nkt)
        print(f'{os.environ.get("SDMI_LIBPATH", "")[-12:]} C={C} N={N} H={H}: per K tile (group 0) total {float(m.sum()):6.0f} | L-issue {m[0]:5.0f} lgkm {m[1]:4.0f} '
              f'barrier {m[2]:5.0f} MFMA {m[3]:5.0f} vmcnt {m[4]:4.0f} epi {m[6]:4.0f}')
        continue
    print(f'conv3x3 B={B} H={H} C={C} N={N}: {t.shape[0]} workgroups x {per:.1f} tiles x {nkt} K tiles; shader cycles, median workgroup')
    for g in (0, 1):
        m = t[:, g].median(0).values
        tot = float(m.sum())
        print(f'  group {g} (wave {4 * g}): total {tot:.0f} = {tot / (per * nkt):.0f} per K tile')
        for nm, v in zip(NAMES, m.tolist()):
            print(f'    {nm:28s} {v:10.0f}  {100 * v / tot:5.1f} %   {v / (per * nkt):7.1f} per K tile')
