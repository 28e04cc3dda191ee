"""Repeatability probe of the fused backward phases (sdmi_st_train_bwd): each phase N times on fixed inputs; reports how
many runs differ from the first and where (rows / columns) the differing elements sit."""
import sys
import torch
sys.path.insert(0, '.')
from slotdiffusion_amd import _lib, kern


def streams(C, seed=0):
    g = torch.Generator().manual_seed(seed)
    mats = {k: (torch.randn(sh, generator=g) * 0.05).bfloat16().cuda() for k, sh in
            dict(po=(C, C), ff2=(C, 4 * C), ff1=(8 * C, C), o2=(C, C), q2=(C, C), o=(C, C), q=(C, C), k=(C, C), v=(C, C),
                 **{'in': (C, C)}).items()}
    units = kern.st_train_units(C)
    wbk = kern.WeightBank.__new__(kern.WeightBank)
    out = {}
    import numpy as np
    descs = []
    for k in ('b1', 'b2', 'ba'):
        out[k] = torch.empty(sum(len(w) for w in units[k]) * 1024, dtype=torch.bfloat16, device='cuda')
        descs.append(kern.WeightBank._st_descs(wbk, units[k], mats, out[k]))
    arr = np.concatenate(descs)
    tab = torch.from_numpy(arr.view(np.uint8).copy()).cuda()
    _lib.call('sdmi_st_pack', torch.cuda.current_stream().cuda_stream, descs=tab.data_ptr(), n_units=len(arr))
    torch.cuda.synchronize()
    return out


def probe(C, S, B, rows, phase, N=40):
    g = torch.Generator().manual_seed(C + S + phase)
    r = lambda *sh: torch.randn(*sh, generator=g).bfloat16().cuda()
    st = streams(C)
    stat = lambda: torch.stack([torch.randn(B * S, generator=g) * 0.1, torch.rand(B * S, generator=g) + 0.5], -1).cuda().contiguous()
    gam = (1 + 0.1 * torch.randn(C, generator=g)).cuda()
    nwg = B * S // rows
    P = lambda t: t.data_ptr()
    e = lambda *sh: torch.empty(*sh, dtype=torch.bfloat16, device='cuda')
    part = torch.zeros(nwg * C * 2, device='cuda')
    if phase == 1:
        ins = dict(dout=r(B, S, C), h=r(B, S, 8 * C), x2=r(B, S, C), st3=stat(), ln3_g=gam)
        outs = dict(dx3=e(B, S, C), dh=e(B, S, 8 * C), dx2=e(B, S, C), da2=e(B, S, C), ln3_part=part)
        kw = dict(wstream_b1=P(st['b1']))
    elif phase == 2:
        ins = dict(dq2=r(B, S, C), x1=r(B, S, C), st2=stat(), ln2_g=gam, dx2=r(B, S, C))
        outs = dict(dx1=e(B, S, C), da1=e(B, S, C), ln2_part=part)
        kw = dict(wstream_b2=P(st['b2']))
    else:
        ins = dict(dqkv=r(B, S, 3 * C), tok=r(B, S, C), st1=stat(), ln1_g=gam, dx1=r(B, S, C))
        outs = dict(dtok=e(B, S, C), dhgn=e(B, S, C), ln1_part=part)
        kw = dict(wstream_a=P(st['ba']))
    first, ndiff, where = None, 0, {}
    for i in range(N):
        for o in outs.values():
            o.fill_(0)
        _lib.call('sdmi_st_train_bwd', torch.cuda.current_stream().cuda_stream, phase=phase, B=B, S=S, C=C, rows=rows,
                  **{k: P(v) for k, v in ins.items()}, **{k: P(v) for k, v in outs.items()}, **kw)
        torch.cuda.synchronize()
        cur = {k: v.clone() for k, v in outs.items()}
        if first is None:
            first = cur
            continue
        bad = False
        for k in cur:
            d = (cur[k].float() != first[k].float())
            if d.any():
                bad = True
                idx = d.view(-1, d.shape[-1]).nonzero()
                rows_ = sorted(set(idx[:, 0].tolist()))[:6]
                cols_ = sorted(set(idx[:, 1].tolist()))
                where.setdefault(k, []).append((int(d.sum()), rows_, cols_[:8], len(cols_),
                                                float((cur[k].float() - first[k].float()).abs().max())))
        ndiff += bad
    print(f'C={C} S={S} B={B} rows={rows} phase={phase}: {ndiff}/{N - 1} runs differ from the first', flush=True)
    for k, v in where.items():
        print('   ', k, v[:3], flush=True)


if __name__ == '__main__':
    cfgs = ((256, 256, 64, 64), (384, 64, 5, 64), (384, 64, 64, 32)) if len(sys.argv) < 2 else ((256, 256, 2, 64), (384, 64, 5, 64), (256, 256, 8, 32), (384, 64, 16, 32), (256, 256, 64, 64))
    for C, S, B, rows in cfgs:
        for ph in (1, 2, 3):
            probe(C, S, B, rows, ph)
