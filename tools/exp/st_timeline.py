"""Phase timeline of st_block_b_kernel (-DST_TIMELINE build): s_memtime of wave 0 at the phase boundaries."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
src = open(os.path.join(os.path.dirname(__file__), 'st_check.py')).read().split("gen = torch.Generator")[0]
sys.argv = [sys.argv[0], str(B)]
exec(compile(src, 'st_check_head', 'exec'))
gen = torch.Generator(device=dev).manual_seed(5)
for name, hw in (('input_blocks.4.1', 16), ('input_blocks.7.1', 8)):
    n = P + name
    heads = u.heads_of[name]
    C = heads * 32
    x = torch.randn(B, hw, hw, C, device=dev, generator=gen).bfloat16()
    slots = torch.randn(B, 7, 192, device=dev, generator=gen)
    with torch.no_grad():
        ctx = model._ctx(slots)
        t = n + '.transformer_blocks.0'
        kv = K.linear_multi(ctx, [(t + '.attn2.to_k.weight', t + '.attn2.to_v.weight')])[0]
        fold = K.cross_prepare(kv, t, heads)
        wts = wb.st_fused_weights(n, x.dtype)
        S = hw * hw
        tok = torch.zeros((B, S, C), dtype=x.dtype, device=dev)
        qkv = torch.zeros((B, S, 3 * C), dtype=x.dtype, device=dev)
        out = torch.zeros_like(x)
        nwg = B * S // 64
        stamps = torch.zeros(nwg * 16, dtype=torch.int64, device=dev)
        args = dict(x=_p(x), tok=_p(tok), qkv=_p(qkv), out=_p(out), gn_gamma=_p(wb.f(n + '.norm.weight')),
                    gn_beta=_p(wb.f(n + '.norm.bias')), wstream_a=_p(wts['wa']), vec_a=_p(wts['va']),
                    wstream_b=_p(wts['wb']), vec_b=_p(wts['vb']), wstream_img=_p(fold['st_img']),
                    vec_img=_p(fold['st_vec']), B=B, S=S, C=C, slots=7, gn_eps=1e-6, ln_eps=1e-5, attn_scale=32.0 ** -0.5)
        sa = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
        a1 = dict(args); a1['vec_img'] = _p(sa)
        for rep in range(3):
            _lib.call('sdmi_st_block', _st(), phase=1, **a1)
        torch.cuda.synchronize()
        ta = sa.view(nwg, 8)[:, :6].double().cpu()
        da = ta[:, 1:] - ta[:, :-1]
        print(f'{name} C={C} S={S} phase A (cycles, median workgroup):', ' | '.join(f'{nm} {float(da[:, i].median()):.0f}' for i, nm in enumerate(('GroupNorm statistics', 'normalise -> operand buffer', 'proj_in GEMM', 'tok epilogue', 'q | k | v passes'))))
        _lib.call('sdmi_st_block', _st(), phase=1, **args)
        a2 = dict(args); a2['gn_gamma'] = _p(stamps)
        for rep in range(3):
            _lib.call('sdmi_st_block', _st(), phase=2, **a2)
            torch.cuda.synchronize()
        full = stamps.view(nwg, 16).double().cpu(); st = full[:, :7]
        for a_, b_, nm in ((0, 8, 'start -> first K/V staged'), (8, 9, 'attention rounds'), (9, 10, 'barrier'), (10, 11, 'tok residual fetched'), (11, 1, 'ring prologue + O -> Y + barrier')):
            dd = full[:, b_] - full[:, a_]
            print(f'      [{nm}] median {float(dd.median()):9.0f} max {float(dd.max()):9.0f}')
        d = (st[:, 1:] - st[:, :-1])
        ff = full[:, 12:16]
        print('      FF loop, sums over the chunks (wave 0): FF1 steps', float(ff[:,0].median()), ' GEGLU + G write', float(ff[:,1].median()), ' barrier', float(ff[:,2].median()), ' FF2 steps + vector fetch', float(ff[:,3].median()))
        names = ['attention + ring prologue', '-', 'to_out gemm + epilogue', 'cross-attention', 'x2 gemm + ff loop', 'final epilogue']
        print(f'{name} C={C} S={S}: s_memtime ticks per phase (median over {nwg} workgroups; ticks = shader cycles):')
        for i, nm in enumerate(names):
            print(f'   {nm:14s} median {float(d[:, i].median()) :9.0f} ticks   max {float(d[:, i].max()):9.0f}')
        tot = st[:, 6] - st[:, 0]
        print(f'   total median {float(tot.median()) * 0.01:.2f} us, span over all workgroups {float(st[:, 6].max() - st[:, 0].min()) * 0.01:.2f} us')
