"""Debug build (-DST_VERIFY -DST_DRAIN): phase A with every ring fragment compared against the unit in memory."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
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
        ref, tok_ref, qkv_ref = ref_block(n, x, ctx, heads, parts=True)
        wts = wb.st_fused_weights(n, x.dtype)
        S = hw * hw
        tok = torch.zeros((B, S, C), dtype=x.dtype, device=dev)
        qkv = torch.zeros((B, S, 3 * C), dtype=x.dtype, device=dev)
        out = torch.zeros_like(x)
        cnt = torch.zeros(64, dtype=torch.int32, device=dev)
        args = dict(x=_p(x), tok=_p(tok), qkv=_p(qkv), out=_p(out), gn_gamma=_p(wb.f(n + '.norm.weight')),
                    gn_beta=_p(wb.f(n + '.norm.bias')), wstream_a=_p(wts['wa']), vec_a=_p(wts['va']),
                    wstream_b=_p(wts['wb']), vec_b=_p(wts['vb']), wstream_img=_p(fold['st_img']),
                    vec_img=_p(cnt), B=B, S=S, C=C, slots=7, gn_eps=1e-6, ln_eps=1e-5, attn_scale=32.0 ** -0.5)
        for rep in range(2):
            cnt.zero_()
            _lib.call('sdmi_st_block', _st(), phase=1, **args)
            torch.cuda.synchronize()
            d = (qkv.float() - qkv_ref).reshape(B, S // 64, 64, 3, 8, C // 8)
            r = qkv_ref.reshape(B, S // 64, 64, 3, 8, C // 8)
            e_wave = (d.pow(2).sum((0, 1, 2, 5)) / r.pow(2).sum((0, 1, 2, 5))).sqrt()
            print(f'{name} rep {rep}: qkv err per (pass, wave):', ' | '.join(' '.join(f'{float(v):.0e}' for v in row) for row in e_wave))
            print('    ring-fragment mismatches per (gemm: proj_in, q, k, v; wave):', cnt[:32].view(4, 8).tolist())
            # phase B on the reference tok / qkv, counters through gn_gamma
            tok.copy_(tok_ref); qkv.copy_(qkv_ref)
            cnt.zero_()
            a2 = dict(args); a2['gn_gamma'] = _p(cnt); a2['vec_img'] = _p(fold['st_vec'])
            _lib.call('sdmi_st_block', _st(), phase=2, **a2)
            torch.cuda.synchronize()
            d = (out.float() - ref).reshape(B, S // 64, 64, 8, C // 8)
            r = ref.reshape(B, S // 64, 64, 8, C // 8)
            e_wave = (d.pow(2).sum((0, 1, 2, 4)) / r.pow(2).sum((0, 1, 2, 4))).sqrt()
            print('    phase B out err per wave:', ' '.join(f'{float(v):.0e}' for v in e_wave))
            print('    phase B mismatches per (to_out, scores, crossout, x2, ff1, ff2; wave):', cnt[:48].view(6, 8).tolist())
