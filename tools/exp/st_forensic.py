"""What do the wrong tiles of phase A contain?  For each bad (block, pass, wave) tile of q|k|v: which weight unit,
if it had replaced the right one in one K tile of one slice, explains the deviation (stale ring slot = unit g - D)."""
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
    NSL, KT = C // 128, C // 64
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
        args = dict(x=_p(x), tok=_p(tok), qkv=_p(qkv), out=_p(out), gn_gamma=_p(wb.f(n + '.norm.weight')),
                    gn_beta=_p(wb.f(n + '.norm.bias')), wstream_a=_p(wts['wa']), vec_a=_p(wts['va']),
                    wstream_b=_p(wts['wb']), vec_b=_p(wts['vb']), wstream_img=_p(fold['st_img']),
                    vec_img=_p(fold['st_vec']), B=B, S=S, C=C, slots=7, gn_eps=1e-6, ln_eps=1e-5, attn_scale=32.0 ** -0.5)
        # the kernel's own arithmetic restated: folded weights, statistics of the bf16 tok
        f = lambda k: wb.t[k].float()
        g1, b1 = f(t + '.norm1.weight'), f(t + '.norm1.bias')
        wqkv = torch.cat([f(t + '.attn1.to_q.weight'), f(t + '.attn1.to_k.weight'), f(t + '.attn1.to_v.weight')])
        Wp = (wqkv * g1).bfloat16().float()                       # [3C, C]
        Win = f(n + '.proj_in.weight').reshape(C, C).bfloat16().float()
        mats = [Win, Wp[:C], Wp[C:2 * C], Wp[2 * C:]]
        for rep in range(3):
            _lib.call('sdmi_st_block', _st(), phase=1, **args)
            torch.cuda.synchronize()
            tb = tok.float()
            mean = tb.mean(-1, keepdim=True)
            rstd = (tb.var(-1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
            acc = tb @ Wp.t()
            ref = rstd * (acc - mean * Wp.sum(1)) + wqkv @ b1
            dacc = (qkv.float() - ref) / rstd                      # deviation of the accumulators
            found = 0
            for b in range(B):
                for rb in range(S // 64):
                    rows = slice(rb * 64, rb * 64 + 64)
                    for ps in range(3):
                        for w in range(8):
                            cols = slice(ps * C + w * NSL * 16, ps * C + (w + 1) * NSL * 16)
                            d = dacc[b, rows, cols]
                            if float(d.norm()) < 0.05 * float(acc[b, rows, cols].norm()):
                                continue
                            found += 1
                            if found > 4:
                                continue
                            # which slice(s) deviate, and which (kt, replacement unit) explains it
                            msg = []
                            rel = acc[b, rows, cols]
                            pr = [float(d[tt * 16:(tt + 1) * 16].norm() / rel[tt * 16:(tt + 1) * 16].norm()) for tt in range(4)]
                            pc = [float(d[:, c_].norm() / rel[:, c_].norm()) for c_ in range(NSL * 16)]
                            print('     per 16-row tile:', ' '.join(f'{v:.2f}' for v in pr), '| per column:', ' '.join(f'{v:.2f}' for v in pc), flush=True)
                            # is the deviation the (negated) contribution of one 32-wide k-step?
                            for s_ in range(NSL):
                                ds = d[:, s_ * 16:(s_ + 1) * 16]
                                if float(ds.norm()) < 0.3 * float(d.norm()):
                                    continue
                                bestk = (9e9, None)
                                for kk in range(C // 32):
                                    wr = mats[1 + ps][(w * NSL + s_) * 16:(w * NSL + s_) * 16 + 16, kk * 32:kk * 32 + 32]
                                    contrib = tb[b, rows, kk * 32:kk * 32 + 32] @ wr.t()
                                    for sign, tag in ((1.0, 'missing'), (-1.0, 'doubled')):
                                        res = float((ds + sign * contrib).norm() / ds.norm())
                                        if res < bestk[0]:
                                            bestk = (res, f'k-step {kk} {tag}')
                                print(f'     slice {s_}: k-step hypothesis {bestk[1]} residual {bestk[0]:.2f}', flush=True)
                            for s_ in range(NSL):
                                ds = d[:, s_ * 16:(s_ + 1) * 16]
                                if float(ds.norm()) < 0.3 * float(d.norm()):
                                    continue
                                best = (9e9, None)
                                for kt in range(KT):
                                    xk = tb[b, rows, kt * 64:(kt + 1) * 64]
                                    wright = mats[1 + ps][(w * NSL + s_) * 16:(w * NSL + s_) * 16 + 16, kt * 64:kt * 64 + 64]
                                    g_right = (1 + ps) * KT * NSL + kt * NSL + s_
                                    # candidate: zero unit
                                    cands = [('zero', torch.zeros_like(wright))]
                                    for g2 in range(4 * KT * NSL):
                                        m2, r2 = divmod(g2, KT * NSL)
                                        kt2, s2 = divmod(r2, NSL)
                                        cands.append((f'unit {g2 - g_right:+d}', mats[m2][(w * NSL + s2) * 16:(w * NSL + s2) * 16 + 16, kt2 * 64:kt2 * 64 + 64]))
                                    for tag, wc in cands:
                                        res = float((ds - xk @ (wc - wright).t()).norm() / ds.norm())
                                        if res < best[0]:
                                            best = (res, f'kt {kt} <- {tag}')
                                msg.append(f'slice {s_}: best {best[1]} (residual {best[0]:.2f})')
                            print(f'  {name} rep {rep} img {b} blk {rb} pass {ps} wave {w}: ' + '; '.join(msg), flush=True)
            print(f'{name} rep {rep}: {found} bad (block, pass, wave) tiles', flush=True)
