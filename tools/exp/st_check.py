"""Fused SpatialTransformer block (sdmi_st_block) against a torch fp32 restatement on the GPU and against the
per-layer HIP launches; then dependent-chain timings (HIP graph) of both forms.
usage: python tools/exp/st_check.py [B]"""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench
from slotdiffusion_amd import kern, _lib
from slotdiffusion_amd.kern import _p, _st

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = 'cuda'
model, cfg, _ = bench.build_model(torch.bfloat16)
model = model.to(dev).eval()
K = model.K()
wb = K.wb
u = model.unet()
P = u.P


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def ref_block(n, x, ctx, heads, parts=False):
    """fp32 torch restatement (attention.py:297-308, 247-251); x [B,H,W,C], ctx [B,N,D]."""
    f = lambda k: wb.t[k].float()
    t = n + '.transformer_blocks.0'
    Bq, H, W, C = x.shape
    xf = x.float()
    h = F.group_norm(xf.permute(0, 3, 1, 2), 32, f(n + '.norm.weight'), f(n + '.norm.bias'), 1e-6).permute(0, 2, 3, 1)
    tok = h.reshape(Bq, H * W, C) @ f(n + '.proj_in.weight').reshape(C, C).t() + f(n + '.proj_in.bias')
    tok0 = tok

    def attn(q, k, v):
        q, k, v = [z.reshape(Bq, -1, heads, 32).transpose(1, 2) for z in (q, k, v)]
        a = torch.softmax(q @ k.transpose(-1, -2) * 32 ** -0.5, -1) @ v
        return a.transpose(1, 2).reshape(Bq, -1, C)
    n1 = F.layer_norm(tok, (C,), f(t + '.norm1.weight'), f(t + '.norm1.bias'))
    q, k, v = n1 @ f(t + '.attn1.to_q.weight').t(), n1 @ f(t + '.attn1.to_k.weight').t(), n1 @ f(t + '.attn1.to_v.weight').t()
    tok = attn(q, k, v) @ f(t + '.attn1.to_out.0.weight').t() + f(t + '.attn1.to_out.0.bias') + tok
    n2 = F.layer_norm(tok, (C,), f(t + '.norm2.weight'), f(t + '.norm2.bias'))
    c = ctx.float()
    tok = attn(n2 @ f(t + '.attn2.to_q.weight').t(), c @ f(t + '.attn2.to_k.weight').t(), c @ f(t + '.attn2.to_v.weight').t()) \
        @ f(t + '.attn2.to_out.0.weight').t() + f(t + '.attn2.to_out.0.bias') + tok
    n3 = F.layer_norm(tok, (C,), f(t + '.norm3.weight'), f(t + '.norm3.bias'))
    hg = n3 @ f(t + '.ff.net.0.proj.weight').t() + f(t + '.ff.net.0.proj.bias')
    xg, gate = hg.chunk(2, -1)
    tok = (xg * F.gelu(gate)) @ f(t + '.ff.net.2.weight').t() + f(t + '.ff.net.2.bias') + tok
    out = tok @ f(n + '.proj_out.weight').reshape(C, C).t() + f(n + '.proj_out.bias') + xf.reshape(Bq, H * W, C)
    out = out.reshape(Bq, H, W, C)
    return (out, tok0, torch.cat([q, k, v], -1)) if parts else out


def chain_time(fn, n=20, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / n


gen = torch.Generator(device=dev).manual_seed(5)
for name, hw in (('input_blocks.4.1', 16), ('input_blocks.7.1', 8), ('output_blocks.5.1', 8), ('output_blocks.8.1', 16)):
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
        kvp = {'kv': kv, 'fold': fold}
        ref, tok_ref, qkv_ref = ref_block(n, x, ctx, heads, parts=True)
        # phase A alone
        wts = wb.st_fused_weights(n, x.dtype)
        S = hw * hw
        tok = torch.zeros((B, S, C), dtype=x.dtype, device=dev)
        qkv = torch.zeros((B, S, 3 * C), dtype=x.dtype, device=dev)
        out = torch.zeros_like(x)
        args = dict(x=_p(x), tok=_p(tok), qkv=_p(qkv), out=_p(out), gn_gamma=_p(wb.f(n + '.norm.weight')),
                    gn_beta=_p(wb.f(n + '.norm.bias')), wstream_a=_p(wts['wa']), vec_a=_p(wts['va']),
                    wstream_b=_p(wts['wb']), vec_b=_p(wts['vb']), wstream_img=_p(fold['st_img']),
                    vec_img=_p(fold['st_vec']), B=B, S=S, C=C, slots=7, gn_eps=1e-6, ln_eps=1e-5, attn_scale=32.0 ** -0.5)
        _lib.call('sdmi_st_block', _st(), phase=1, **args)
        torch.cuda.synchronize()
        print(f'{name} C={C} S={S} B={B}: phase A tok rel {rel(tok, tok_ref):.3e}  qkv rel {rel(qkv, qkv_ref):.3e}', flush=True)
        # phase B on the REFERENCE's tok / qkv (isolates phase B)
        tok.copy_(tok_ref)
        qkv.copy_(qkv_ref)
        _lib.call('sdmi_st_block', _st(), phase=2, **args)
        torch.cuda.synchronize()
        print(f'   phase B (on reference tok / qkv): out rel {rel(out, ref):.3e}', flush=True)
        fused = K.st_fused(x, n, heads, kvp)
        kern._ST_FUSED = False
        unf = u._st(K, name, x, heads, kvp)
        kern._ST_FUSED = True
        torch.cuda.synchronize()
        print(f'   fused vs fp32 {rel(fused, ref):.3e}   per-layer launches vs fp32 {rel(unf, ref):.3e}   fused vs per-layer '
              f'{rel(fused, unf):.3e}   finite {bool(torch.isfinite(fused.float()).all())}', flush=True)
        if B >= 16:
            tf = chain_time(lambda: K.st_fused(x, n, heads, kvp))
            kern._ST_FUSED = False
            tu = chain_time(lambda: u._st(K, name, x, heads, kvp))
            kern._ST_FUSED = True
            fl = 2.0 * B * S * (C * C * 4 + C * C + 2 * 128 * C + 8 * C * C + 5 * C * C) + 4.0 * B * S * S * C
            print(f'   chain: fused {tf:.1f} us ({fl / tf / 1e6:.0f} TF/s)   per-layer {tu:.1f} us', flush=True)
