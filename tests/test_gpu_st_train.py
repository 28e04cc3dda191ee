"""Training twin of the fused SpatialTransformer block (sdmi.h: sdmi_st_train_fwd / sdmi_st_pack; csrc/st_train.hip)
-- GPU parity tests through the C ABI.

Forward: every tensor the fused launches store (GroupNorm output, the three LayerNorm outputs and their row statistics,
q | k | v, both attention outputs and their log-sum-exp rows, the GEGLU pre-activations ...) against a plain fp32 torch
restatement of the reference module (attention.py:297-308, 247-251, 182-206, 44-65) on the bf16-rounded weights;
block output and gradients (input, slot keys / values, every parameter of the block) against the per-layer launches the
fused path replaces; run-to-run repeatability (the eight waves run free inside a GEMM phase)."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests import common as C

pytestmark = pytest.mark.gpu


def _model(seed=3):
    from slotdiffusion_amd.models import SADiffusion
    cfg = C.clevrtex_cfg()
    m = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'], cfg['loss_dict'],
                    compute_dtype=torch.bfloat16, seed=seed)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():                      # zero-initialised layers -> small random values (nothing multiplies by 0)
        init = {s.name: s.init for s in m._spec}
        for n, p in m.named_parameters():
            if init[n] == 'zlin' and p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif n.endswith('.bias') and 'diffusion_model' in n:      # biases / LayerNorm betas away from zero
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return m.cuda().train()


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-20))


def _ref_weights(m, n, requires_grad=False):
    """{name: fp32 leaf} of block `n`: matrices rounded to bf16 (what the kernels multiply), vectors as they are."""
    out = {}
    for k, p in m.named_parameters():
        if k.startswith(n + '.'):
            v = p.detach().float()
            v = v.reshape(v.shape[0], -1).bfloat16().float() if v.dim() > 1 else v.clone()
            out[k] = v.requires_grad_(requires_grad)
    return out


def _ref_block(m, n, x, kv, heads, W=None):
    """fp32 torch restatement on the bf16-rounded weights; x [B,S,C], kv [B,N,2C] fp32 -> dict of every stored tensor."""
    t = n + '.transformer_blocks.0'
    W = W if W is not None else _ref_weights(m, n)
    Wt = lambda k: W[k]
    Vt = lambda k: W[k]
    B, S, Cc = x.shape
    r = {}
    r['hgn'] = F.group_norm(x.permute(0, 2, 1), 32, Vt(n + '.norm.weight'), Vt(n + '.norm.bias'), 1e-6).permute(0, 2, 1)
    xg = x.view(B, S, 32, Cc // 32).permute(0, 2, 1, 3).reshape(B, 32, -1)
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    r['gn_stats'] = torch.stack([mean, (var + 1e-6).rsqrt()], -1)
    r['tok'] = r['hgn'] @ Wt(n + '.proj_in.weight').t() + Vt(n + '.proj_in.bias')

    def ln(v, name):
        mu, va = v.mean(-1), v.var(-1, unbiased=False)
        return F.layer_norm(v, (Cc,), Vt(name + '.weight'), Vt(name + '.bias'), 1e-5), \
            torch.stack([mu, (va + 1e-5).rsqrt()], -1).reshape(-1, 2)

    def attn(q, k, v):
        Sq, Sk = q.shape[1], k.shape[1]
        qh = q.view(B, Sq, heads, 32).transpose(1, 2)
        kh = k.reshape(B, Sk, heads, 32).transpose(1, 2)
        vh = v.reshape(B, Sk, heads, 32).transpose(1, 2)
        s_ = qh @ kh.transpose(-1, -2) * 32.0 ** -0.5
        return (s_.softmax(-1) @ vh).transpose(1, 2).reshape(B, Sq, Cc), torch.logsumexp(s_, -1)

    r['n1'], r['st1'] = ln(r['tok'], t + '.norm1')
    r['qkv'] = r['n1'] @ torch.cat([Wt(t + '.attn1.to_q.weight'), Wt(t + '.attn1.to_k.weight'), Wt(t + '.attn1.to_v.weight')]).t()
    r['a1'], r['lse1'] = attn(r['qkv'][..., :Cc], r['qkv'][..., Cc:2 * Cc], r['qkv'][..., 2 * Cc:])
    r['x1'] = r['a1'] @ Wt(t + '.attn1.to_out.0.weight').t() + Vt(t + '.attn1.to_out.0.bias') + r['tok']
    r['n2'], r['st2'] = ln(r['x1'], t + '.norm2')
    r['q2'] = r['n2'] @ Wt(t + '.attn2.to_q.weight').t()
    r['a2'], r['lse2'] = attn(r['q2'], kv[..., :Cc], kv[..., Cc:])
    r['x2'] = r['a2'] @ Wt(t + '.attn2.to_out.0.weight').t() + Vt(t + '.attn2.to_out.0.bias') + r['x1']
    r['n3'], r['st3'] = ln(r['x2'], t + '.norm3')
    r['h'] = r['n3'] @ Wt(t + '.ff.net.0.proj.weight').t() + Vt(t + '.ff.net.0.proj.bias')
    r['g'] = r['h'][..., :4 * Cc] * F.gelu(r['h'][..., 4 * Cc:])
    r['x3'] = r['g'] @ Wt(t + '.ff.net.2.weight').t() + Vt(t + '.ff.net.2.bias') + r['x2']
    r['out'] = r['x3'] @ Wt(n + '.proj_out.weight').t() + Vt(n + '.proj_out.bias') + x
    return r


CASES = [('input_blocks.4.1', 16, 3, 7, 64), ('output_blocks.8.1', 16, 2, 7, 32), ('input_blocks.7.1', 8, 5, 7, 64),
         ('output_blocks.5.1', 8, 4, 7, 32), ('input_blocks.5.1', 16, 2, 11, 64), ('input_blocks.8.1', 8, 3, 15, 32),
         ('output_blocks.7.1', 16, 1, 16, 32)]


def _inputs(m, name, hw, B, n_slots):
    u = m.unet()
    n = u.P + name
    heads = u.heads_of[name]
    Cc = heads * 32
    g = torch.Generator().manual_seed(11 + hw + n_slots)
    x = torch.randn(B, hw, hw, Cc, generator=g).bfloat16().cuda()
    kv = (0.7 * torch.randn(B, n_slots, 2 * Cc, generator=g)).bfloat16().cuda()
    return u, n, heads, Cc, x, kv


@pytest.mark.parametrize('name,hw,B,n_slots,rows', CASES)
def test_fused_training_forward_stores_what_backward_reads(name, hw, B, n_slots, rows):
    from slotdiffusion_amd import kern
    m = _model()
    u, n, heads, Cc, x, kv = _inputs(m, name, hw, B, n_slots)
    wb = m.KG().wb
    with torch.no_grad():
        out, sv = kern.StBlockFn.run_forward(wb, x, kv, n, heads, rows)
        again = [kern.StBlockFn.run_forward(wb, x, kv, n, heads, rows) for _ in range(4)]
    torch.cuda.synchronize()
    ref = _ref_block(m, n, x.float().view(B, hw * hw, Cc), kv.float(), heads)
    sv = dict(sv, out=out.view(B, hw * hw, Cc))
    # bars: one bf16 rounding per stored tensor on top of its inputs' (rel-L2); statistics are fp32
    bar = dict(hgn=4e-3, gn_stats=1e-4, tok=5e-3, n1=6e-3, st1=2e-3, qkv=8e-3, a1=1e-2, lse1=2e-3, x1=8e-3, n2=8e-3, st2=2e-3,
               q2=1e-2, a2=1.2e-2, lse2=2e-3, x2=8e-3, n3=8e-3, st3=2e-3, h=1e-2, g=1.5e-2, x3=1e-2, out=1e-2)
    errs = {k: _rel(sv[k].view(ref[k].shape), ref[k]) for k in bar}
    print(f'{name} C={Cc} S={hw * hw} B={B} slots={n_slots} rows={rows}: ' + ' '.join(f'{k}={v:.1e}' for k, v in errs.items()))
    for k, v in errs.items():
        assert torch.isfinite(sv[k].float()).all(), k
        assert v < bar[k], (k, v)
    for o2, sv2 in again:
        assert torch.equal(o2, out), 'fused training forward is not repeatable run to run'
        assert all(torch.equal(sv2[k], sv[k]) for k in sv2), 'stored tensors differ between runs'


@pytest.mark.parametrize('name,hw,B,n_slots,rows', CASES[:5])
def test_fused_training_block_gradients_match_per_layer_launches(name, hw, B, n_slots, rows):
    """Block output, input / slot gradients and every parameter gradient of the block: fused forward (+ the backward it
    feeds) against the per-layer launches, same weights and inputs."""
    from slotdiffusion_amd import kern
    m = _model(seed=5)
    u, n, heads, Cc, x, kv = _inputs(m, name, hw, B, n_slots)
    g = torch.Generator().manual_seed(5)
    dout = torch.randn(x.shape, generator=g).bfloat16().cuda()
    KG = m.KG()
    ga = m.grad_arena()
    lo = min(m._offsets[k][0] for k in m._offsets if k.startswith(n + '.'))
    hi = max(sum(m._offsets[k]) for k in m._offsets if k.startswith(n + '.'))

    def run(fused):
        old = kern._ST_TRAIN, kern._ST_TRAIN_MIN_WGS, kern._ST_ROWS
        kern._ST_TRAIN, kern._ST_TRAIN_MIN_WGS, kern._ST_ROWS = fused, 0, rows
        try:
            ga.zero_()
            xi, kvi = x.clone().requires_grad_(True), kv.clone().requires_grad_(True)
            out = u._st(KG, name, xi, heads, kvi)
            out.backward(dout)
            torch.cuda.synchronize()
            return out.detach().float(), xi.grad.float(), kvi.grad.float(), ga[lo:hi].clone()
        finally:
            kern._ST_TRAIN, kern._ST_TRAIN_MIN_WGS, kern._ST_ROWS = old
    calls = []
    orig = kern.StBlockFn.run_forward
    kern.StBlockFn.run_forward = staticmethod(lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    try:
        of, dxf, dkvf, gf = run(True)
    finally:
        kern.StBlockFn.run_forward = orig
    assert calls, 'the block must take the fused training path'
    op, dxp, dkvp, gp = run(False)
    e = dict(out=_rel(of, op), dx=_rel(dxf, dxp), dkv=_rel(dkvf, dkvp), params=_rel(gf, gp))
    print(f'{name} C={Cc} S={hw * hw} B={B} slots={n_slots} rows={rows}: ' + ' '.join(f'{k}={v:.2e}' for k, v in e.items()))
    assert e['out'] < 1.5e-2 and e['dx'] < 2e-2 and e['dkv'] < 2e-2 and e['params'] < 2e-2, e
    # per-tensor: no parameter of the block may be left without its gradient
    for k in m._offsets:
        if k.startswith(n + '.') and 'attn2.to_k' not in k and 'attn2.to_v' not in k:    # (kv is an input of the block here)
            o, cnt = m._offsets[k]
            a, b_ = gf[o - lo:o - lo + cnt], gp[o - lo:o - lo + cnt]
            assert float(b_.norm()) > 0 and _rel(a, b_) < 4e-2, (k, _rel(a, b_))


def test_st_pack_units_are_the_swizzled_lds_image():
    """sdmi_st_pack against a host-side gather: unit = 16 rows x 64 k, physical 16-byte chunk p of row r = logical chunk
    p ^ ((r >> 1) & 7); row-major (cs = 1) and transposed (rs = 1) sources."""
    import numpy as np
    from slotdiffusion_amd import _lib
    g = torch.Generator().manual_seed(1)
    W = torch.randn(96, 192, generator=g).bfloat16().cuda()
    units = [(16 * i, 64 * j, 0) for i in range(6) for j in range(3)] + [(16 * i, 64 * j, 1) for i in range(12) for j in range(1)]
    dst = torch.zeros(len(units) * 1024, dtype=torch.bfloat16, device='cuda')
    arr = np.zeros(len(units), dtype=np.dtype([('src', '<u8'), ('dst', '<u8'), ('rs', '<i4'), ('cs', '<i4')]))
    for i, (r0, k0, tr) in enumerate(units):
        if tr:      # unit rows walk W's columns, unit k walks W's rows
            arr[i] = (W.data_ptr() + 2 * (k0 * 192 + r0), dst.data_ptr() + 2048 * i, 1, 192)
        else:
            arr[i] = (W.data_ptr() + 2 * (r0 * 192 + k0), dst.data_ptr() + 2048 * i, 192, 1)
    tab = torch.from_numpy(arr.view(np.uint8).copy()).cuda()
    _lib.call('sdmi_st_pack', torch.cuda.current_stream().cuda_stream, descs=tab.data_ptr(), n_units=len(units))
    torch.cuda.synchronize()
    Wc, got = W.cpu(), dst.cpu().view(len(units), 16, 8, 8)
    for i, (r0, k0, tr) in enumerate(units):
        src = Wc[k0:k0 + 64, r0:r0 + 16].t() if tr else Wc[r0:r0 + 16, k0:k0 + 64]
        for r in range(16):
            for p in range(8):
                lc = p ^ ((r >> 1) & 7)
                assert torch.equal(got[i, r, p], src[r, lc * 8:lc * 8 + 8]), (i, r, p)


@pytest.mark.parametrize('name,hw,B,n_slots,rows', CASES)
def test_fused_backward_data_path_matches_the_per_layer_kernels(name, hw, B, n_slots, rows):
    """sdmi_st_train_bwd (three launches) against the per-layer backward kernels on the SAME stored tensors: every
    intermediate gradient, the input / slot gradients and the parameter gradients; repeatable run to run."""
    from slotdiffusion_amd import kern
    m = _model(seed=7)
    u, n, heads, Cc, x, kv = _inputs(m, name, hw, B, n_slots)
    g = torch.Generator().manual_seed(9)
    dout = torch.randn(x.shape, generator=g).bfloat16().cuda()
    wb = m.KG().wb
    ga = m.grad_arena()
    lo = min(m._offsets[k][0] for k in m._offsets if k.startswith(n + '.'))
    hi = max(sum(m._offsets[k]) for k in m._offsets if k.startswith(n + '.'))
    with torch.no_grad():
        out, sv = kern.StBlockFn.run_forward(wb, x, kv, n, heads, rows)

    def run(fused):
        ga.zero_()
        kern.StBlockFn.capture = {}
        try:
            with torch.no_grad():
                wb._join_queued = True            # (outside an autograd pass: the join below is called by hand)
                if fused:
                    dx, dkv = kern.StBlockFn.backward_fused(wb, n, heads, x, kv, sv, dout, rows)
                else:
                    dx, dkv = kern.StBlockFn.backward_layers(wb, n, heads, x, kv, sv, dout)
                wb.join()
            torch.cuda.synchronize()
            cap = {k: v.float().clone() for k, v in kern.StBlockFn.capture.items()}
        finally:
            kern.StBlockFn.capture = None
        return dict(cap, dx=dx.float().clone(), dkv=dkv.float().clone(), params=ga[lo:hi].clone())
    f, p = run(True), run(False)
    errs = {k: _rel(f[k], p[k]) for k in p}
    print(f'{name} C={Cc} S={hw * hw} B={B} slots={n_slots} rows={rows}: ' + ' '.join(f'{k}={v:.1e}' for k, v in errs.items()))
    for k, v in errs.items():
        assert torch.isfinite(f[k]).all(), k
        assert v < 1.5e-2, (k, v)
    for k in m._offsets:
        if k.startswith(n + '.') and 'attn2.to_k' not in k and 'attn2.to_v' not in k:
            o, cnt = m._offsets[k]
            a, b_ = f['params'][o - lo:o - lo + cnt], p['params'][o - lo:o - lo + cnt]
            assert float(b_.norm()) > 0 and _rel(a, b_) < 3e-2, (k, _rel(a, b_))
    bad = {}
    for _ in range(3):
        f2 = run(True)
        for k in f:
            if not torch.equal(f[k], f2[k]):
                bad[k] = max(bad.get(k, 0.0), float((f[k] - f2[k]).abs().max()))
    p2 = run(False)
    badp = {k: float((p[k] - p2[k]).abs().max()) for k in p if not torch.equal(p[k], p2[k])}
    assert not bad, f'fused backward is not repeatable run to run: {bad} (per-layer form: {badp})'


@pytest.mark.parametrize('name,hw,B,n_slots,rows', [('input_blocks.4.1', 16, 2, 7, 64), ('output_blocks.5.1', 8, 4, 11, 32),
                                                    ('input_blocks.8.1', 8, 3, 15, 32), ('output_blocks.8.1', 16, 2, 15, 64)])
def test_fused_training_block_gradients_match_torch_autograd(name, hw, B, n_slots, rows):
    """Fused forward + fused backward against torch autograd through the fp32 restatement of the reference block
    (attention.py:297-308, 247-251, 182-206, 44-65) on the bf16-rounded weights: output, input / slot gradients and every
    parameter gradient, bf16 bar 2e-2 (rel-L2)."""
    from slotdiffusion_amd import kern
    m = _model(seed=11)
    u, n, heads, Cc, x, kv = _inputs(m, name, hw, B, n_slots)
    g = torch.Generator().manual_seed(13)
    dout = torch.randn(x.shape, generator=g).bfloat16().cuda()
    S = hw * hw
    # reference (CPU, fp32)
    W = _ref_weights(m, n, requires_grad=True)
    W = {k: v.cpu().detach().requires_grad_(True) for k, v in W.items()}
    xr = x.float().cpu().view(B, S, Cc).requires_grad_(True)
    kvr = kv.float().cpu().requires_grad_(True)
    ref = _ref_block(m, n, xr, kvr, heads, W)['out']
    ref.backward(dout.float().cpu().view(B, S, Cc))
    # fused path
    old = kern._ST_TRAIN, kern._ST_TRAIN_MIN_WGS, kern._ST_ROWS
    kern._ST_TRAIN, kern._ST_TRAIN_MIN_WGS, kern._ST_ROWS = True, 0, rows
    try:
        ga = m.grad_arena()
        ga.zero_()
        xi, kvi = x.clone().requires_grad_(True), kv.clone().requires_grad_(True)
        out = u._st(m.KG(), name, xi, heads, kvi)
        assert isinstance(out.grad_fn, kern.StBlockFn._backward_cls), 'the block must take the fused training path'
        out.backward(dout)
        torch.cuda.synchronize()
    finally:
        kern._ST_TRAIN, kern._ST_TRAIN_MIN_WGS, kern._ST_ROWS = old
    errs = dict(out=_rel(out.detach().view(B, S, Cc), ref.detach()), dx=_rel(xi.grad.view(B, S, Cc), xr.grad),
                dkv=_rel(kvi.grad, kvr.grad))
    for k, p_ in m.named_parameters():
        if k.startswith(n + '.') and 'attn2.to_k' not in k and 'attn2.to_v' not in k:
            errs[k[len(n) + 1:]] = _rel(p_.grad.reshape(W[k].shape), W[k].grad)
    print(f'{name} C={Cc} S={S} B={B} slots={n_slots} rows={rows}: ' + ' '.join(f'{k}={v:.1e}' for k, v in errs.items()))
    assert errs['out'] < 1e-2
    worst = max(errs, key=errs.get)
    assert errs[worst] < 2e-2, (worst, errs[worst])
