"""Fused SpatialTransformer block (sdmi.h: sdmi_st_block; csrc/st_fused.hip) -- GPU parity tests through the C ABI.

The block is checked against the oracle's restatement of the reference module (oracle.slotdiff_oracle.
_spatial_transformer: attention.py:297-308, 247-251) in fp32 on the CPU, against the per-layer HIP launches it
replaces, and for run-to-run repeatability (its eight waves run free inside a GEMM phase: a data race would show as
bits that change between runs)."""
import pytest
import torch

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
    return m.cuda().eval()


def _rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())


@pytest.mark.parametrize('rows', [64, 32])
@pytest.mark.parametrize('name,hw,B,n_slots', [('input_blocks.4.1', 16, 3, 7), ('output_blocks.8.1', 16, 2, 7),
                                               ('input_blocks.7.1', 8, 5, 7), ('output_blocks.5.1', 8, 2, 7),
                                               # the video configurations' slot counts (savi_diffusion.py:143-144):
                                               # 16-column score groups, C = 256 blocks
                                               ('input_blocks.4.1', 16, 3, 11), ('output_blocks.8.1', 16, 2, 15),
                                               ('input_blocks.5.1', 16, 2, 16), ('input_blocks.4.1', 16, 2, 8)])
def test_fused_block_matches_oracle_and_per_layer_launches(name, hw, B, n_slots, rows):
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd import kern
    m = _model()
    kern._ST_MIN_WGS = 0                 # (the production gate keeps small grids on the per-layer launches)
    kern._ST_ROWS = rows                 # token rows per workgroup: both instantiations
    K, u = m.K(), m.unet()
    n = u.P + name
    heads = u.heads_of[name]
    Cc = heads * 32
    g = torch.Generator().manual_seed(11 + hw)
    x = torch.randn(B, hw, hw, Cc, generator=g).bfloat16()
    slots = torch.randn(B, n_slots, 192, generator=g).bfloat16()
    W = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    ref = O._spatial_transformer(W, n, x.float().permute(0, 3, 1, 2), slots.float(), heads).permute(0, 2, 3, 1)
    with torch.no_grad():
        xd, ctx = x.cuda(), slots.cuda()
        t = n + '.transformer_blocks.0'
        kv = K.linear_multi(ctx, [(t + '.attn2.to_k.weight', t + '.attn2.to_v.weight')])[0]
        kvp = {'kv': kv, 'fold': K.cross_prepare(kv, t, heads)}
        assert kvp['fold'] is not None and 'st_img' in kvp['fold']
        fused = K.st_fused(xd, n, heads, kvp)
        assert fused is not None, 'the block must qualify for the fused path'
        old = kern._ST_FUSED
        kern._ST_FUSED = False
        try:
            per_layer = u._st(K, name, xd, heads, kvp)
        finally:
            kern._ST_FUSED = old
        again = [K.st_fused(xd, n, heads, kvp) for _ in range(8)]
    torch.cuda.synchronize()
    e_f, e_p = _rel(fused, ref), _rel(per_layer, ref)
    print(f'{name} C={Cc} S={hw * hw} B={B} slots={n_slots}: fused vs oracle {e_f:.3e}, per-layer launches vs oracle {e_p:.3e}')
    assert torch.isfinite(fused.float()).all()
    assert e_f < 1.5e-2                                   # bf16 bar of the kernel tests (rel-L2)
    assert e_f < 2.0 * e_p + 2e-3                          # and no worse than the launches it replaces
    assert all(torch.equal(fused, a) for a in again), 'fused block is not repeatable run to run'
    kern._ST_ROWS = 0


def test_fused_block_engages_in_the_sampler_and_keeps_eps():
    """UNet eps with the fused blocks against the per-layer launches on the same inputs (the sampler fixtures of
    test_gpu_model.py cover the path end to end against the reference)."""
    from slotdiffusion_amd import kern, ops
    kern._ST_MIN_WGS, kern._ST_ROWS = 0, 0
    m = _model(seed=5)
    B = 2
    g = torch.Generator().manual_seed(2)
    x_t = ops.nchw_to_nhwc(torch.randn(B, 3, 32, 32, generator=g).cuda(), torch.float32, 4)
    t = torch.tensor([250.0, 731.0]).cuda()
    slots = torch.randn(B, 7, 192, generator=g).cuda()
    calls = []
    orig = kern.Kern.st_fused

    def spy(self, *a, **k):
        r = orig(self, *a, **k)
        calls.append(r is not None)
        return r
    kern.Kern.st_fused = spy
    try:
        with torch.no_grad():
            e_fused = m._unet_eps(x_t, t, slots).float().clone()
    finally:
        kern.Kern.st_fused = orig
    old = kern._ST_FUSED
    kern._ST_FUSED = False
    try:
        with torch.no_grad():
            e_ref = m._unet_eps(x_t, t, slots).float().clone()
    finally:
        kern._ST_FUSED = old
    assert sum(calls) == 10 and len(calls) == 16          # the 16^2 and 8^2 levels; the six 4^2 blocks keep their launches
    assert _rel(e_fused, e_ref) < 1.5e-2


@pytest.mark.parametrize('n_slots', [11, 15])
@pytest.mark.parametrize('name,hw', [('input_blocks.7.1', 8), ('input_blocks.10.1', 4), ('middle_block.1', 4),
                                     ('output_blocks.8.1', 16)])
def test_folded_cross_attention_with_sixteen_wide_groups(name, hw, n_slots):
    """11 / 15 slots (video configurations) on the folded slot cross-attention outside the fused block: the per-image
    score GEMM with the 16-column softmax epilogue + the output GEMM against the explicit q-projection / attention /
    output-projection launches and the oracle, at the levels the fused block does not serve with these slot counts."""
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd import kern
    m = _model()
    K, u = m.K(), m.unet()
    n = u.P + name
    heads = u.heads_of[name]
    Cc = heads * 32
    B = 3
    g = torch.Generator().manual_seed(23 + hw + n_slots)
    x = torch.randn(B, hw, hw, Cc, generator=g).bfloat16()
    slots = torch.randn(B, n_slots, 192, generator=g).bfloat16()
    W = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    ref = O._spatial_transformer(W, n, x.float().permute(0, 3, 1, 2), slots.float(), heads).permute(0, 2, 3, 1)
    old = kern._ST_FUSED
    kern._ST_FUSED = False
    try:
        with torch.no_grad():
            xd, ctx = x.cuda(), slots.cuda()
            t = n + '.transformer_blocks.0'
            kv = K.linear_multi(ctx, [(t + '.attn2.to_k.weight', t + '.attn2.to_v.weight')])[0]
            fold = K.cross_prepare(kv, t, heads)
            assert fold is not None and fold['wq'].shape[1] == heads * 16
            folded = u._st(K, name, xd, heads, {'kv': kv, 'fold': fold})
            explicit = u._st(K, name, xd, heads, {'kv': kv, 'fold': None})
    finally:
        kern._ST_FUSED = old
    torch.cuda.synchronize()
    e_f, e_x = _rel(folded, ref), _rel(explicit, ref)
    assert torch.isfinite(folded.float()).all()
    assert e_f < 1.5e-2 and e_f < 2.0 * e_x + 2e-3, (e_f, e_x)


@pytest.mark.parametrize('name,hw,B', [('input_blocks.7.1', 8, 64), ('output_blocks.5.1', 8, 20), ('input_blocks.4.1', 16, 8)])
def test_fused_block_feed_forward_split_over_workgroup_pairs(name, hw, B):
    """sdmi.h: ff_split = 2 -- every 32 token rows served by a pair of workgroups that each stream half of the hidden
    chunks; the pair's fp32 partials are finished (+ bias + x) by the kernel behind: the GroupNorm's prologue (part
    source) or sdmi_splitk_finish.  Both finishes against the unsplit launch and the oracle; repeatable."""
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd import kern, ops
    m = _model()
    kern._ST_MIN_WGS, kern._ST_ROWS = 0, 32
    K, u = m.K(), m.unet()
    n = u.P + name
    heads = u.heads_of[name]
    Cc = heads * 32
    g = torch.Generator().manual_seed(31 + hw + B)
    x = torch.randn(B, hw, hw, Cc, generator=g).bfloat16().cuda()
    slots = torch.randn(B, 7, 192, generator=g).bfloat16().cuda()
    gam, bet = (1 + 0.1 * torch.randn(Cc, generator=g)).cuda(), (0.1 * torch.randn(Cc, generator=g)).cuda()
    try:
        with torch.no_grad():
            t = n + '.transformer_blocks.0'
            kv = K.linear_multi(slots, [(t + '.attn2.to_k.weight', t + '.attn2.to_v.weight')])[0]
            kvp = {'kv': kv, 'fold': K.cross_prepare(kv, t, heads)}
            plain = K.st_fused(x, n, heads, kvp)                     # outside the deferring context: one workgroup per 32 rows
            y_plain = ops.group_norm(plain, gam, bet, eps=1e-5, act='silu')
            outs = []
            for finish in ('flush', 'groupnorm', 'flush'):
                with ops.defer_splitk():
                    o = K.st_fused(x, n, heads, kvp)
                    assert (B * hw * hw // 32 <= 128) == (o.data_ptr() in ops._PENDING)
                    if finish == 'groupnorm':
                        y = ops.group_norm(o, gam, bet, eps=1e-5, act='silu')     # finishes the pair sums in its prologue
                        assert o.data_ptr() not in ops._PENDING
                outs.append(o.clone())
        torch.cuda.synchronize()
    finally:
        kern._ST_MIN_WGS, kern._ST_ROWS = 128, 0
    if B * hw * hw // 32 > 128:
        assert all(torch.equal(o, plain) for o in outs)      # (a grid that fills the chip keeps the single launch)
        return
    W = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
    ref = O._spatial_transformer(W, n, x.float().cpu().permute(0, 3, 1, 2), slots.float().cpu(), heads).permute(0, 2, 3, 1)
    e_s, e_p = _rel(outs[0], ref), _rel(plain, ref)
    print(f'{name} B={B}: split vs oracle {e_s:.3e}, unsplit {e_p:.3e}, split vs unsplit {_rel(outs[0], plain):.3e}')
    assert e_s < 1.5e-2 and e_s < 1.5 * e_p + 1e-3
    assert _rel(outs[0], plain) < 3e-3
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])     # both finishes, run to run
    assert _rel(y, y_plain) < 4e-3
