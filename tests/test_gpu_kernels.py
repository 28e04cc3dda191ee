"""Per-kernel parity on a real MI355X: every libsdmi kernel (called through the C ABI via
slotdiffusion_amd.ops) against the fp32 torch-CPU restatement of the same op.

Tolerances: fp32 path  -> |err| <= 2e-5 * scale (+ index outputs bit-exact);
            bf16 path  -> relative L2 error <= 1.5e-2 (bf16 has 8 mantissa bits)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _ops():
    from slotdiffusion_amd import ops
    return ops


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def check(out, ref, dtype, what=''):
    out = out.float().cpu()
    if dtype == torch.float32:
        scale = float(ref.abs().max()) + 1e-6
        err = float((out - ref).abs().max())
        assert err <= 2e-5 * scale + 1e-6, f'{what}: fp32 max err {err:.3e} (scale {scale:.3e})'
    else:
        e = rel_l2(out, ref)
        assert e <= 1.5e-2, f'{what}: bf16 rel-L2 {e:.3e}'


def q(t, dtype):
    """Quantise a CPU fp32 tensor to the compute dtype's grid (so both sides see equal inputs)."""
    return t.to(dtype).float()


def nhwc(t, dtype, cpad=None):
    t = t.permute(0, 2, 3, 1).contiguous()
    if cpad and cpad != t.shape[-1]:
        t = F.pad(t, (0, cpad - t.shape[-1]))
    return t.to(dtype).to(DEV)


def pack_w(w, dtype, cpad=None):
    """[Cout,Cin,kh,kw] -> [Cout, kh*kw*Cpad] K-contiguous."""
    w = w.permute(0, 2, 3, 1).contiguous()
    if cpad and cpad != w.shape[-1]:
        w = F.pad(w, (0, cpad - w.shape[-1]))
    return w.reshape(w.shape[0], -1).to(dtype).to(DEV)


CONV_CASES = [
    # B, Cin, Cout, H, k, stride, pad(t,b,l,r), ups, extras
    (2, 16, 32, 12, 3, 1, (1, 1, 1, 1), False, ''),
    (3, 64, 128, 16, 3, 1, (1, 1, 1, 1), False, 'bias,rowvec,res,silu'),
    (2, 128, 128, 32, 3, 1, (1, 1, 1, 1), False, 'bias'),            # 128x128 tiles
    (2, 32, 48, 16, 3, 2, (1, 1, 1, 1), False, 'bias'),              # UNet Downsample
    (2, 32, 40, 16, 3, 2, (0, 1, 0, 1), False, 'bias'),              # VQ-VAE asymmetric pad
    (2, 32, 64, 8, 3, 1, (1, 1, 1, 1), True, 'bias'),                # nearest x2 folded in
    (2, 3, 64, 16, 3, 1, (1, 1, 1, 1), False, 'bias'),               # Cin=3 (channel padded)
    (2, 128, 3, 16, 3, 1, (1, 1, 1, 1), False, 'bias'),              # Cout=3
    (2, 64, 96, 8, 1, 1, (0, 0, 0, 0), False, 'bias,res'),           # 1x1
    (2, 64, 128, 16, 1, 2, (0, 0, 0, 0), False, ''),                 # ResNet downsample 1x1 s2
    (4, 512, 512, 4, 3, 1, (1, 1, 1, 1), False, 'bias,rowvec'),      # skinny: split-K path
    (1, 896, 384, 8, 3, 1, (1, 1, 1, 1), False, 'bias'),             # non power-of-two Cin
    (1, 960, 64, 4, 3, 1, (1, 1, 1, 1), False, 'bias'),              # 16 K-splits of 135 K tiles: the last one is empty
    # LDS-DMA kernels (>= 192 tiles of 256x128, or of 128x128 when M is too small for that)
    (48, 64, 128, 32, 3, 1, (1, 1, 1, 1), False, 'bias,rowvec,res,silu'),   # 256x128, 3 stages
    (25, 128, 192, 30, 3, 1, (1, 1, 1, 1), False, 'bias'),           # 128x128 x4 stages, ragged M / N
    (48, 64, 128, 32, 5, 1, (2, 2, 2, 2), False, 'bias'),            # 5x5 taps
    (48, 64, 136, 64, 3, 2, (1, 1, 1, 1), False, 'bias,res'),        # stride 2, ragged N
    (48, 328, 128, 32, 1, 1, (0, 0, 0, 0), False, 'bias,res'),       # 1x1 with a K tail
    # direct 3x3 kernel (64 -> 64 channels at full resolution, >= 512 tiles of 4 x 64 pixels)
    (8, 64, 64, 128, 3, 1, (1, 1, 1, 1), False, 'bias,res'),
    (8, 64, 48, 128, 3, 1, (1, 1, 1, 1), False, 'bias,rowvec,silu'),     # ragged N
    (16, 64, 64, 64, 3, 1, (1, 1, 1, 1), False, ''),
]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', CONV_CASES)
def test_igemm_conv(case, dtype):
    ops = _ops()
    B, Cin, Cout, H, k, stride, pad, ups, extras = case
    g = torch.Generator().manual_seed(hash(case) % 10000)
    x = q(torch.randn(B, Cin, H, H, generator=g), dtype)
    w = q(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k), dtype)
    bias = torch.randn(Cout, generator=g) if 'bias' in extras else None
    xin = F.interpolate(x, scale_factor=2, mode='nearest') if ups else x
    ref = F.conv2d(F.pad(xin, (pad[2], pad[3], pad[0], pad[1])), w, bias, stride=stride)
    Ho = ref.shape[2]
    rowvec = torch.randn(B, Cout, generator=g) if 'rowvec' in extras else None
    res = q(torch.randn(B, Cout, Ho, Ho, generator=g), dtype) if 'res' in extras else None
    if rowvec is not None:
        ref = ref + rowvec[:, :, None, None]
    if res is not None:
        ref = ref + res
    if 'silu' in extras:
        ref = F.silu(ref)
    vec = ops.vec_of(dtype)
    cpad = (Cin + vec - 1) // vec * vec
    out = ops.conv2d(nhwc(x, dtype, cpad), pack_w(w, dtype, cpad),
                     bias.to(DEV) if bias is not None else None, kh=k, kw=k, stride=stride,
                     pad=pad, ups=ups, rowvec=rowvec.to(DEV) if rowvec is not None else None,
                     residual=nhwc(res, dtype) if res is not None else None,
                     act='silu' if 'silu' in extras else None)
    assert out.shape == (B, Ho, Ho, Cout)
    check(out.permute(0, 3, 1, 2), ref, dtype, f'conv {case}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('mnk', [(70, 50, 24), (448, 512, 192), (1024, 4096, 512), (64, 1536, 512),
                                 (4, 2944, 512), (8192, 192, 256), (50000, 130, 264)])
def test_igemm_linear(mnk, dtype):
    ops = _ops()
    M, N, K = mnk
    g = torch.Generator().manual_seed(M + N + K)
    x = q(torch.randn(M, K, generator=g), dtype)
    w = q(torch.randn(N, K, generator=g) / math.sqrt(K), dtype)
    b = torch.randn(N, generator=g)
    r = q(torch.randn(M, N, generator=g), dtype)
    ref = F.gelu(F.linear(x, w, b) + r)
    out = ops.linear(x.to(dtype).to(DEV), w.to(dtype).to(DEV), b.to(DEV),
                     residual=r.to(dtype).to(DEV), act='gelu')
    check(out, ref, dtype, f'linear {mnk}')
    # fp32 output from a low-precision GEMM + strided A view
    xx = torch.cat([x, x], 1).to(dtype).to(DEV)
    out2 = ops.linear(xx[:, K:], w.to(dtype).to(DEV), b.to(DEV), out_dtype=torch.float32)
    check(out2, F.linear(x, w, b), dtype, 'linear strided/f32 out')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('geglu', [False, True])
@pytest.mark.parametrize('mnk', [(1024, 1536, 512), (4096, 384, 384), (16384, 256, 256), (448, 64, 192),
                                 (70, 200, 136), (1000, 96, 264)])
def test_igemm_layernorm_fold_and_geglu(mnk, geglu, dtype):
    """Register-resident GEMM with the fused epilogues (sdmi.h: ln_colsum / geglu):
    linear(LayerNorm(x)) [-> value * gelu(gate)] in one launch, the norm folded into pre-scaled
    weights + a per-row (mean, rstd) correction, against torch's LayerNorm -> Linear -> GEGLU."""
    ops = _ops()
    M, N, K = mnk
    g = torch.Generator().manual_seed(M + N + K)
    x = q(torch.randn(M, K, generator=g) * 1.5 + 0.7 * torch.randn(M, 1, generator=g), dtype)
    NW = 2 * N if geglu else N
    w = torch.randn(NW, K, generator=g) / math.sqrt(K)
    b = torch.randn(NW, generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g)
    eps = 1e-5
    wp = q(w * gamma, dtype)                                # the operand the kernel multiplies with
    # reference: exact LayerNorm statistics, the same rounded operand (the fold is an identity then)
    mu, var = x.double().mean(1, keepdim=True), x.double().var(1, unbiased=False, keepdim=True)
    n = (x.double() - mu) / (var + eps).sqrt()
    ref = (n @ wp.double().t() + (w.double() @ beta.double() + b.double())).float()
    if geglu:
        ref = ref[:, :N] * F.gelu(ref[:, N:])
    out = ops.linear(x.to(dtype).to(DEV), wp.to(dtype).to(DEV), (w @ beta + b).to(DEV),
                     ln_colsum=wp.sum(1).to(DEV), ln_eps=eps, geglu=geglu)
    assert out.shape == (M, N)
    check(out, ref, dtype, f'ln-fold {mnk} geglu={geglu}')
    if not geglu:        # GEGLU epilogue alone (no norm), relu epilogue with the fold
        w2 = q(torch.randn(2 * N, K, generator=g) / math.sqrt(K), dtype)
        b2 = torch.randn(2 * N, generator=g)
        h = F.linear(x, w2, b2)
        out2 = ops.linear(x.to(dtype).to(DEV), w2.to(dtype).to(DEV), b2.to(DEV), geglu=True)
        check(out2, h[:, :N] * F.gelu(h[:, N:]), dtype, f'geglu {mnk}')
        # ... keeping the pre-activation (training): h as the plain linear gives it, y from the ROUNDED h
        hk = torch.full((M, 2 * N + 8), 3.0, dtype=dtype, device=DEV)
        y2 = ops.linear(x.to(dtype).to(DEV), w2.to(dtype).to(DEV), b2.to(DEV), geglu=True, out2=hk[:, :2 * N])
        plain = ops.linear(x.to(dtype).to(DEV), w2.to(dtype).to(DEV), b2.to(DEV))
        assert float((hk[:, :2 * N].float() - plain.float()).abs().max()) <= (0 if dtype == torch.float32 else 2e-2)
        assert bool((hk[:, 2 * N:] == 3.0).all())
        hr = hk[:, :2 * N].float().cpu()
        assert float((y2.float().cpu() - hr[:, :N] * F.gelu(hr[:, N:])).abs().max()) <= \
            (1e-5 if dtype == torch.float32 else 2e-2 * float(y2.float().abs().max()))
        out3 = ops.linear(x.to(dtype).to(DEV), wp.to(dtype).to(DEV), (w @ beta + b).to(DEV),
                          ln_colsum=wp.sum(1).to(DEV), ln_eps=eps, act='relu', out_dtype=torch.float32)
        check(out3, F.relu(ref), dtype, f'ln-fold relu {mnk}')


def _e4m3(t):
    """fp32 -> nearest e4m3fn value (saturating at +-448), via torch's own float8 type on the CPU."""
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()


@pytest.mark.parametrize('case', [(2, 16, 16, 64, 96, 3, 1), (2, 28, 28, 128, 128, 3, 1), (1, 14, 14, 256, 320, 3, 2),
                                  (64, 8, 8, 128, 64, 1, 1), (3, 9, 11, 80, 72, 3, 1)])
def test_igemm_fp8_operands(case):
    """SDMI_FP8: e4m3fn operands (sdmi_quant_fp8) through v_mfma_scale_f32_32x32x64_f8f6f4.  The
    quantiser must equal torch's e4m3fn rounding bit for bit; the convolution must equal the fp32
    convolution of the DEQUANTISED operands (fp32 accumulation: only summation order differs)."""
    ops = _ops()
    B, H, W, Cin, Cout, k, stride = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Cin, H, W, generator=g) * 1.3
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    sx, sw = 8.0, 448.0 / float(w.abs().max())
    xq = ops.quant_fp8(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV), sx)
    wq = ops.quant_fp8(w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV), sw)
    x_ref = _e4m3(x.to(torch.bfloat16).float() * sx)
    w_ref = _e4m3(w * sw)
    assert torch.equal(xq.cpu().view(torch.float8_e4m3fn).float(), x_ref.permute(0, 2, 3, 1))
    assert torch.equal(wq.cpu().view(torch.float8_e4m3fn).float().view(Cout, k, k, Cin), w_ref.permute(0, 2, 3, 1))
    pad = k // 2
    ref = F.conv2d(x_ref, w_ref, None, stride=stride, padding=pad) / (sx * sw) + b.view(1, -1, 1, 1)
    out = ops.conv2d(xq, wq, b.to(DEV), kh=k, kw=k, stride=stride, pad=(pad,) * 4, out_dtype=torch.float32,
                     alpha=1.0 / (sx * sw))
    # (the f8f6f4 MFMA adds its 64 products per instruction with a bounded-width adder tree: a few
    # 1e-5 relative to the output scale, against 2e-5 for the fp32-accumulating bf16 / fp32 paths)
    o, scale = out.permute(0, 3, 1, 2).float().cpu(), float(ref.abs().max())
    assert float((o - ref).abs().max()) <= 1e-4 * scale, f'fp8 conv {case}'
    # and it stays close to the unquantised convolution (e4m3: 3 mantissa bits per operand)
    full = F.conv2d(x, w, b, stride=stride, padding=pad)
    assert rel_l2(out.permute(0, 3, 1, 2), full) <= 6e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', [(2, 16, 16, 256, 128), (3, 8, 8, 384, 384), (2, 32, 32, 128, 128), (1, 64, 64, 192, 64),
                                  (64, 4, 4, 512, 512)])
def test_two_source_groupnorm_and_1x1_conv(case, dtype):
    """The UNet's skip concat read in place (sdmi.h: x2 / a2): GroupNorm(+SiLU) and the 1x1 skip
    convolution over [a | b] without materialising the concatenation -- identical bits to the same
    kernels on the concatenated tensor."""
    ops = _ops()
    B, H, W, Ca, Cb = case
    g = torch.Generator().manual_seed(sum(case))
    a = q(torch.randn(B, H, W, Ca, generator=g), dtype).to(dtype).to(DEV)
    b = q(torch.randn(B, H, W, Cb, generator=g) * 1.7 + 0.3, dtype).to(dtype).to(DEV)
    cat = ops.concat_channels(a, b)
    C = Ca + Cb
    gamma, beta = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV), (0.1 * torch.randn(C, generator=g)).to(DEV)
    y_ref = ops.group_norm(cat, gamma, beta, eps=1e-5, act='silu')
    y_two = ops.group_norm(a, gamma, beta, eps=1e-5, act='silu', x2=b)
    assert y_two.shape == cat.shape and torch.equal(y_ref, y_two)
    N = 192
    w = q(torch.randn(N, C, generator=g) / math.sqrt(C), dtype).to(dtype).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    o_ref = ops.conv2d(cat, w, bias, kh=1, kw=1, pad=(0, 0, 0, 0))
    o_two = ops.conv2d(a, w, bias, kh=1, kw=1, pad=(0, 0, 0, 0), x2=b)
    assert torch.equal(o_ref, o_two)
    check(o_two, F.linear(cat.float().cpu(), w.float().cpu(), bias.cpu()), dtype, f'two-source 1x1 {case}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_bmm_nt_and_softmax(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    Z, S, C = 2, 160, 64
    qk = q(torch.randn(Z, S, 2 * C, generator=g), dtype)
    sc = torch.empty(Z, S, S, dtype=dtype, device=DEV)
    qkd = qk.to(dtype).to(DEV)
    ops.bmm_nt(qkd[..., :C], qkd[..., C:], sc)
    ref = torch.bmm(qk[..., :C], qk[..., C:].transpose(1, 2))
    check(sc, ref, dtype, 'bmm_nt')
    ops.softmax_rows_(sc, scale=C ** -0.5)
    refs = F.softmax(q(ref, dtype) * C ** -0.5, dim=-1)
    if dtype == torch.float32:
        check(sc, refs, dtype, 'softmax rows')
    else:
        assert rel_l2(sc, refs) < 3e-2
    # weight-as-row-operand form with per-row bias (V^T in the VQ-VAE attention)
    wv = q(torch.randn(C, C, generator=g) / 8, dtype)
    h = q(torch.randn(Z, S, C, generator=g), dtype)
    bv = torch.randn(C, generator=g)
    vt = torch.empty(Z, C, S, dtype=dtype, device=DEV)
    ops.bmm_nt(wv.to(dtype).to(DEV).unsqueeze(0).expand(Z, C, C), h.to(dtype).to(DEV), vt,
               bias_m=bv.to(DEV))
    check(vt, (F.linear(h, wv, bv)).transpose(1, 2), dtype, 'bmm_nt weight-row')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(2, 64, 16, 'silu', 1e-5), (3, 128, 32, 'silu', 1e-5),
                                   (2, 896, 8, 'silu', 1e-5), (2, 384, 8, None, 1e-6),
                                   (2, 64, 64, 'relu', 1e-5), (1, 1024, 4, 'silu', 1e-5),
                                   # group sizes 8 / 16 / 20 / 24 channels (a 16-byte vector inside one group,
                                   # across two groups at odd offsets) on the two-slot single-pass kernel
                                   (3, 256, 16, 'silu', 1e-5), (2, 512, 8, 'silu', 1e-5),
                                   (2, 640, 16, 'silu', 1e-5), (2, 768, 8, 'silu', 1e-5),
                                   (2, 224, 16, 'silu', 1e-5), (2, 160, 8, 'silu', 1e-5)])
def test_groupnorm(shape, dtype):
    ops = _ops()
    B, C, H, act, eps = shape
    g = torch.Generator().manual_seed(C + H)
    x = q(torch.randn(B, C, H, H, generator=g) * 2 + 0.7, dtype)
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ref = F.group_norm(x, 32, gam, bet, eps)
    ref = {'silu': F.silu, 'relu': F.relu, None: lambda v: v}[act](ref)
    out = ops.group_norm(nhwc(x, dtype), gam.to(DEV), bet.to(DEV), eps=eps, act=act)
    check(out.permute(0, 3, 1, 2), ref, dtype, f'groupnorm {shape}')
    # the saved statistics (what the backward reads): every group written exactly once
    _, st = ops.group_norm(nhwc(x, dtype), gam.to(DEV), bet.to(DEV), eps=eps, act=act, return_stats=True)
    xg = x.view(B, 32, -1).double()
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    assert torch.allclose(st[..., 0].cpu().double(), mean, atol=1e-5, rtol=1e-5)
    assert torch.allclose(st[..., 1].cpu().double(), (var + eps).rsqrt(), atol=1e-5, rtol=1e-5)
    # fused residual + relu (ResNet BasicBlock tail)
    r = q(torch.randn(B, C, H, H, generator=g), dtype)
    out = ops.group_norm(nhwc(x, dtype), gam.to(DEV), bet.to(DEV), eps=eps, act='relu',
                         residual=nhwc(r, dtype))
    check(out.permute(0, 3, 1, 2), F.relu(F.group_norm(x, 32, gam, bet, eps) + r), dtype,
          'groupnorm+res')


@pytest.mark.parametrize('kind', ['silu', 'gelu'])
def test_fast_activations_bf16_storage(kind):
    """bf16 storage takes the fast activation forms (v_rcp SiLU, branch-free erf; common.h act_apply<true>): after
    the rounding to bf16 the result must be the correctly rounded one almost everywhere and never more than one
    bf16 ulp away; fp32 storage keeps the correctly rounded forms (2e-7 of the exact value)."""
    ops = _ops()
    x = torch.cat([torch.linspace(-12, 12, 200001), torch.tensor([0.0, -0.0, 100.0, -100.0, 1.3120, -1.3120])])
    xb = x.to(torch.bfloat16)
    f = {'silu': F.silu, 'gelu': F.gelu}[kind]
    exact = f(xb.double())
    out = ops.act(xb.to(DEV), kind).cpu()
    ref = exact.to(torch.bfloat16)
    # (GELU below x = -4 is 0.5 x (1 + erf) with 1 + erf cancelling: any fp32 evaluation -- torch's included -- is
    # only good to ~1e-7 ABSOLUTE there, on values below 1e-5)
    main = xb.float() > -4.0
    same = (out == ref) | ((out == 0) & (ref == 0))
    assert float(same[main].float().mean()) >= 0.995, float(same[main].float().mean())
    ulp = (exact.abs() * 2.0 ** -7).clamp_min(2.0 ** -126)
    assert bool(((out.double() - exact).abs()[main] <= ulp[main]).all())
    assert bool(((out.double() - exact).abs()[~main] <= ulp[~main] + 2e-7).all())
    out32 = ops.act(xb.float().to(DEV), kind).cpu().double()
    assert bool(((out32 - exact).abs() <= 4e-7 * exact.abs() + 2e-7).all())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C', [192, 256, 384, 512])
def test_layernorm(C, dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(C)
    x = q(torch.randn(37, C, generator=g) * 1.5 - 0.3, dtype)
    gam, bet = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    out = ops.layer_norm(x.to(dtype).to(DEV), gam.to(DEV), bet.to(DEV))
    check(out, F.layer_norm(x, (C,), gam, bet, 1e-5), dtype, f'layernorm {C}')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cfg', [(2, 8, 256, 256), (2, 12, 64, 64), (3, 16, 16, 16), (2, 8, 256, 7),
                                 (2, 16, 16, 15), (2, 5, 784, 784), (1, 3, 200, 1000), (1, 2, 300, 1300),
                                 (1, 2, 33, 512), (1, 2, 65, 513), (1, 1, 5, 1)])
def test_attention(cfg, dtype):
    ops = _ops()
    B, heads, Sq, Skv = cfg
    if Skv > ops.ATTN_LDS_MAX_KV and dtype != torch.bfloat16:
        pytest.skip('beyond 400 keys only the bf16 matrix-core kernel keeps K/V in LDS')
    C = heads * 32
    g = torch.Generator().manual_seed(Sq * 7 + Skv)
    qq = q(torch.randn(B, Sq, C, generator=g), dtype)
    kv = q(torch.randn(B, Skv, 2 * C, generator=g), dtype)

    def split(t, S):
        return t.view(B, S, heads, 32).permute(0, 2, 1, 3)
    sim = torch.einsum('bhid,bhjd->bhij', split(qq, Sq), split(kv[..., :C].contiguous(), Skv)) * 32 ** -0.5
    ref = torch.einsum('bhij,bhjd->bhid', sim.softmax(-1), split(kv[..., C:].contiguous(), Skv))
    ref = ref.permute(0, 2, 1, 3).reshape(B, Sq, C)
    kvd = kv.to(dtype).to(DEV)
    out = ops.attention(qq.to(dtype).to(DEV), kvd[..., :C], kvd[..., C:], heads)
    check(out, ref, dtype, f'attention {cfg}')


@pytest.mark.parametrize('cfg', [(2, 6, 785, 785), (1, 3, 100, 300), (2, 2, 40, 17), (1, 2, 70, 256), (1, 1, 31, 257)])
def test_attention_head_dim_64(cfg):
    """bf16 matrix-core attention at the DINO ViT's head size (keys chunked through LDS, 256 at a time)
    against fp32 torch on the bf16-rounded inputs."""
    ops = _ops()
    B, heads, Sq, Skv = cfg
    hd, dtype = 64, torch.bfloat16
    C = heads * hd
    g = torch.Generator().manual_seed(Sq + Skv)
    qq = q(torch.randn(B, Sq, C, generator=g), dtype)
    kv = q(torch.randn(B, Skv, 2 * C, generator=g), dtype)
    sp = lambda t, S: t.view(B, S, heads, hd).permute(0, 2, 1, 3)
    sim = torch.einsum('bhid,bhjd->bhij', sp(qq, Sq), sp(kv[..., :C].contiguous(), Skv)) * hd ** -0.5
    ref = torch.einsum('bhij,bhjd->bhid', sim.softmax(-1), sp(kv[..., C:].contiguous(), Skv))
    ref = ref.permute(0, 2, 1, 3).reshape(B, Sq, C)
    kvd = kv.to(dtype).to(DEV)
    lse = torch.empty(B, heads, Sq, device=DEV)
    out = ops.attention(qq.to(dtype).to(DEV), kvd[..., :C], kvd[..., C:], heads, head_dim=hd, lse=lse)
    check(out, ref, dtype, f'attention hd64 {cfg}')
    assert float((lse.cpu() - torch.logsumexp(sim, -1)).abs().max()) < 2e-2


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cfg', [(2, 7, 192, 384, 3, 1024), (2, 15, 192, 384, 2, 256),
                                 (1, 11, 128, 256, 3, 100)])
def test_slot_attention_fused(cfg, dtype):
    from oracle import slotdiff_oracle as O
    ops = _ops()
    B, N, D, Hd, iters, M = cfg
    g = torch.Generator().manual_seed(N * D)
    r = lambda *s: torch.randn(*s, generator=g)
    name = 'sa'
    W = {f'{name}.project_q.0.weight': 1 + 0.1 * r(D), f'{name}.project_q.0.bias': 0.1 * r(D),
         f'{name}.project_q.1.weight': r(D, D) / math.sqrt(D),
         f'{name}.gru.weight_ih': r(3 * D, D) / math.sqrt(D), f'{name}.gru.weight_hh': r(3 * D, D) / math.sqrt(D),
         f'{name}.gru.bias_ih': 0.1 * r(3 * D), f'{name}.gru.bias_hh': 0.1 * r(3 * D),
         f'{name}.mlp.0.weight': 1 + 0.1 * r(D), f'{name}.mlp.0.bias': 0.1 * r(D),
         f'{name}.mlp.1.weight': r(Hd, D) / math.sqrt(D), f'{name}.mlp.1.bias': 0.1 * r(Hd),
         f'{name}.mlp.3.weight': r(D, Hd) / math.sqrt(Hd), f'{name}.mlp.3.bias': 0.1 * r(D)}
    kv = q(r(B, M, 2 * D), dtype)
    k, v = kv[..., :D], kv[..., D:]
    init = r(N, D)
    # oracle with identity input-LN / projections: feed k, v directly
    slots = init.repeat(B, 1, 1)
    seg = None
    for it in range(iters):
        prev = slots
        qv = F.linear(F.layer_norm(slots, (D,), W[f'{name}.project_q.0.weight'],
                                   W[f'{name}.project_q.0.bias']), W[f'{name}.project_q.1.weight'])
        att = F.softmax(D ** -0.5 * torch.einsum('bmc,bnc->bmn', k, qv), dim=-1)
        if it == iters - 1:
            seg = att.clone()
        att = att + 1e-6
        att = att / att.sum(1, keepdim=True)
        upd = torch.einsum('bmn,bmc->bnc', att, v)
        slots = O.gru_cell(W, f'{name}.gru', upd.reshape(B * N, D), prev.reshape(B * N, D)).view(B, N, D)
        hid = F.relu(F.linear(F.layer_norm(slots, (D,), W[f'{name}.mlp.0.weight'], W[f'{name}.mlp.0.bias']),
                              W[f'{name}.mlp.1.weight'], W[f'{name}.mlp.1.bias']))
        slots = slots + F.linear(hid, W[f'{name}.mlp.3.weight'], W[f'{name}.mlp.3.bias'])
    P = dict(lnq_g=W[f'{name}.project_q.0.weight'], lnq_b=W[f'{name}.project_q.0.bias'],
             wq=W[f'{name}.project_q.1.weight'], w_ih=W[f'{name}.gru.weight_ih'],
             w_hh=W[f'{name}.gru.weight_hh'], b_ih=W[f'{name}.gru.bias_ih'], b_hh=W[f'{name}.gru.bias_hh'],
             lnm_g=W[f'{name}.mlp.0.weight'], lnm_b=W[f'{name}.mlp.0.bias'], w1=W[f'{name}.mlp.1.weight'],
             b1=W[f'{name}.mlp.1.bias'], w2=W[f'{name}.mlp.3.weight'], b2=W[f'{name}.mlp.3.bias'])
    P = {a: b.contiguous().to(DEV) for a, b in P.items()}
    kvd = kv.to(dtype).to(DEV)
    s_out, seg_out = ops.slot_attention(kvd[..., :D], kvd[..., D:], init.to(DEV), P, iters=iters,
                                        eps=1e-6)
    # math is fp32 in both dtypes (inputs identical after quantisation) -> fp32-level tolerance
    check(s_out, slots, torch.float32, f'slots {cfg}')
    check(seg_out, seg, torch.float32, f'seg {cfg}')
    agree = (seg_out.cpu().argmax(-1) == seg.argmax(-1)).float().mean()
    assert agree == 1.0, f'mask argmax agreement {agree}'


def test_vq_nearest_bit_exact():
    from oracle import slotdiff_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    cb = torch.randn(4096, 3, generator=g) / math.sqrt(3)
    z = torch.randn(4, 3, 32, 32, generator=g) * 1.3
    W = {'p.quantize.embedding.weight': cb}
    zq_ref, idx_ref = O.vq_quantize(W, z, 'p')
    zp = F.pad(z.permute(0, 2, 3, 1), (0, 1)).contiguous().to(DEV)
    idx, zq = ops.vq_nearest(zp, cb.to(DEV))
    assert torch.equal(idx.cpu(), idx_ref), float((idx.cpu() != idx_ref).float().mean())
    assert torch.equal(zq.cpu()[..., :3], zq_ref.permute(0, 2, 3, 1))
    assert float(zq[..., 3].abs().max()) == 0.
    # adversarial near-ties: latents on code midpoints
    zt = ((cb[:2048] + cb[2048:]) * 0.5).view(2, 1024, 3).permute(0, 2, 1).reshape(2, 3, 32, 32)
    _, idx_ref = O.vq_quantize(W, zt, 'p')
    idx, _ = ops.vq_nearest(F.pad(zt.permute(0, 2, 3, 1), (0, 1)).contiguous().to(DEV), cb.to(DEV))
    assert torch.equal(idx.cpu(), idx_ref)
    # ragged codebooks (the kernel scans pairs of codes in chunks of 64: pads must never win) and exact
    # duplicates (ties -> the FIRST index, also across chunk and half-scan boundaries)
    for n in (1, 2, 5, 63, 64, 65, 130, 1000, 4095):
        cbn = cb[:n].contiguous()
        zq_ref, idx_ref = O.vq_quantize({'p.quantize.embedding.weight': cbn}, z[:1], 'p')
        idx, zq = ops.vq_nearest(zp[:1].contiguous(), cbn.to(DEV))
        assert torch.equal(idx.cpu(), idx_ref), n
        assert torch.equal(zq.cpu()[..., :3], zq_ref.permute(0, 2, 3, 1)), n
    # the latent formed on load as a linear combination (the sampler's x0): lincomb's arithmetic, no launch
    e_ = torch.randn(4, 32, 32, 4, generator=g).to(DEV)
    e_[..., 3] = float('nan')                              # the pad channel of the second operand is never read
    x0 = ops.lincomb(1.0, zp, -0.731, torch.nan_to_num(e_), div=0.682)
    i_ref, q_ref = ops.vq_nearest(x0, cb.to(DEV), scale=1.7)
    i_c, q_c = ops.vq_nearest(zp, cb.to(DEV), scale=1.7, comb=(1.0, -0.731, e_, 0.682))
    assert torch.equal(i_c, i_ref) and torch.equal(q_c, q_ref)
    i_c, _ = ops.vq_nearest(zp[:1, :8, :8].contiguous(), cb.to(DEV), scale=1.7,
                            comb=(1.0, -0.731, e_[:1, :8, :8].contiguous(), 0.682))       # (LDS-scan kernel)
    assert torch.equal(i_c, ops.vq_nearest(x0[:1, :8, :8].contiguous(), cb.to(DEV), scale=1.7)[0])
    # few latents: the LDS-scan kernel (the codes-in-registers kernel serves >= 512 latents)
    zs_ = z[:1, :, :8, :8].contiguous()
    for n in (5, 130, 4096):
        cbn = cb[:n].contiguous()
        _, idx_ref = O.vq_quantize({'p.quantize.embedding.weight': cbn}, zs_, 'p')
        idx, _ = ops.vq_nearest(F.pad(zs_.permute(0, 2, 3, 1), (0, 1)).contiguous().to(DEV), cbn.to(DEV))
        assert torch.equal(idx.cpu(), idx_ref), n
    dup = torch.cat([cb[:70], cb[:70], cb[:2000], cb[:70]])
    _, idx_ref = O.vq_quantize({'p.quantize.embedding.weight': dup}, z[:2], 'p')
    idx, _ = ops.vq_nearest(zp[:2].contiguous(), dup.to(DEV))
    assert torch.equal(idx.cpu(), idx_ref) and int(idx.max()) < 2140


@pytest.mark.parametrize('small', [False, True])
def test_vq_nearest_nonfinite_latents(small):
    """NaN / Inf latents (a diverged low-precision sampling pass) must not fault: indices stay in range and follow
    torch.argmin (first NaN distance, else first minimum) like the reference's quantizer (quantize.py:85-94)."""
    from oracle import slotdiff_oracle as O
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    cb = torch.randn(4096, 3, generator=g) / math.sqrt(3)
    cb[5, 1] = 0.0                                         # inf * 0 -> a NaN distance next to +inf ones
    hw = 8 if small else 32                                # 64 latents: LDS-scan kernel; 1024: codes in registers
    z = torch.randn(1, 3, hw, hw, generator=g)
    z[0, 0, 0, 1] = float('nan')
    z[0, 1, 0, 2] = float('inf')
    z[0, 2, 0, 3] = -float('inf')
    z[0, :, 1, 0] = float('inf')
    z[0, 0, 1, 1] = 3e38                                   # |z|^2 overflows
    _, idx_ref = O.vq_quantize({'p.quantize.embedding.weight': cb}, z, 'p')
    zp = F.pad(z.permute(0, 2, 3, 1), (0, 1)).contiguous().to(DEV)
    idx, zq = ops.vq_nearest(zp, cb.to(DEV))
    torch.cuda.synchronize()
    assert int(idx.min()) >= 0 and int(idx.max()) < 4096
    assert torch.equal(idx.cpu(), idx_ref), (idx.cpu() != idx_ref).nonzero()[:8]


def test_elementwise_family():
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    a, b, c = (torch.randn(2, 32, 32, 4, generator=g) for _ in range(3))
    out = ops.lincomb(0.7, a.to(DEV), -1.3, b.to(DEV), 0.25, c.to(DEV), b.to(DEV), div=0.9)
    ref = ((0.7 * a + (-1.3) * b) + 0.25 * (c - b)) / 0.9
    assert float((out.cpu() - ref).abs().max()) <= 2e-6
    out = ops.lincomb(1.0, a.to(DEV), -0.5, b.to(DEV), div=0.3)
    assert torch.equal(out.cpu(), (a - 0.5 * b) / torch.tensor(0.3))
    ca, cb = torch.rand(2, generator=g), torch.rand(2, generator=g)
    out = ops.row_lincomb(a.to(DEV), b.to(DEV), ca.to(DEV), cb.to(DEV))
    assert float((out.cpu() - (ca.view(2, 1, 1, 1) * a + cb.view(2, 1, 1, 1) * b)).abs().max()) <= 1e-6
    # layouts
    x = torch.randn(2, 3, 8, 8, generator=g)
    for dt in (torch.float32, torch.bfloat16):
        n = ops.nchw_to_nhwc(x.to(DEV), dt, 8)
        assert n.shape == (2, 8, 8, 8) and float(n[..., 3:].float().abs().max()) == 0
        back = ops.nhwc_to_nchw(n, 3)
        assert torch.equal(back.cpu(), x.to(dt).float())
    # timestep embedding, fractional t
    from oracle import slotdiff_oracle as O
    t = torch.tensor([0., 37., 123.456, 998.999])
    emb = ops.timestep_embedding(t.to(DEV), 128)
    assert float((emb.cpu() - O.timestep_embedding(t, 128)).abs().max()) <= 3e-4  # args up to 1e3 rad
    small = torch.tensor([0.5, 3.25])
    assert float((ops.timestep_embedding(small.to(DEV), 128).cpu() - O.timestep_embedding(small, 128)).abs().max()) <= 2e-6
    # geglu, concat, add_pos, act
    h = torch.randn(5, 7, 2 * 64, generator=g)
    xg, gate = h.chunk(2, -1)
    assert float((ops.geglu(h.to(DEV)).cpu() - xg * F.gelu(gate)).abs().max()) <= 2e-6
    p, r = torch.randn(3, 10, 16, generator=g), torch.randn(3, 10, 24, generator=g)
    assert torch.equal(ops.concat_channels(p.to(DEV), r.to(DEV)).cpu(), torch.cat([p, r], -1))
    pos = torch.randn(10, 16, generator=g)
    assert torch.equal(ops.add_pos(p.to(DEV), pos.to(DEV)).cpu(), p + pos)
    assert float((ops.act(p.to(DEV), 'silu').cpu() - F.silu(p)).abs().max()) <= 2e-6
    # mse
    pr, tg = torch.randn(2, 32, 32, 4, generator=g), torch.randn(2, 32, 32, 4, generator=g)
    val, grad = ops.mse(pr.to(DEV), tg.to(DEV), want_grad=True)
    assert abs(float(val) - float(F.mse_loss(pr, tg))) <= 1e-6
    assert float((grad.cpu() - 2 * (pr - tg) / pr.numel()).abs().max()) <= 1e-9


def test_mask_upsample_argmax_matches_torch():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    seg = torch.softmax(torch.randn(2, 32 * 32, 7, generator=g) * 2, -1)
    m = seg.permute(0, 2, 1).reshape(2 * 7, 1, 32, 32)
    ref = F.interpolate(m, (128, 128), mode='bilinear', align_corners=False).view(2, 7, 128, 128)
    up, idx = ops.mask_upsample_argmax(seg.to(DEV), 32, 32, 128, 128)
    assert float((up.cpu() - ref).abs().max()) <= 2e-7
    assert torch.equal(idx.cpu(), ref.argmax(1))


def test_eval_metrics_on_device():
    """slotdiffusion_amd.metrics (sdmi_contingency / sdmi_sqerr_rows + the reference's formulas on
    the table) against the reference's eval_utils values (tests/golden/metrics_b4.npz) and the
    oracle on larger random id maps; the contingency table itself is bit-exact."""
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd import metrics as MT
    from tests import common as C
    M = C.load_golden('metrics_b4.npz')
    gt, pred = M['gt'].long(), M['pred'].long()
    assert torch.equal(MT.adjusted_rand_index(gt.cuda(), pred.cuda(), False), M['ari_per_image'])
    assert torch.equal(MT.adjusted_rand_index(gt.cuda(), pred.cuda(), True), M['fari_per_image'])
    assert torch.equal(MT.adjusted_rand_index(gt.view(2, 2, 32, 32).cuda(), pred.view(2, 2, 32, 32).cuda(), True),
                       M['fari_video'])
    for name, fn in (('ari', MT.ARI_metric), ('fari', MT.fARI_metric), ('miou', MT.miou_metric),
                     ('fmiou', MT.fmiou_metric), ('mbo', MT.mbo_metric)):
        assert abs(float(fn(gt.cuda(), pred.cuda())) - float(M[name])) <= 1e-6, name
    # larger maps: table against a CPU bincount, scores against the oracle
    g = torch.Generator().manual_seed(11)
    B, T, H, W = 3, 6, 128, 128
    gtv = torch.randint(0, 12, (B, T, H, W), generator=g)
    pv = torch.where(torch.rand(B, T, H, W, generator=g) < 0.7, (gtv + 3) % 15, torch.randint(0, 15, (B, T, H, W), generator=g))
    tab = MT.contingency(gtv.cuda(), pv.cuda()).cpu()
    ref = torch.stack([torch.bincount((gtv[b] * 15 + pv[b]).flatten(), minlength=12 * 15).view(12, 15) for b in range(B)])
    assert tab.shape == (B, 12, 15) and torch.equal(tab.long(), ref)
    assert torch.equal(MT.adjusted_rand_index(gtv.cuda(), pv.cuda(), True), O.adjusted_rand_index(gtv, pv, True))
    r = O.seg_metrics(gtv[:, 0], pv[:, 0])
    assert abs(MT.fmiou_metric(gtv[:, 0].cuda(), pv[:, 0].cuda()) - r['fmiou']) <= 1e-6
    assert abs(MT.mbo_metric(gtv[:, 0].cuda(), pv[:, 0].cuda()) - r['mbo']) <= 1e-6
    x = torch.rand(4, 3, 128, 128, generator=g)
    y = (x + 0.03 * torch.randn(4, 3, 128, 128, generator=g)).clamp(0, 1)
    assert abs(MT.psnr_metric(x.cuda(), y.cuda()) - O.psnr_metric(x, y)) <= 1e-6
    assert abs(MT.mse_metric(x.cuda(), y.cuda()) - O.mse_metric(x, y)) <= 1e-6 * O.mse_metric(x, y)
    with pytest.raises(RuntimeError):
        MT.ARI_metric(gt, pred)           # CPU tensors: no fallback


def test_training_step_helpers():
    """The small stream-ordered helpers that keep framework kernels out of the training step:
    sdmi_draw_tn (timestep / noise draws + schedule gathers), sdmi_memset0, sdmi_scale_dev,
    sdmi_counters_inc, the zero-padding cast."""
    from slotdiffusion_amd import _lib
    ops = _ops()
    st = torch.cuda.current_stream().cuda_stream
    B, hw, T = 4096, 64, 1000
    tab_a = torch.linspace(1.0, 0.1, T, device=DEV)
    tab_b = torch.linspace(0.0, 0.9, T, device=DEV)
    seed_dev = torch.full((1,), 3, dtype=torch.int64, device=DEV)

    def draw(seed):
        t = torch.empty(B, dtype=torch.int64, device=DEV)
        tf, ca, cb = (torch.empty(B, device=DEV) for _ in range(3))
        nz = torch.empty(B, hw, 4, device=DEV)
        _lib.call('sdmi_draw_tn', st, t=t.data_ptr(), tf=tf.data_ptr(), ca=ca.data_ptr(), cb=cb.data_ptr(),
                  noise=nz.data_ptr(), tab_a=tab_a.data_ptr(), tab_b=tab_b.data_ptr(), B=B, T=T, per=hw * 4,
                  seed=seed, seed_dev=seed_dev.data_ptr())
        return t.cpu(), tf.cpu(), ca.cpu(), cb.cpu(), nz.cpu()
    t, tf, ca, cb, nz = draw(11)
    assert int(t.min()) >= 0 and int(t.max()) < T and torch.equal(tf, t.float())
    assert torch.equal(ca, tab_a.cpu()[t]) and torch.equal(cb, tab_b.cpu()[t])
    hist = torch.bincount(t // 100, minlength=10).float() / B          # uniform over the schedule
    assert float((hist - 0.1).abs().max()) < 0.02, hist
    z = nz[..., :3]
    assert float(nz[..., 3].abs().max()) == 0.0
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3
    assert abs(float((z ** 4).mean()) - 3.0) < 0.06 and float(z.abs().max()) > 4.0     # Gaussian tails
    t2, _, _, _, nz2 = draw(11)
    assert torch.equal(t, t2) and torch.equal(nz, nz2)                 # same seed word: reproducible
    _lib.call('sdmi_counters_inc', st, seed=seed_dev.data_ptr())
    assert int(seed_dev) == 4
    t3, _, _, _, nz3 = draw(11)
    assert not torch.equal(t, t3) and float((nz3 - nz).abs().mean()) > 0.5
    step = torch.zeros(1, dtype=torch.int32, device=DEV)
    _lib.call('sdmi_counters_inc', st, step=step.data_ptr(), seed=seed_dev.data_ptr())
    assert int(step) == 1 and int(seed_dev) == 5
    # memset / scale / zero-padding cast
    x = torch.randn(1000, 5, device=DEV)
    y = ops.cast2d(x, torch.bfloat16, cols=3, ldd=8)
    assert y.shape == (1000, 8) and float(y[:, 3:].float().abs().max()) == 0.0
    assert torch.equal(y[:, :3], x[:, :3].bfloat16())
    assert float(ops.zeros((777,), torch.float32, DEV).abs().max()) == 0.0
    s = torch.tensor([0.25], device=DEV)
    for dt in (torch.float32, torch.bfloat16):
        xs = torch.randn(3333, device=DEV).to(dt)
        out = torch.empty_like(xs)
        _lib.call('sdmi_scale_dev', st, x=xs.data_ptr(), y=out.data_ptr(), s=s.data_ptr(),
                  dtype=(_lib.BF16 if dt == torch.bfloat16 else _lib.F32), n=xs.numel())
        assert torch.equal(out, xs * 0.25)


def test_ssim_on_device_matches_oracle():
    """sdmi_ssim (fp64 windows on the device) against the oracle's scipy restatement (skimage itself
    is absent: parity unpinned at that boundary), incl. ragged sizes and identical images."""
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd import metrics
    g = torch.Generator().manual_seed(9)
    for B, H, W in ((2, 128, 128), (3, 37, 53), (1, 11, 11)):
        x = torch.rand(B, 3, H, W, generator=g)
        y = (x + 0.05 * torch.randn(B, 3, H, W, generator=g)).clamp(0, 1)
        ref = O.ssim_metric(x.numpy(), y.numpy())
        got = metrics.ssim_metric(x.to(DEV), y.to(DEV))
        assert abs(got - ref) < 1e-6, (got, ref)
    assert abs(metrics.ssim_metric(x.to(DEV), x.to(DEV)) - 1.0) < 1e-9
    slots = torch.randn(6, 3, 7, 16, generator=g)
    assert torch.equal(metrics.shuffle_slots(slots.to(DEV)).cpu(), O.shuffle_slots(slots))


def test_folded_slot_cross_attention_block():
    """Slot cross-attention with the 7 keys folded into the query weights and the values into the
    output projection (kern.Kern.cross_prepare / cross_block: sdmi_expand_heads, batched igemm with the
    LayerNorm-fold + softmax8 epilogue, batched igemm with bias + residual) against the plain sequence
    LayerNorm -> to_q -> softmax(q k^T / sqrt(d)) v -> to_out + residual in fp32 torch."""
    ops = _ops()
    dtype = torch.bfloat16
    for (B, HW, C, heads, S) in [(3, 256, 256, 8, 7), (2, 64, 384, 12, 7), (4, 16, 512, 16, 5)]:
        g = torch.Generator().manual_seed(B + HW + C)
        hd = C // heads
        tok = q(torch.randn(B, HW, C, generator=g) * 1.2 + 0.4, dtype)
        kv = q(torch.randn(B, S, 2 * C, generator=g), dtype)
        wq = torch.randn(C, C, generator=g) / math.sqrt(C)
        wo = q(torch.randn(C, C, generator=g) / math.sqrt(C), dtype)
        bo = torch.randn(C, generator=g)
        gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
        # reference
        n = F.layer_norm(tok, (C,), gamma, beta, 1e-5)
        qq = (n @ wq.t()).view(B, HW, heads, hd).permute(0, 2, 1, 3)
        kk = kv[..., :C].view(B, S, heads, hd).permute(0, 2, 1, 3)
        vv = kv[..., C:].view(B, S, heads, hd).permute(0, 2, 1, 3)
        att = ((qq @ kk.transpose(-1, -2)) * hd ** -0.5).softmax(-1) @ vv
        ref = att.permute(0, 2, 1, 3).reshape(B, HW, C) @ wo.t() + bo + tok
        # folded path (weight-side operands as WeightBank.cross_fold_weights builds them)
        kvd, tokd = kv.to(dtype).to(DEV), tok.to(dtype).to(DEV)
        wt = (wq * gamma).t().contiguous().to(dtype).to(DEV).unsqueeze(0)
        tb = (wq @ beta).to(dtype).view(1, 1, -1).to(DEV)
        ones = torch.ones(1, 1, C, dtype=dtype, device=DEV)
        R = heads * 8
        kexp, vexp = ops.expand_heads(kvd, heads, hd ** -0.5)
        wqe = ops.bmm_nt(kexp, wt, torch.empty(B, R, C, dtype=dtype, device=DEV))
        colsum = ops.bmm_nt(wqe, ones, torch.empty(B, R, 1, dtype=torch.float32, device=DEV)).view(B, R)
        biasq = ops.bmm_nt(kexp, tb, torch.empty(B, R, 1, dtype=torch.float32, device=DEV)).view(B, R)
        w2 = ops.bmm_nt(wo.to(dtype).to(DEV).view(1, C, C).expand(B, -1, -1), vexp,
                        torch.empty(B, C, R, dtype=dtype, device=DEV))
        P = ops.cross_scores(tokd, wqe, colsum, biasq, 1e-5, S)
        Pf = P.float().cpu().view(B, HW, heads, 8)
        assert float(Pf[..., S:].abs().max()) == 0.0                       # pad / absent slots get no mass
        assert float((Pf.sum(-1) - 1).abs().max()) < 2e-2
        out = ops.bmm_nt(P, w2, torch.empty_like(tokd), bias=bo.to(DEV), residual=tokd)
        e = rel_l2(out, ref)
        assert e <= 2e-2, (B, HW, C, e)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', [(2, 16, 16, 256, 256, 128, 128), (3, 8, 8, 384, 384, 384, 0), (64, 4, 4, 512, 512, 512, 512),
                                  (2, 32, 32, 128, 128, 256, 0)])
def test_conv3x3_with_extra_1x1_sources(case, dtype):
    """sdmi_igemm a2 / a3 on the convolution fast path: a 3x3 "same" convolution over h plus 1x1 taps
    over one or two more tensors as ONE implicit GEMM (a ResBlock's out_layers.3 + skip_connection),
    against the two convolutions in torch."""
    ops = _ops()
    B, H, W, C, N, Ca, Cb = case
    g = torch.Generator().manual_seed(sum(case))
    h = q(torch.randn(B, C, H, W, generator=g), dtype)
    xa = q(torch.randn(B, Ca, H, W, generator=g), dtype)
    xb = q(torch.randn(B, Cb, H, W, generator=g), dtype) if Cb else None
    w2 = q(torch.randn(N, C, 3, 3, generator=g) / math.sqrt(9 * C), dtype)
    ws = q(torch.randn(N, Ca + Cb, 1, 1, generator=g) / math.sqrt(Ca + Cb), dtype)
    bias = torch.randn(N, generator=g)
    xs = xa if xb is None else torch.cat([xa, xb], 1)
    ref = F.conv2d(h, w2, None, padding=1) + F.conv2d(xs, ws, bias)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)
    wk = torch.cat([w2.permute(0, 2, 3, 1).reshape(N, -1), ws.reshape(N, -1)], 1).contiguous().to(dtype).to(DEV)
    out = ops.conv2d(nhwc(h), wk, bias.to(DEV), kh=3, kw=3, pad=(1, 1, 1, 1), x2=nhwc(xa),
                     x3=(nhwc(xb) if xb is not None else None))
    check(out.permute(0, 3, 1, 2), ref, dtype, f'conv3x3 + 1x1 sources {case}')


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('shape', [(37, 64), (130, 1024), (9, 3136), (6, 520), (5, 4100), (7, 36)])
def test_softmax_rows(shape, dtype):
    """sdmi_softmax_rows (in place, scaled) against torch: register-resident rows up to 64 lanes x 8 vectors,
    the sweeping kernel beyond that / for unaligned widths."""
    ops = _ops()
    g = torch.Generator().manual_seed(shape[0] * shape[1])
    x = (torch.randn(shape, generator=g) * 3).to(dtype)
    ref = torch.softmax(x.float() * 0.37, -1)
    y = x.to(DEV).clone()
    ops.softmax_rows_(y, scale=0.37)
    tol = 1e-6 if dtype == torch.float32 else 4e-3
    assert float((y.float().cpu() - ref).abs().max()) <= tol
    assert float((y.float().cpu().sum(-1) - 1).abs().max()) <= (1e-5 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize('B,H,ci,co', [(3, 8, 256, 256), (2, 4, 512, 512), (2, 16, 128, 192),
                                        (64, 4, 512, 512), (64, 8, 384, 384), (64, 16, 256, 256)])   # the sampler's three at B = 64
def test_upsample_conv_as_four_parity_convs(B, H, ci, co):
    """nearest-2x + 3x3 convolution (unet.py:108-121) as four 2x2 convolutions of the low-resolution input
    (kern.ups_parity_split, sdmi.h: osy / osx) against torch fp32 and against the in-gather form it replaces."""
    from slotdiffusion_amd import kern
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, ci, H, H, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) * (ci * 9) ** -0.5
    b = torch.randn(co, generator=g)
    xb, wb_ = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(F.interpolate(xb, scale_factor=2, mode='nearest'), wb_, b, padding=1).permute(0, 2, 3, 1)
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    old = ops.conv2d(xd, w.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV), b.to(DEV), ups=True)
    out = torch.full((B, 2 * H, 2 * H, co), float('nan'), dtype=torch.bfloat16, device=DEV)
    for (py, px), wp in kern.ups_parity_split(w).items():
        ops.conv2d(xd, wp.bfloat16().to(DEV), b.to(DEV), kh=2, kw=2, pad=(1 - py, py, 1 - px, px), out=out,
                   split_k=1, sub=(2, 2, py, px))
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()              # every output pixel written by exactly one parity
    e_new = float((out.float().cpu() - ref).norm() / ref.norm())
    e_old = float((old.float().cpu() - ref).norm() / ref.norm())
    print(f'ups conv {ci}->{co} @{H}: parity form {e_new:.2e}, in-gather form {e_old:.2e} (rel-L2 vs torch fp32)')
    assert e_new < 6e-3 and e_old < 6e-3                  # bf16 operands, fp32 accumulation, bf16 output
    # ... and all four parities in ONE launch (sdmi.h: parity4): bit-identical to the four launches
    parts = kern.ups_parity_split(w)
    w4 = torch.stack([parts[py, px] for py in (0, 1) for px in (0, 1)]).bfloat16().to(DEV).contiguous()
    one = torch.full((B, 2 * H, 2 * H, co), float('nan'), dtype=torch.bfloat16, device=DEV)
    ops.conv2d(xd, w4, b.to(DEV), kh=2, kw=2, out=one, parity4=True)
    assert torch.equal(one, out)


@pytest.mark.parametrize('B,HW,C,slots,padded', [(64, 16, 512, 7, False), (5, 64, 384, 7, True), (3, 16, 256, 5, True),
                                                  (2, 32, 128, 7, False)])
def test_cross_fold_one_launch(B, HW, C, slots, padded):
    """sdmi_cross_fold against the two batched launches it replaces (softmax8 igemm + output projection) and a
    torch fp32 restatement of the folded layer."""
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    R = C // 4
    tok = torch.randn(B, HW, C, generator=g).bfloat16().to(DEV)
    if padded:          # operands inside the fused block's padded per-image storage (row / image pitches differ)
        src = torch.zeros(B, 128 * C + C * 128).bfloat16().to(DEV)
        wq = src[:, :R * C].view(B, R, C)
        w2 = src[:, 128 * C:].view(B, C, 128)[:, :, :R]
    else:
        wq = torch.empty(B, R, C).bfloat16().to(DEV)
        w2 = torch.empty(B, C, R).bfloat16().to(DEV)
    wq.copy_((torch.randn(B, R, C, generator=g) * C ** -0.5).bfloat16())
    w2.copy_((torch.randn(B, C, R, generator=g) * 0.3).bfloat16())
    colsum = wq.float().sum(-1).contiguous()
    biasq = torch.randn(B, R, generator=g).to(DEV)
    bias = torch.randn(C, generator=g).to(DEV)
    out = ops.cross_fold(tok, wq, colsum, biasq, w2, bias, 1e-5, slots)
    packed = ops.cross_fold(tok, torch.index_select(wq.reshape(B, R * C), 1, ops.cross_fold_pack_index(R, C).to(DEV)),
                            colsum, biasq,
                            torch.index_select(w2.reshape(B, C * R), 1, ops.cross_fold_pack_index(C, R).to(DEV)),
                            bias, 1e-5, slots, packed=True)
    P = ops.cross_scores(tok, wq, colsum, biasq, 1e-5, slots)
    two = ops.bmm_nt(P, w2, torch.empty_like(tok), bias=bias, residual=tok)
    x = tok.float()
    n = (x - x.mean(-1, keepdim=True)) * torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5)
    sc = (n @ wq.float().transpose(1, 2) + biasq[:, None]).view(B, HW, R // 8, 8)
    sc[..., slots:] = float('-inf')
    ref = torch.softmax(sc, -1).view(B, HW, R) @ w2.float().transpose(1, 2) + bias + x
    torch.cuda.synchronize()
    e1 = float((out.float() - ref).norm() / ref.norm())
    e2 = float((two.float() - ref).norm() / ref.norm())
    print(f'cross_fold C={C} HW={HW}: one launch {e1:.2e}, two launches {e2:.2e} (rel-L2 vs torch fp32)')
    assert e1 < 6e-3 and e1 < 2 * e2 + 1e-3
    assert torch.equal(packed, out)                       # fragment-order operands: the same arithmetic


def test_fp8_quant_group_device_scales():
    """sdmi_fp8_quant_group: a table of tensors quantised to e4m3fn with scale = 448 / amax taken on the device (no
    read-back), against sdmi_quant_fp8 with the same scale computed from torch's amax; the reciprocal scale lands
    in the tensor's slot (SdmiGemmArgs.alpha_dev).  Sizes: several workgroups, one partial workgroup, 16 elements;
    an all-zero tensor takes the 1e-12 floor."""
    from slotdiffusion_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(5)
    srcs = [torch.randn(9 * 4096 + 1600, device=DEV, generator=g) * 0.03,
            torch.randn(48, device=DEV, generator=g) * 7.0,
            torch.zeros(4096, device=DEV),
            torch.randn(16, device=DEV, generator=g) * 1e-3,
            torch.randn(3 * 4096, device=DEV, generator=g) * 300.0]
    Desc = _lib.CSTRUCT['SdmiFp8Desc']
    for sdt in (torch.float32, torch.bfloat16):
        xs = [s.to(sdt) for s in srcs]
        dsts = [torch.full((x.numel(),), 0x55, dtype=torch.uint8, device=DEV) for x in xs]
        arr = (Desc * len(xs))()
        blk = 0
        for d, x, o in zip(arr, xs, dsts):
            d.src, d.dst, d.n, d.block_begin = x.data_ptr(), o.data_ptr(), x.numel(), blk
            blk += (x.numel() + 4095) // 4096
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
        amax = torch.full((len(xs),), 123, dtype=torch.int32, device=DEV)       # garbage: the call zeroes it
        inv = torch.zeros(len(xs), device=DEV)
        for _ in range(2):       # a second call on the same workspace gives the same result
            _lib.call('sdmi_fp8_quant_group', None, descs=tab.data_ptr(), n_desc=len(xs), src_dtype=ops._DT[sdt],
                      total_blocks=blk, amax_bits=amax.data_ptr(), inv_scale=inv.data_ptr())
        torch.cuda.synchronize()
        for i, (x, o) in enumerate(zip(xs, dsts)):
            am = max(float(x.float().abs().max()), 1e-12)
            am32 = torch.tensor(am, dtype=torch.float32)
            assert float(amax.view(torch.float32)[i]) == float(x.float().abs().max())
            assert abs(float(inv[i]) - float(am32 / 448.0)) <= 1e-7 * float(am32 / 448.0) + 1e-30
            ref = ops.quant_fp8(x.view(1, -1), float(torch.tensor(448.0) / am32)).view(-1)
            mism = float((ref != o).float().mean())
            assert mism <= 1e-4, (sdt, i, mism)            # (a one-ulp difference of the scale may flip a rounding tie)
            # the bytes decode to x / inv within the format's resolution (2^-4 relative for normals)
            if am > 1e-12:
                dec = o.view(torch.float8_e4m3fn).float() * inv[i]
                assert float((dec - x.float()).abs().max()) <= am * 2 ** -4


@pytest.mark.parametrize('B,HW,C', [(2, 1024, 128), (3, 256, 256), (2, 64, 384), (1, 16384, 64), (2, 4096, 128)])
def test_groupnorm_writes_both_forms(B, HW, C):
    """fp8 configuration, training: one GroupNorm launch writes the bf16 output (the backward pass of the convolution
    behind it reads that) AND its e4m3fn copy (the forward GEMM's operand) -- both bit-identical to the launches that
    write one form each, with the fused dropout on (the same mask in both forms)."""
    from slotdiffusion_amd import ops
    g = torch.Generator(device=DEV).manual_seed(B + HW + C)
    x = torch.randn(B, HW, C, device=DEV, generator=g).bfloat16()
    gamma, beta = torch.randn(C, device=DEV, generator=g), torch.randn(C, device=DEV, generator=g)
    seed_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
    for drop in (None, (0.1, 1234, seed_dev)):
        y = ops.group_norm(x, gamma, beta, eps=1e-5, act='silu', drop=drop)
        y8 = ops.group_norm(x, gamma, beta, eps=1e-5, act='silu', drop=drop, fp8_scale=8.0)
        also = []
        yb = ops.group_norm(x, gamma, beta, eps=1e-5, act='silu', drop=drop, fp8_scale=8.0, fp8_also=also)
        assert yb.dtype == torch.bfloat16 and len(also) == 1 and also[0].dtype == torch.uint8
        assert torch.equal(yb, y) and torch.equal(also[0], y8)
