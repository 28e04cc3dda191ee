"""The large-tile implicit GEMMs behind sdmi_igemm's dispatch (halo-staged 3x3 kernel csrc/igemm_halo.h, symmetric-wave
and LDS-DMA 128 x 128 kernels) against torch fp32 convolutions of the bf16-rounded operands, at the benchmark's batch
(>= 192 tiles of 256 x 128, >= 8 K tiles): the UNet's 32^2 convolutions (unet.py:219-259), the skip-concat forms of the
sampler (two / three A sources), linear layers, ragged M / N / K, several output tiles per workgroup."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _ops():
    from slotdiffusion_amd import ops
    return ops


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


def _ref_conv(x, w4, bias=None, k=3):
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w4.float().permute(0, 3, 1, 2), bias, padding=k // 2)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize('B,H,C,N', [(64, 32, 128, 128),      # 256 tiles: one per CU
                                      (64, 32, 256, 256),      # 512 tiles: two per workgroup (flat pipeline)
                                      (50, 32, 192, 192),      # 200 x 2 tiles, ragged N (1.5 tiles), Cin = 3 K tiles per tap
                                      (99, 24, 128, 128)])     # non-power-of-two image, ragged M (222.75 tiles)
def test_big_conv3x3_plain(B, H, C, N):
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(B + H + C + N)
    w = (torch.randn(N, 3, 3, C, device=DEV, generator=g) / (9 * C) ** 0.5).bfloat16()
    x = torch.randn(B, H, H, C, device=DEV, generator=g).bfloat16()
    y = ops.conv2d(x, w, None)
    ref = _ref_conv(x, w)
    assert _rel(y, ref) < 4e-3
    # run-to-run bitwise repeatable (no race in the DMA ring: a late piece would change single tiles)
    for _ in range(3):
        assert torch.equal(ops.conv2d(x, w, None), y)


@pytest.mark.parametrize('B,H,C,N', [(256, 16, 256, 256),     # W = 16: a tile is a whole image (top and bottom rows zero)
                                      (16, 64, 128, 128),      # W = 64: four image rows per tile, 50-piece patches
                                      (64, 32, 64, 128),       # one 64-channel chunk per tile (patch buffers alternate per tile)
                                      (64, 32, 384, 384),      # six chunks, three column tiles, three tiles per workgroup
                                      (200, 16, 128, 192),     # W = 16, ragged N (1.5 column tiles)
                                      (4, 128, 128, 128)])     # 128-column image: 4 x 64-pixel tiles with real left / right neighbours
def test_halo_conv3x3_geometries(B, H, C, N):
    """The halo-staged 3x3 kernel (csrc/igemm_halo.h) at the image widths it serves; bias + residual + row vector."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(B + H + C + N + 1)
    w = (torch.randn(N, 3, 3, C, device=DEV, generator=g) / (9 * C) ** 0.5).bfloat16()
    x = torch.randn(B, H, H, C, device=DEV, generator=g).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    rowvec = torch.randn(B, N, device=DEV, generator=g)
    res = torch.randn(B, H, H, N, device=DEV, generator=g).bfloat16()
    y = ops.conv2d(x, w, bias, rowvec=rowvec, residual=res)
    ref = _ref_conv(x, w, bias) + rowvec.view(B, 1, 1, N) + res.float()
    assert _rel(y, ref) < 4e-3
    # image borders exactly: an all-ones input and filter count the taps inside the image
    ones = torch.ones(-(-208 * 256 // (H * H)), H, H, 64, device=DEV).bfloat16()      # >= 208 tiles of 256 pixels
    w1 = torch.ones(128, 3, 3, 64, device=DEV).bfloat16()
    cnt = ops.conv2d(ones, w1, None).float()[..., 0] / 64.0
    ky = torch.full((H,), 3.0, device=DEV); ky[0] = ky[-1] = 2.0
    assert torch.equal(cnt, (ky.view(1, H, 1) * ky.view(1, 1, H)).expand_as(cnt))
    for _ in range(3):
        assert torch.equal(ops.conv2d(x, w, bias, rowvec=rowvec, residual=res), y)


def test_big_conv3x3_full_epilogue():
    """bias + per-image time-embedding row + residual + SiLU (the ResBlock forms, unet.py:271-285)."""
    ops = _ops()
    B, H, C, N = 64, 32, 128, 128
    g = torch.Generator(device=DEV).manual_seed(7)
    w = (torch.randn(N, 3, 3, C, device=DEV, generator=g) / (9 * C) ** 0.5).bfloat16()
    x = torch.randn(B, H, H, C, device=DEV, generator=g).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    rowvec = torch.randn(B, N, device=DEV, generator=g)
    res = torch.randn(B, H, H, N, device=DEV, generator=g).bfloat16()
    y = ops.conv2d(x, w, bias, rowvec=rowvec, residual=res)
    ref = _ref_conv(x, w, bias) + rowvec.view(B, 1, 1, N) + res.float()
    assert _rel(y, ref) < 4e-3
    y = ops.conv2d(x, w, bias, act='silu')
    assert _rel(y, F.silu(_ref_conv(x, w, bias))) < 5e-3


@pytest.mark.parametrize('three', [False, True])
@pytest.mark.parametrize('shape', [(64, 32, 128, 128, 128, 64), (64, 16, 256, 256, 256, 128), (64, 16, 256, 256, 128, 64)])
def test_big_conv3x3_extra_sources(three, shape):
    """out_layers.3 + skip_connection as one GEMM over [im2col(h) | skip a | skip b] (kern.res_tail): the 32^2 level
    (symmetric-wave kernel) and the 16^2 level at B = 64 (twelve-wave LDS-DMA kernel with extra-source loaders)."""
    ops = _ops()
    B, H, C, N, C2, C3 = shape
    g = torch.Generator(device=DEV).manual_seed(8 + three)
    h = torch.randn(B, H, H, C, device=DEV, generator=g).bfloat16()
    a = torch.randn(B, H, H, C2, device=DEV, generator=g).bfloat16()
    b = torch.randn(B, H, H, C3, device=DEV, generator=g).bfloat16() if three else None
    w3 = (torch.randn(N, 3, 3, C, device=DEV, generator=g) / (9 * C) ** 0.5).bfloat16()
    wa = (torch.randn(N, C2, device=DEV, generator=g) / C2 ** 0.5).bfloat16()
    wb = (torch.randn(N, C3, device=DEV, generator=g) / C3 ** 0.5).bfloat16() if three else None
    wcat = torch.cat([w3.reshape(N, -1), wa] + ([wb] if three else []), dim=1).contiguous()
    bias = torch.randn(N, device=DEV, generator=g)
    y = ops.conv2d(h, wcat, bias, kh=3, kw=3, pad=(1, 1, 1, 1), x2=a, x3=b, cout=N)
    ref = _ref_conv(h, w3, bias) + a.float() @ wa.float().t()
    if three:
        ref = ref + b.float() @ wb.float().t()
    assert _rel(y, ref) < 4e-3


@pytest.mark.parametrize('M,N,K', [(65536, 256, 512), (65536, 128, 520), (50000, 384, 1024)])
def test_big_linear(M, N, K):
    """1x1 / linear problems incl. a K tail (520 = 8 K tiles + 8 columns) and ragged M."""
    ops = _ops()
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    w = (torch.randn(N, K, device=DEV, generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device=DEV, generator=g)
    y = ops.linear(x, w, bias)
    ref = x.float() @ w.float().t() + bias
    assert _rel(y, ref) < 4e-3
