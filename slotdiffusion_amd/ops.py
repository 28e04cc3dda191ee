"""Tensor-level wrappers over the libsdmi C ABI (forward kernels).

PyTorch is plumbing here: tensors provide device memory (`data_ptr()`) and the current HIP
stream; every FLOP happens in the hand-written kernels of ``csrc/``.  All activations are NHWC.
"""
import torch

from . import _lib
from ._lib import ACT, BF16, F32, FP8, call

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.uint8: FP8}   # uint8 = raw e4m3fn bytes


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dt(t):
    return _DT[t.dtype]


def _p(t):
    return 0 if t is None else t.data_ptr()


def vec_of(dtype):
    return 8 if dtype == torch.bfloat16 else 4


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.SdmiError('libsdmi kernels need device tensors (no CPU fallback)')


def zero_(t):
    """Stream-ordered zero fill of a contiguous tensor (memset node in a graph; no framework kernel)."""
    call('sdmi_memset0', _stream(), ptr=_p(t), bytes=t.numel() * t.element_size())
    return t


def zeros(shape, dtype, device):
    return zero_(torch.empty(shape, dtype=dtype, device=device))


# ------------------------------------------------------------------------------------------
# implicit GEMM
# ------------------------------------------------------------------------------------------
_SPLITK_BELOW = 384       # output tiles (64 x 64)
_SPLITK_MINKB = 2048      # bytes of K per row


def splitk_workspace(M, N, K, elt, device):
    """fp32 partial-sum workspace for skinny problems (few output tiles, deep K) or None: the
    launcher then picks the K split itself (igemm.hip: dispatch)."""
    t64 = ((M + 63) // 64) * ((N + 63) // 64)
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    if not (N > 64 and t128 >= 192) and t64 < _SPLITK_BELOW and K * elt >= _SPLITK_MINKB:
        return torch.empty((16 * M * N,), dtype=torch.float32, device=device)
    return None


# Deferred split-K second stages (inference; sdmi.h: defer_epilogue).  Inside `defer_splitk()` a convolution whose
# launch splits K leaves its partials in the workspace; if the very next kernel is the GroupNorm that reads the result
# (unet.py:243-250: conv -> GroupNorm32 -> SiLU), that kernel's prologue finishes the reduction (sdmi_groupnorm: part)
# and also stores the tensor for its other readers.  ANY other launch first finishes what is pending (`_lib.pre_call`),
# so a pending tensor is never read unfinished; results are bit-identical either way.
# The protection covers readers that go through `_lib.call` only -- a torch op, `.cpu()` or a side-stream reader of a
# pending tensor inside the context would see unwritten memory.  Every consumer inside `UNetRunner.forward` is a library
# call; SDMI_DEBUG_DEFER=1 makes a violation visible instead of silent: a deferring convolution first fills its output
# with NaN (the finishing kernel overwrites every element), so any reader that slipped past the hook yields NaN.
_DEBUG_DEFER = bool(__import__('slotdiffusion_amd.policy', fromlist=['flag']).flag('DEBUG_DEFER'))


class _PendingSplit:
    __slots__ = ('out', 'kw', 'splits', 'keep')

    def __init__(self, out, kw, splits, keep):
        self.out, self.kw, self.splits, self.keep = out, kw, splits, keep


_PENDING = {}            # out.data_ptr() -> _PendingSplit
_DEFER = [0]


def _finish_split(p):
    kw = dict(p.kw)
    kw['split_k'] = p.splits
    call('sdmi_splitk_finish', _stream(), **kw)


def flush_pending():
    while _PENDING:
        _, p = _PENDING.popitem()
        _finish_split(p)


class defer_splitk:
    """Context: split-K convolutions may leave their second stage to the GroupNorm behind them."""

    def __enter__(self):
        _DEFER[0] += 1
        _lib.pre_call = _pre_call
        return self

    def __exit__(self, *a):
        _DEFER[0] -= 1
        flush_pending()
        if _DEFER[0] == 0:
            _lib.pre_call = None


# GroupNorm statistics written by the producing convolution's epilogue (sdmi.h: gn_part; inference, inside
# defer_splitk()): (out tensor, partial sums, nsplit, groups) of the LAST convolution launched.  Valid only for the
# launch that directly follows it -- the GroupNorm reading `out` then runs as the apply pass alone.
_GN_LAST = [None]
# Measured SLOWER in the sampler (the 32 x 32-tile statistics cost the symmetric-wave kernel +2.8 us per
# launch, and the apply pass with 8 - 32 partials to fold per group is a two-round-trip latency chain like the
# single-pass kernel it replaces: 75.4 -> 76.8 ms per pass): off.
GN_EPILOGUE_STATS = False          # (module variable: the test of the mechanism turns it on)


def _pre_call():
    _GN_LAST[0] = None
    if _PENDING:
        flush_pending()


def quant_fp8(x, scale, out=None):
    """Per-tensor e4m3fn quantisation of a contiguous [..., C] tensor (C % 16 == 0): uint8 bytes of
    clamp(x * scale, +-448).  Operands of the SDMI_FP8 igemm path."""
    _need_gpu(x)
    C = x.shape[-1]
    assert x.is_contiguous() and C % 16 == 0
    if out is None:
        out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    call('sdmi_quant_fp8', _stream(), src=_p(x), dst=_p(out), src_dtype=_dt(x), rows=x.numel() // C,
         cols=C, lds=C, ldd=C, scale=float(scale))
    return out


def conv2d(x, w, bias=None, *, kh=3, kw=3, stride=1, pad=(1, 1, 1, 1), ups=False, cout=None,
           rowvec=None, residual=None, act=None, out=None, out_dtype=None, ldc=None,
           split_k=0, alpha=1.0, x2=None, x3=None, sub=None, zero_pad=True, alpha_dev=None, parity4=False):
    """x [B,H,W,Cin] NHWC; w [Cout][kh][kw][Cin] (flat or 4-D channels_last view).
    pad = (top, bottom, left, right).  Returns [B,Ho,Wo,ldc or Cout].
    uint8 x / w = e4m3fn operands (quant_fp8): out_dtype is required, alpha undoes the scales; alpha_dev = a
    one-element fp32 tensor multiplied into alpha on the device (a scale sdmi_fp8_quant_group derived there).
    sub = (sy, sx, oy, ox) with `out` [B, sy*Ho, sx*Wo, ldc]: output pixel (y, x) is stored at (sy*y + oy, sx*x + ox)
    (sdmi.h: osy / osx / ooy / oox; bias-only epilogue, no split-K).
    parity4: w [4][Cout][2*2*Cin] = the four parity filters of an upsample convolution (kern.ups_parity_split, z = 2 py +
    px), out [B, 2H, 2W, ldc]: all four 2x2 parity convolutions in ONE launch (sdmi.h: parity4)."""
    _need_gpu(x, w)
    B, H, W, Cin = x.shape
    assert x.is_contiguous()
    C1 = Cin
    extra = 0
    if x2 is not None:          # extra sources appended along K as 1x1 taps (sdmi.h: a2 / a3)
        assert stride == 1 and not ups and x2.is_contiguous() and x2.shape[:3] == x.shape[:3]
        assert (kh, kw) == (1, 1) or pad == (kh // 2, kh // 2, kw // 2, kw // 2)
        extra = x2.shape[-1]
        if x3 is not None:
            assert x3.is_contiguous() and x3.shape[:3] == x.shape[:3]
            extra += x3.shape[-1]
    Hs, Ws = (2 * H, 2 * W) if ups else (H, W)
    Ho = (Hs + pad[0] + pad[1] - kh) // stride + 1
    Wo = (Ws + pad[2] + pad[3] - kw) // stride + 1
    K = kh * kw * Cin + extra
    if parity4:
        assert (kh, kw) == (2, 2) and stride == 1 and not ups and x2 is None and out is not None and rowvec is None and \
            residual is None and act is None and w.dim() == 3 and w.shape[0] == 4 and w.is_contiguous()
        pad, sub, split_k, cout = (1, 0, 1, 0), (2, 2, 0, 0), 1, w.shape[1]
        Ho, Wo = H, W
    N = cout if cout is not None else w.numel() // K
    assert w.numel() == N * K * (4 if parity4 else 1), (w.shape, N, K)
    odt = out_dtype or x.dtype
    ldc = ldc or N
    if out is None:
        out = torch.empty((B, Ho, Wo, ldc), dtype=odt, device=x.device)
        if ldc != N and zero_pad:         # channel-pad columns must read as zeros downstream
            zero_(out)
    M = B * Ho * Wo
    if x.dtype == torch.uint8:
        assert out_dtype is not None and w.dtype == torch.uint8
    ws = splitk_workspace(M, N, K, x.element_size(), x.device) if split_k != 1 else None
    kwargs = dict(
        a=_p(x), w=_p(w), out=_p(out), bias=_p(bias), rowvec=_p(rowvec),
        residual=_p(residual), workspace=_p(ws), dtype=_dt(x), out_dtype=_DT[odt], M=M, N=N, K=K,
        lda=C1, ldw=K, ldc=ldc, ldr=(residual.shape[-1] if residual is not None else 0),
        a2=_p(x2), lda2=(x2.shape[-1] if x2 is not None else 0), K1=(kh * kw * C1 if x2 is not None else 0),
        a3=_p(x3), lda3=(x3.shape[-1] if x3 is not None else 0),
        K2=(kh * kw * C1 + x2.shape[-1] if x3 is not None else 0),
        B=B, H=H, W=W, Cin=Cin, Ho=Ho, Wo=Wo, KH=kh, KW=kw, stride=stride, pad_t=pad[0],
        pad_l=pad[2], ups=int(ups), act=ACT[act], alpha=float(alpha), bias_m=0,
        ldrv=(rowvec.stride(0) if rowvec is not None else 0),
        split_k=(split_k if ws is not None or split_k == 1 else 1), batch=1, alpha_dev=_p(alpha_dev))
    if sub is not None:
        assert split_k == 1 and rowvec is None and residual is None and act is None and x2 is None
        assert tuple(out.shape[:3]) == (B, sub[0] * Ho, sub[1] * Wo) and out.is_contiguous()
        kwargs.update(oH=sub[0] * Ho, oW=sub[1] * Wo, osy=sub[0], osx=sub[1], ooy=sub[2], oox=sub[3])
        if parity4:
            kwargs.update(parity4=1, batch=4, sw=N * K)
    if _DEFER[0] and alpha_dev is None and ws is not None and split_k == 0 and act is None and ldc == N and N > 64 and N % 8 == 0 and \
            odt == torch.bfloat16 and (residual is None or (residual.is_contiguous() and residual.shape[-1] == N)):
        splits = _lib.query('sdmi_igemm_split_plan', **kwargs)
        if splits > 1:
            if _DEBUG_DEFER:
                out.fill_(float('nan'))
            call('sdmi_igemm', _stream(), defer_epilogue=1, **kwargs)
            _PENDING[out.data_ptr()] = _PendingSplit(out, kwargs, splits, (ws, bias, rowvec, residual, float(alpha)))
            return out
    gn = None
    hw = Ho * Wo
    if _DEFER[0] and GN_EPILOGUE_STATS and sub is None and odt == torch.bfloat16 and x.dtype != torch.float32 and \
            ws is None and M % 128 == 0 and N % 128 == 0 and N % 32 == 0 and (N // 32) in (4, 8, 16, 32) and \
            hw >= 32 and hw & (hw - 1) == 0 and ldc == N and act is None and B * hw * N < (1 << 30):
        # the GroupNorm (32 groups) that may follow gets its statistics from this launch's epilogue
        gn = torch.empty((B, hw // 32, 32, 2), dtype=torch.float32, device=x.device)
        kwargs.update(gn_part=_p(gn), gn_groups=32, split_k=1)
    call('sdmi_igemm', _stream(), **kwargs)
    if gn is not None:
        _GN_LAST[0] = (out, gn, hw // 32, 32)
    return out


def linear(x, w, bias=None, *, act=None, residual=None, out=None, out_dtype=None, n=None,
           alpha=1.0, lda=None, k=None, ln_colsum=None, ln_eps=0.0, geglu=False, out2=None):
    """x [..., K] (last dim contiguous; row pitch lda) ; w [N, K] -> [..., N].
    ln_colsum: fused LayerNorm prologue (w pre-scaled by gamma, bias = W beta + b; see sdmi.h);
    geglu: w [2N, K] (value rows, gate rows) -> value * gelu(gate), N columns; out2 [..., 2N]: also keep
    the pre-activation (training)."""
    _need_gpu(x, w)
    K = k or x.shape[-1]
    lda = lda or x.stride(-2) if x.dim() > 1 else K
    M = x.numel() // x.shape[-1]
    N = n if n is not None else w.numel() // K // (2 if geglu else 1)
    odt = out_dtype or x.dtype
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=odt, device=x.device)
    ws = None
    t64 = ((M + 63) // 64) * ((N + 63) // 64)
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    if not (N > 64 and t128 >= 192) and t64 < 384 and K * x.element_size() >= 2048 and \
            ln_colsum is None and not geglu:
        ws = torch.empty((16 * M * N,), dtype=torch.float32, device=x.device)
    call('sdmi_igemm', _stream(), a=_p(x), w=_p(w), out=_p(out), bias=_p(bias),
         residual=_p(residual), workspace=_p(ws), dtype=_dt(x), out_dtype=_DT[odt], M=M, N=N, K=K,
         lda=lda, ldw=K, ldc=out.stride(-2) if out.dim() > 1 else N,
         ldr=(residual.stride(-2) if residual is not None else 0), B=M, H=1, W=1, Cin=K, Ho=1,
         Wo=1, KH=1, KW=1, stride=1, pad_t=0, pad_l=0, ups=0, act=ACT[act], alpha=alpha,
         bias_m=0, split_k=(0 if ws is not None else 1), batch=1, ln_colsum=_p(ln_colsum),
         ln_eps=float(ln_eps), geglu=int(geglu), out2=_p(out2),
         ldc2=(out2.stride(-2) if out2 is not None else 0))
    return out


def bmm_nt(a, b, out, *, alpha=1.0, bias_m=None, bias=None, residual=None):
    """Batched out[z] = alpha * a[z] @ b[z]^T (+ bias_m[:, None] | + bias[None, :]) (+ residual[z]);
    a [Z,M,K] (stride(0) may be 0: shared), b [Z,N,K] or [1,N,K] (shared)."""
    _need_gpu(a, b, out)
    Z, M, K = a.shape
    N = b.shape[1]
    assert bias_m is None or bias is None
    call('sdmi_igemm', _stream(), a=_p(a), w=_p(b), out=_p(out), bias=_p(bias_m if bias_m is not None else bias),
         dtype=_dt(a), out_dtype=_dt(out), M=M, N=N, K=K, lda=a.stride(1), ldw=b.stride(1), ldc=out.stride(1),
         B=M, H=1, W=1, Cin=K, Ho=1, Wo=1, KH=1, KW=1, stride=1, act=0, alpha=alpha,
         bias_m=int(bias_m is not None), split_k=1, batch=Z, sa=a.stride(0),
         sw=(b.stride(0) if b.shape[0] == Z and Z > 1 else 0), sc=out.stride(0),
         residual=_p(residual), ldr=(residual.stride(1) if residual is not None else 0),
         sr=(residual.stride(0) if residual is not None else 0))
    return out


def expand_heads(kv, heads, scale, gw=8):
    """kv [B,S,2C] (K | V) -> (kexp, vexp) [B, heads*gw, C]: slot j's key (scaled) / value restricted to
    head h's channels in row h*gw+j, zeros elsewhere (sdmi.h: sdmi_expand_heads; gw = 8, or 16 from 9 slots)."""
    _need_gpu(kv)
    B, S, C2 = kv.shape
    C = C2 // 2
    kexp = torch.empty((B, heads * gw, C), dtype=kv.dtype, device=kv.device)
    vexp = torch.empty_like(kexp)
    call('sdmi_expand_heads', _stream(), kv=_p(kv), kexp=_p(kexp), vexp=_p(vexp), dtype=_dt(kv), B=B, S=S,
         C=C, heads=heads, ldkv=kv.stride(1), scale=float(scale), gw=int(gw))
    return kexp, vexp


def cross_scores(tok, wq, colsum, biasq, eps, slots=7):
    """Attention probabilities of the folded slot cross-attention: softmax over each head's `slots` scores of
    LayerNorm(tok[b]) @ wq[b]^T, the norm folded into the GEMM (sdmi.h: ln_colsum, softmax8).
    tok [B,HW,C]; wq [B,R,C] (R = heads * 8, or heads * 16 from 9 slots); colsum / biasq [B,R] fp32 -> P [B,HW,R]."""
    _need_gpu(tok, wq)
    B, HW, C = tok.shape
    R = wq.shape[1]
    out = torch.empty((B, HW, R), dtype=tok.dtype, device=tok.device)
    call('sdmi_igemm', _stream(), a=_p(tok), w=_p(wq), out=_p(out), bias=_p(biasq), dtype=_dt(tok),
         out_dtype=_dt(out), M=HW, N=R, K=C, lda=tok.stride(1), ldw=wq.stride(1), ldc=R, B=HW, H=1, W=1, Cin=C, Ho=1,
         Wo=1, KH=1, KW=1, stride=1, act=0, alpha=1.0, bias_m=0, split_k=1, batch=B, sa=tok.stride(0),
         sw=wq.stride(0), sc=HW * R, ln_colsum=_p(colsum), ln_eps=float(eps), s_colsum=colsum.stride(0),
         s_bias=biasq.stride(0), softmax8=int(slots))
    return out


CROSS_FOLD_SHAPES = ((512, 128), (384, 96), (256, 64), (128, 32))       # (C, R) instantiations of sdmi_cross_fold


def cross_fold_pack_index(rows, cols):
    """Gather index that re-lays a [rows, cols] operand in MFMA-fragment order (sdmi.h: SdmiCrossFoldArgs.packed):
    [tile of 16 rows][k step of 32][lane][8], lane = (k sub-group) * 16 + row in tile."""
    t, ks, lane, e = torch.meshgrid(torch.arange(rows // 16), torch.arange(cols // 32), torch.arange(64),
                                    torch.arange(8), indexing='ij')
    return ((t * 16 + (lane & 15)) * cols + ks * 32 + (lane >> 4) * 8 + e).reshape(-1)


def gather_rows(src, idx):
    """src [B, n_src] (2-byte elements, row pitch src.stride(0)), idx int64 [n] on the device -> [B, n] = src[:, idx]."""
    _need_gpu(src, idx)
    assert src.element_size() == 2 and src.stride(1) == 1 and idx.dtype == torch.int64 and idx.is_contiguous()
    B, n = src.shape[0], idx.numel()
    out = torch.empty((B, n), dtype=src.dtype, device=src.device)
    call('sdmi_gather_rows', _stream(), src=_p(src), dst=_p(out), idx=_p(idx), n=n, s_src=src.stride(0), s_dst=n, B=B)
    return out


def cross_fold(tok, wq, colsum, biasq, w2, bias, eps, slots=7, packed=False):
    """The folded slot cross-attention layer in one launch (sdmi.h: sdmi_cross_fold): out = softmax_slots(LayerNorm(tok)
    wq^T + biasq) w2^T + bias + tok.  tok [B,HW,C] contiguous; wq [B,R,C], w2 [B,C,R] (row / image pitches free), or
    packed: both [B, R*C] in fragment order (cross_fold_pack_index)."""
    _need_gpu(tok, wq, w2)
    B, HW, C = tok.shape
    R = wq.shape[1] // C if packed else wq.shape[1]
    assert tok.is_contiguous() and wq.stride(-1) == 1 and w2.stride(-1) == 1
    out = torch.empty_like(tok)
    call('sdmi_cross_fold', _stream(), tok=_p(tok), out=_p(out), wq=_p(wq), w2=_p(w2), colsum=_p(colsum),
         biasq=_p(biasq), bias=_p(bias), B=B, HW=HW, C=C, R=R, slots=int(slots),
         ld_wq=(0 if packed else wq.stride(1)), ld_w2=(0 if packed else w2.stride(1)), s_wq=wq.stride(0),
         s_w2=w2.stride(0), s_colsum=colsum.stride(0), s_bias=biasq.stride(0), ln_eps=float(eps), packed=int(packed),
         _meta=dict(flops=4.0 * B * HW * R * C, bytes=2.0 * (2 * B * HW * C + 2 * B * R * C)))
    return out


# ------------------------------------------------------------------------------------------
# normalisation
# ------------------------------------------------------------------------------------------
def group_norm(x, gamma, beta, *, eps, act=None, groups=32, residual=None, out=None,
               return_stats=False, drop=None, fp8_scale=None, x2=None, fp8_also=None):
    """x [B,H,W,C] (or [B,HW,C]) NHWC -> same shape; fp32 statistics.  drop = (p, seed, seed_dev):
    inverted dropout fused behind the activation (training-mode ResBlocks).  fp8_scale: the output
    is written as e4m3fn bytes (uint8 tensor) for the fp8 convolution behind the norm.  fp8_also = a list
    (with fp8_scale): the launch writes BOTH forms -- the returned tensor is the bf16 output (the backward pass of
    the convolution reads it), the e4m3fn bytes are appended to the list (training in the fp8 configuration)."""
    _need_gpu(x)
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    C1 = 0
    if x2 is not None:          # input = channel concatenation [x | x2], read in place (sdmi.h: x2)
        assert x.is_contiguous() and x2.is_contiguous() and x2.shape[:-1] == x.shape[:-1] and out is None
        C1, C = C, C + x2.shape[-1]
    nsplit = max(1, min(16, HW // 64))
    partial = torch.empty((B * nsplit * groups * 2,), dtype=torch.float32, device=x.device)
    stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    oshape = x.shape[:-1] + (C,)
    if fp8_scale is not None and fp8_also is not None:
        out = torch.empty(oshape, dtype=x.dtype, device=x.device)
        out8 = torch.empty(oshape, dtype=torch.uint8, device=x.device)
        fp8_also.append(out8)
        kw = dict(y=_p(out), y8=_p(out8), y8_scale=float(fp8_scale))
    elif fp8_scale is not None:
        out = torch.empty(oshape, dtype=torch.uint8, device=x.device)
        kw = dict(y=0, y8=_p(out), y8_scale=float(fp8_scale))
    else:
        if out is None:
            out = torch.empty(oshape, dtype=x.dtype, device=x.device)
        kw = dict(y=_p(out))
    if x2 is not None:
        kw.update(x2=_p(x2), C1=C1)
    last = _GN_LAST[0]
    if last is not None and x2 is None and last[0].data_ptr() == x.data_ptr() and last[0].numel() == x.numel() and \
            last[3] == groups and x.is_contiguous():
        # statistics came with the producing convolution: the apply pass alone (one read, one write, no reduction)
        kw.update(x=_p(x), gamma=_p(gamma), beta=_p(beta), stats=_p(stats), partial=_p(last[1]), dtype=_dt(x), B=B,
                  HW=HW, C=C, groups=groups, eps=eps, act=ACT[act], nsplit=last[2], residual=_p(residual))
        if drop is not None and drop[0] > 0.0:
            kw.update(drop_p=float(drop[0]), drop_seed=int(drop[1]), drop_seed_dev=_p(drop[2]))
        call('sdmi_groupnorm_apply', _stream(), **kw)
        return (out, stats) if return_stats else out
    pend = _PENDING.pop(x.data_ptr(), None) if _PENDING else None
    if pend is not None:
        if not x.is_contiguous() or x.dtype != torch.bfloat16 or pend.out.numel() != x.numel():
            _finish_split(pend)               # (not a form the norm's prologue takes: the stand-alone second stage)
        else:                                 # this launch finishes the split-K reduction of its input
            ws, pb, prv, pres, palpha = pend.keep
            kw.update(part=_p(ws), part_splits=pend.splits, part_alpha=palpha, part_bias=_p(pb),
                      part_rowvec=_p(prv), part_ldrv=(prv.stride(0) if prv is not None else 0),
                      part_residual=_p(pres), raw_out=_p(x))
    kw.update(x=_p(x), gamma=_p(gamma), beta=_p(beta), stats=_p(stats),
              partial=_p(partial), dtype=_dt(x), B=B, HW=HW, C=C, groups=groups, eps=eps,
              act=ACT[act], nsplit=nsplit, residual=_p(residual))
    if drop is not None and drop[0] > 0.0:
        kw.update(drop_p=float(drop[0]), drop_seed=int(drop[1]), drop_seed_dev=_p(drop[2]))
    call('sdmi_groupnorm', _stream(), **kw)      # one fused launch for small images, else two
    return (out, stats) if return_stats else out


def layer_norm(x, gamma, beta, *, eps=1e-5, out=None, stats=None):  # noqa: D401
    _need_gpu(x)
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    call('sdmi_layernorm', _stream(), x=_p(x), y=_p(out), gamma=_p(gamma), beta=_p(beta),
         stats=_p(stats), dtype=_dt(x), rows=rows, C=C, ldx=C, ldy=C, eps=eps)
    return out


def softmax_rows_(x, scale=1.0):
    """In-place row softmax over the last dim; rows may be padded (uniform row pitch x.stride(-2))."""
    cols = x.shape[-1]
    ld = x.stride(-2) if x.dim() > 1 else cols
    call('sdmi_softmax_rows', _stream(), x=_p(x), dtype=_dt(x), rows=x.numel() // cols, cols=cols,
         ld=ld, scale=scale)
    return x


# ------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------
def attention(q, k, v, heads, *, out=None, lse=None, head_dim=32):
    """q [B,Sq,*] k,v [B,Skv,*] (views with last-dim stride 1; head h at channel h*32)."""
    _need_gpu(q, k, v)
    B, Sq = q.shape[0], q.shape[1]
    Skv = k.shape[1]
    C = heads * head_dim
    if out is None:
        out = torch.empty((B, Sq, C), dtype=q.dtype, device=q.device)
    call('sdmi_attention', _stream(), q=_p(q), k=_p(k), v=_p(v), out=_p(out), lse=_p(lse),
         dtype=_dt(q), B=B, heads=heads, Sq=Sq, Skv=Skv, ldq=q.stride(1), ldk=k.stride(1),
         ldv=v.stride(1), ldo=out.stride(1), scale=head_dim ** -0.5, head_dim=head_dim)
    return out


def slot_attention(k, v, slots_in, P, *, iters, eps, trace=None):
    """k, v [B,M,D] views (row pitch = stride(1)); slots_in [N,D] or [B,N,D] fp32.
    P: dict of fp32 weights (lnq_g, lnq_b, wq, w_ih, w_hh, b_ih, b_hh, lnm_g, lnm_b, w1, b1, w2, b2).
    Returns slots [B,N,D] fp32, seg [B,M,N] fp32."""
    _need_gpu(k, v, slots_in)
    B, M, D = k.shape
    N = slots_in.shape[-2]
    Hid = P['w1'].shape[0]
    slots = torch.empty((B, N, D), dtype=torch.float32, device=k.device)
    seg = torch.empty((B, M, N), dtype=torch.float32, device=k.device)
    call('sdmi_slot_attention', _stream(), k=_p(k), v=_p(v), slots_in=_p(slots_in),
         slots_out=_p(slots), seg=_p(seg), trace=_p(trace), dtype=_dt(k), B=B, M=M, N=N, D=D,
         Hid=Hid, iters=iters, ldkv=k.stride(1),
         slots_bstride=(N * D if slots_in.dim() == 3 else 0), eps=eps, scale=D ** -0.5,
         **{n: _p(P[n]) for n in ('lnq_g', 'lnq_b', 'wq', 'w_ih', 'w_hh', 'b_ih', 'b_hh',
                                  'lnm_g', 'lnm_b', 'w1', 'b1', 'w2', 'b2')})
    return slots, seg


# ------------------------------------------------------------------------------------------
# VQ + elementwise
# ------------------------------------------------------------------------------------------
def vq_nearest(z, codebook, *, scale=1.0, want_idx=True, want_zq=True, comb=None):
    """z [..., ldz] fp32 NHWC latent (first 3 channels used). -> (idx int64 [...], zq like z).
    comb = (c0, c1, z2, div): the latent is (c0 * z + c1 * z2) / div, formed on load with lincomb's arithmetic."""
    _need_gpu(z, codebook)
    ldz = z.shape[-1]
    R = z.numel() // ldz
    idx = torch.empty(z.shape[:-1], dtype=torch.int64, device=z.device) if want_idx else None
    zq = torch.empty_like(z) if want_zq else None
    kw = {}
    if comb is not None:
        c0, c1, z2, div = comb
        assert z2.shape == z.shape and z2.is_contiguous() and z2.dtype == torch.float32
        kw = dict(z2=_p(z2), zc0=float(c0), zc1=float(c1), zdiv=float(div))
    call('sdmi_vq_nearest', _stream(), z=_p(z), codebook=_p(codebook), idx=_p(idx), zq=_p(zq), R=R,
         dim=codebook.shape[1], ldz=ldz, n_codes=codebook.shape[0], scale=scale, **kw)
    return idx, zq


def lincomb(c0=0., x0=None, c1=0., x1=None, c2=0., x2=None, x3=None, div=0., out=None):
    """out = ((c0*x0 + c1*x1) + c2*(x2 - x3)) / div  (fp32 tensors; see sdmi.h)."""
    ref = x0 if x0 is not None else (x1 if x1 is not None else x2)
    _need_gpu(ref)
    if out is None:
        out = torch.empty_like(ref)
    call('sdmi_lincomb', _stream(), y=_p(out), x0=_p(x0), x1=_p(x1), x2=_p(x2), x3=_p(x3),
         c0=float(c0), c1=float(c1), c2=float(c2), div=float(div), n=out.numel())
    return out


def row_lincomb(x0, x1, ca, cb, out=None):
    B = x0.shape[0]
    if out is None:
        out = torch.empty_like(x0)
    call('sdmi_row_lincomb', _stream(), y=_p(out), x0=_p(x0), x1=_p(x1), ca=_p(ca), cb=_p(cb), B=B,
         per=x0.numel() // B)
    return out


def nchw_to_nhwc(src, dtype, cpad=None):
    _need_gpu(src)
    B, C, H, W = src.shape
    cpad = cpad or C
    dst = torch.empty((B, H, W, cpad), dtype=dtype, device=src.device)
    call('sdmi_nchw_to_nhwc', _stream(), src=_p(src.contiguous()), dst=_p(dst), dtype=_DT[dtype],
         B=B, C=C, H=H, W=W, Cpad=cpad)
    return dst


def nhwc_to_nchw(src, C=None):
    _need_gpu(src)
    B, H, W, cpad = src.shape
    C = C or cpad
    dst = torch.empty((B, C, H, W), dtype=torch.float32, device=src.device)
    call('sdmi_nhwc_to_nchw', _stream(), src=_p(src), dst=_p(dst), dtype=_dt(src), B=B, C=C, H=H,
         W=W, Cpad=cpad)
    return dst


def cast2d(src, dst_dtype, cols=None, out=None, ldd=None):
    """Cast a [rows, lds] matrix (first `cols` columns) into [rows, ldd] of dst_dtype."""
    _need_gpu(src)
    lds = src.shape[-1]
    cols = cols or lds
    rows = src.numel() // lds
    ldd = ldd or cols
    zpad = 0
    if out is None:
        out = torch.empty(src.shape[:-1] + (ldd,), dtype=dst_dtype, device=src.device)
        zpad = int(ldd != cols)           # the kernel writes the pad columns itself
    call('sdmi_cast2d', _stream(), src=_p(src), dst=_p(out), src_dtype=_dt(src),
         dst_dtype=_DT[dst_dtype], rows=rows, cols=cols, lds=lds, ldd=ldd, zpad=zpad)
    return out


def timestep_embedding(t, dim, max_period=10000.):
    _need_gpu(t)
    out = torch.empty((t.shape[0], dim), dtype=torch.float32, device=t.device)
    call('sdmi_timestep_embedding', _stream(), t=_p(t), out=_p(out), B=t.shape[0], dim=dim,
         max_period=max_period)
    return out


def act(x, kind, dst_dtype=None):
    _need_gpu(x)
    dst_dtype = dst_dtype or x.dtype
    y = torch.empty(x.shape, dtype=dst_dtype, device=x.device)
    call('sdmi_act', _stream(), x=_p(x), y=_p(y), src_dtype=_dt(x), dst_dtype=_DT[dst_dtype],
         act=ACT[kind], n=x.numel())
    return y


def geglu(h):
    C = h.shape[-1] // 2
    y = torch.empty(h.shape[:-1] + (C,), dtype=h.dtype, device=h.device)
    call('sdmi_geglu', _stream(), h=_p(h), y=_p(y), dtype=_dt(h), rows=h.numel() // (2 * C), C=C)
    return y


def add_pos(x, pos):
    """x [B, P, C] + pos [P, C] (fp32)."""
    y = torch.empty_like(x)
    B = x.shape[0]
    call('sdmi_add_pos', _stream(), x=_p(x), pos=_p(pos), y=_p(y), dtype=_dt(x), B=B,
         per=x.numel() // B)
    return y


def concat_channels(a, b):
    Ca, Cb = a.shape[-1], b.shape[-1]
    y = torch.empty(a.shape[:-1] + (Ca + Cb,), dtype=a.dtype, device=a.device)
    call('sdmi_concat_channels', _stream(), a=_p(a), b=_p(b), y=_p(y), dtype=_dt(a),
         rows=a.numel() // Ca, Ca=Ca, Cb=Cb)
    return y


def mask_upsample_argmax(seg, h, w, H, W, want_up=True):
    """seg [B, h*w, N] fp32 -> (up [B,N,H,W] fp32 or None, idx [B,H,W] int64)."""
    B, _, N = seg.shape
    up = torch.empty((B, N, H, W), dtype=torch.float32, device=seg.device) if want_up else None
    idx = torch.empty((B, H, W), dtype=torch.int64, device=seg.device)
    call('sdmi_mask_upsample_argmax', _stream(), seg=_p(seg), up=_p(up), idx=_p(idx), B=B, N=N,
         h=h, w=w, H=H, W=W)
    return up, idx


def transpose2d(x, out=None):
    """[Z,R,C] (row pitch = stride(1), unit column stride) -> contiguous [Z,C,R]."""
    _need_gpu(x)
    Z, R, C = x.shape
    assert x.stride(2) == 1
    if out is None:
        out = torch.empty((Z, C, R), dtype=x.dtype, device=x.device)
    call('sdmi_transpose2d', _stream(), src=_p(x), dst=_p(out), dtype=_dt(x), Z=Z, R=R, C=C,
         lds=x.stride(1), ldd=out.stride(1), ss=x.stride(0), sd=out.stride(0))
    return out


def softmax_rows_bwd_(p, dp, scale=1.0):
    """dp <- scale * p * (dp - rowsum(dp * p)) in place (p = softmax output, rows of the last dim;
    p and dp share the row pitch)."""
    _need_gpu(p, dp)
    cols = p.shape[-1]
    ld = p.stride(-2) if p.dim() > 1 else cols
    assert dp.stride(-2) == ld
    call('sdmi_softmax_rows_bwd', _stream(), p=_p(p), dp=_p(dp), dtype=_dt(p), rows=p.numel() // cols,
         cols=cols, ld=ld, scale=scale)
    return dp


ATTN_LDS_MAX_KV = 400       # longer key sequences do not fit the LDS-resident attention kernels
ATTN_MFMA_MAX_KV = 1 << 20  # ... except the bf16 matrix-core kernels (head_dim 32 / 64), which chunk the keys through LDS


def attention_long(q, k, v, heads, head_dim, keep_p=False):
    """Multi-head attention composed from GEMMs for what the attention kernels do not cover (fp32 parity
    runs beyond 400 keys; head sizes other than 32 / 64): per head, batched MFMA GEMMs S = q k^T -> row
    softmax -> O = P v, with the score matrix in HBM.  q [B,Sq,*], k / v [B,Skv,*] views with head h at
    channel h * head_dim.  -> out [B,Sq,heads*hd] (, P [heads][B,Sq,Sp] if keep_p)."""
    _need_gpu(q, k, v)
    B, Sq, Skv, hd = q.shape[0], q.shape[1], k.shape[1], head_dim
    vec = vec_of(q.dtype)
    Sp = (Skv + vec - 1) // vec * vec                 # key pitch of the score matrix
    out = torch.empty((B, Sq, heads * hd), dtype=q.dtype, device=q.device)
    P = zeros((heads if keep_p else 1, B, Sq, Sp), q.dtype, q.device)     # pad columns stay zero
    vt = zeros((B, hd, Sp), q.dtype, q.device)
    for h in range(heads):
        sl = slice(h * hd, (h + 1) * hd)
        sc = P[h if keep_p else 0]
        bmm_nt(q[..., sl], k[..., sl], sc[..., :Skv])
        softmax_rows_(sc[..., :Skv], scale=float(hd) ** -0.5)
        transpose2d(v[..., sl], out=vt[..., :Skv])
        bmm_nt(sc, vt, out[..., sl])
    return (out, P) if keep_p else out


def mse(pred, target, want_grad=False, gscale=1.0, l1=False, oscale=0.0):
    n = pred.numel()
    nblk = max(1, min(1024, (n + 2047) // 2048))
    partial = torch.empty((nblk,), dtype=torch.float32, device=pred.device)
    out = torch.empty((1,), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    call('sdmi_mse', _stream(), pred=_p(pred), target=_p(target), out=_p(out), dpred=_p(dpred),
         partial=_p(partial), dtype=_dt(pred), n=n, nblk=nblk, gscale=gscale, l1=int(l1),
         oscale=float(oscale))
    return (out, dpred) if want_grad else out
