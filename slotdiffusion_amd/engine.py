"""Forward programs of the hot path on the libsdmi kernels (NHWC, fp32 or bf16 compute).

The networks are executed from the flat checkpoint-key dict (see spec.py / module.py): each
function below is the MI355X-side restatement of one reference `forward`, built from fused
kernel calls instead of nn.Module graphs.  Reference call sites are cited per function.

Weight preparation (WeightBank): GEMM/conv operands are needed in the compute dtype, K-contiguous,
with Cin padded to the 16-byte vector width and with per-block projections fused along N
(q|k|v, cross-attn k|v, all ResBlock time-embedding projections, slot-attention k|v).  The bank
materialises those once per weight version; biases / norm affines stay fp32 masters.
"""
import torch

from . import ops, spec


class WeightBank:
    def __init__(self, tensors, dtype):
        self.t = tensors                  # {checkpoint key: fp32 master tensor on device}
        self.dtype = dtype
        self.cache = {}

    def invalidate(self):
        self.cache.clear()

    def f(self, name):
        """fp32 master (bias, norm affine, codebook ...)."""
        return self.t[name]

    def w(self, name):
        """GEMM operand [N, K] in compute dtype (conv weights: K = kh*kw*Cin_padded)."""
        if name in self.cache:
            return self.cache[name]
        p = self.t[name]
        vec = ops.vec_of(self.dtype)
        if p.dim() == 4:
            cout, cin, kh, kw = p.shape
            assert p.is_contiguous(memory_format=torch.channels_last) or (kh == 1 and kw == 1)
            flat = p.permute(0, 2, 3, 1)          # [Cout,kh,kw,Cin] view of the same bytes
            if not flat.is_contiguous():
                flat = flat.contiguous()
            cpad = (cin + vec - 1) // vec * vec
            if cpad != cin or self.dtype != torch.float32:
                out = ops.cast2d(flat.reshape(cout * kh * kw, cin), self.dtype, cols=cin, ldd=cpad)
                out = out.view(cout, kh * kw * cpad)
            else:
                out = flat.reshape(cout, kh * kw * cin)
        else:
            n, k = p.shape
            kpad = (k + vec - 1) // vec * vec
            if kpad != k or self.dtype != torch.float32:
                out = ops.cast2d(p, self.dtype, cols=k, ldd=kpad)
            else:
                out = p
        self.cache[name] = out
        return out

    def fused(self, key, names):
        """Concatenate several [N_i, K] operands along N into one prepared operand."""
        if key in self.cache:
            return self.cache[key]
        parts = [self.w(n) for n in names]
        k = parts[0].shape[1]
        out = torch.empty((sum(p.shape[0] for p in parts), k), dtype=self.dtype,
                          device=parts[0].device)
        o = 0
        for p in parts:
            ops.cast2d(p, self.dtype, out=out[o:o + p.shape[0]])
            o += p.shape[0]
        self.cache[key] = out
        return out

    def fused_f32(self, key, names):
        if key in self.cache:
            return self.cache[key]
        parts = [self.t[n].reshape(-1, 1) for n in names]
        out = torch.empty((sum(p.shape[0] for p in parts),), dtype=torch.float32,
                          device=parts[0].device)
        o = 0
        for p in parts:
            ops.cast2d(p, torch.float32, out=out[o:o + p.shape[0]].view(-1, 1))
            o += p.shape[0]
        self.cache[key] = out
        return out


def _bias(wb, name):
    return wb.t.get(name + '.bias')


# ------------------------------------------------------------------------------------------
# a1/a2: ResNet-18(GN) + SoftPositionEmbed + encoder head     (resnet.py:294-312,
#        img_based/models/slot_attention.py:305-316, models/utils.py:60-63)
# ------------------------------------------------------------------------------------------
def resnet_encoder(wb, x, plan, prefix='encoder'):
    """x [B,H,W,Cpad] compute dtype -> [B,H/4,W/4,256]."""
    h = ops.conv2d(x, wb.w(f'{prefix}.conv1.weight'))
    h = ops.group_norm(h, wb.f(f'{prefix}.bn1.weight'), wb.f(f'{prefix}.bn1.bias'), eps=1e-5,
                       act='relu')
    for blk, cin, cout, stride, has_ds in plan:
        b = f'{prefix}.{blk}'
        o = ops.conv2d(h, wb.w(f'{b}.conv1.weight'), stride=stride)
        o = ops.group_norm(o, wb.f(f'{b}.bn1.weight'), wb.f(f'{b}.bn1.bias'), eps=1e-5, act='relu')
        o = ops.conv2d(o, wb.w(f'{b}.conv2.weight'))
        idt = h
        if has_ds:
            idt = ops.conv2d(h, wb.w(f'{b}.downsample.0.weight'), kh=1, kw=1, stride=stride,
                             pad=(0, 0, 0, 0))
            idt = ops.group_norm(idt, wb.f(f'{b}.downsample.1.weight'),
                                 wb.f(f'{b}.downsample.1.bias'), eps=1e-5)
        # relu(gn(conv2) + identity) fused in the GN apply kernel
        h = ops.group_norm(o, wb.f(f'{b}.bn2.weight'), wb.f(f'{b}.bn2.bias'), eps=1e-5, act='relu',
                           residual=idt)
    return h


def position_embedding(wb, name='encoder_pos_embedding'):
    """Linear(4->C)(grid) -> fp32 [h*w, C]; input independent, cached with the weights."""
    key = name + '/pos'
    if key not in wb.cache:
        grid = wb.t[f'{name}.grid']                         # [1,h,w,4] fp32
        g = grid.reshape(-1, 4).contiguous()
        w = wb.t[f'{name}.dense.weight']                    # [C,4] fp32, K=4 -> one fp32 vector
        wb.cache[key] = ops.linear(g, w, wb.t[f'{name}.dense.bias'])
    return wb.cache[key]


def encoder_out(wb, img_nhwc, plan):
    """-> tokens [B, h*w, enc_out] in compute dtype."""
    feat = resnet_encoder(wb, img_nhwc, plan)
    B, h, w, C = feat.shape
    tok = ops.add_pos(feat.view(B, h * w, C), position_embedding(wb))
    tok = ops.layer_norm(tok, wb.f('encoder_out_layer.0.weight'), wb.f('encoder_out_layer.0.bias'))
    tok = ops.linear(tok, wb.w('encoder_out_layer.1.weight'), wb.f('encoder_out_layer.1.bias'),
                     act='relu')
    return ops.linear(tok, wb.w('encoder_out_layer.3.weight'), wb.f('encoder_out_layer.3.bias'))


# ------------------------------------------------------------------------------------------
# a3: Slot Attention with mask (img_based/models/sa_diffusion.py:16-70)
# ------------------------------------------------------------------------------------------
def slot_attention(wb, tokens, slots_init, iters, eps, name='slot_attention'):
    """tokens [B,M,Cin]; slots_init [N,D] or [B,N,D] fp32 -> slots [B,N,D] fp32, seg [B,M,N]."""
    x = ops.layer_norm(tokens, wb.f(f'{name}.norm_inputs.weight'), wb.f(f'{name}.norm_inputs.bias'))
    wkv = wb.fused(f'{name}/kv', [f'{name}.project_k.weight', f'{name}.project_v.weight'])
    kv = ops.linear(x, wkv)                                   # [B,M,2D]
    D = kv.shape[-1] // 2
    P = dict(lnq_g=wb.f(f'{name}.project_q.0.weight'), lnq_b=wb.f(f'{name}.project_q.0.bias'),
             wq=wb.f(f'{name}.project_q.1.weight'), w_ih=wb.f(f'{name}.gru.weight_ih'),
             w_hh=wb.f(f'{name}.gru.weight_hh'), b_ih=wb.f(f'{name}.gru.bias_ih'),
             b_hh=wb.f(f'{name}.gru.bias_hh'), lnm_g=wb.f(f'{name}.mlp.0.weight'),
             lnm_b=wb.f(f'{name}.mlp.0.bias'), w1=wb.f(f'{name}.mlp.1.weight'),
             b1=wb.f(f'{name}.mlp.1.bias'), w2=wb.f(f'{name}.mlp.3.weight'),
             b2=wb.f(f'{name}.mlp.3.bias'))
    return ops.slot_attention(kv[..., :D], kv[..., D:], slots_init, P, iters=iters, eps=eps)


# ------------------------------------------------------------------------------------------
# a9-a11: LDM UNet (unet.py:551-576, 271-285; attention.py:297-308, 247-251, 182-206)
# ------------------------------------------------------------------------------------------
class UNetRunner:
    """Holds the block plan plus per-weight-version fused operands for one UNet."""

    def __init__(self, wb, cfg, prefix='dm_decoder.model.diffusion_model'):
        self.wb, self.cfg, self.P = wb, cfg, prefix + '.'
        self.plan = spec.unet_plan(cfg)
        self.mc = cfg['model_channels']
        self.res_names = []
        for blk in self.plan['input'] + [self.plan['middle']] + self.plan['output']:
            for l in blk:
                if l[0] == 'res':
                    self.res_names.append((l[1], l[3]))
        self.emb_off = {}
        o = 0
        for n, c in self.res_names:
            self.emb_off[n] = (o, c)
            o += c
        self.emb_total = o
        self.st_names = [l[1] for blk in self.plan['input'] + [self.plan['middle']] +
                         self.plan['output'] for l in blk if l[0] == 'st']

    # -- per-call invariants -------------------------------------------------------------
    def time_rowvecs(self, t):
        """t [B] fp32 -> fp32 [B, sum(Cout)]: every ResBlock's Linear(SiLU(emb)) in ONE GEMM."""
        wb, P = self.wb, self.P
        dt = wb.dtype
        e = ops.timestep_embedding(t, self.mc)
        if dt != torch.float32:
            e = ops.act(e, None, dt)
        e = ops.linear(e, wb.w(P + 'time_embed.0.weight'), wb.f(P + 'time_embed.0.bias'), act='silu')
        # SiLU(emb) is what every ResBlock consumes -> fold the SiLU into this GEMM's epilogue
        e = ops.linear(e, wb.w(P + 'time_embed.2.weight'), wb.f(P + 'time_embed.2.bias'), act='silu')
        w = wb.fused(P + '/emb_w', [P + n + '.emb_layers.1.weight' for n, _ in self.res_names])
        b = wb.fused_f32(P + '/emb_b', [P + n + '.emb_layers.1.bias' for n, _ in self.res_names])
        return ops.linear(e, w, b, out_dtype=torch.float32)

    def context_kv(self, ctx):
        """ctx [B,N,Dc] compute dtype -> {st name: kv [B,N,2C]}; constant across all NFEs."""
        out = {}
        for n in self.st_names:
            t = self.P + n + '.transformer_blocks.0.attn2'
            w = self.wb.fused(t + '/kv', [t + '.to_k.weight', t + '.to_v.weight'])
            out[n] = ops.linear(ctx, w)
        return out

    # -- blocks ---------------------------------------------------------------------------
    def _res(self, name, x, rowvecs):
        wb, n = self.wb, self.P + name
        off, cout = self.emb_off[name]
        rv = rowvecs[:, off:off + cout]      # strided view; the kernel takes its row pitch
        h = ops.group_norm(x, wb.f(n + '.in_layers.0.weight'), wb.f(n + '.in_layers.0.bias'),
                           eps=1e-5, act='silu')
        h = ops.conv2d(h, wb.w(n + '.in_layers.2.weight'), wb.f(n + '.in_layers.2.bias'), rowvec=rv)
        h = ops.group_norm(h, wb.f(n + '.out_layers.0.weight'), wb.f(n + '.out_layers.0.bias'),
                           eps=1e-5, act='silu')
        skip = x
        if (n + '.skip_connection.weight') in wb.t:
            skip = ops.conv2d(x, wb.w(n + '.skip_connection.weight'),
                              wb.f(n + '.skip_connection.bias'), kh=1, kw=1, pad=(0, 0, 0, 0))
        return ops.conv2d(h, wb.w(n + '.out_layers.3.weight'), wb.f(n + '.out_layers.3.bias'),
                          residual=skip)

    def _st(self, name, x, heads, kv):
        wb, n = self.wb, self.P + name
        B, H, W, C = x.shape
        h = ops.group_norm(x, wb.f(n + '.norm.weight'), wb.f(n + '.norm.bias'), eps=1e-6)
        tok = ops.linear(h.view(B, H * W, C), wb.w(n + '.proj_in.weight'), wb.f(n + '.proj_in.bias'))
        t = n + '.transformer_blocks.0'
        # self attention
        n1 = ops.layer_norm(tok, wb.f(t + '.norm1.weight'), wb.f(t + '.norm1.bias'))
        wqkv = wb.fused(t + '.attn1/qkv', [t + '.attn1.to_q.weight', t + '.attn1.to_k.weight',
                                           t + '.attn1.to_v.weight'])
        qkv = ops.linear(n1, wqkv)
        a = ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], heads)
        tok = ops.linear(a, wb.w(t + '.attn1.to_out.0.weight'), wb.f(t + '.attn1.to_out.0.bias'),
                         residual=tok)
        # slot cross attention (K/V precomputed per sample call)
        n2 = ops.layer_norm(tok, wb.f(t + '.norm2.weight'), wb.f(t + '.norm2.bias'))
        q = ops.linear(n2, wb.w(t + '.attn2.to_q.weight'))
        a = ops.attention(q, kv[..., :C], kv[..., C:], heads)
        tok = ops.linear(a, wb.w(t + '.attn2.to_out.0.weight'), wb.f(t + '.attn2.to_out.0.bias'),
                         residual=tok)
        # GEGLU feed-forward
        n3 = ops.layer_norm(tok, wb.f(t + '.norm3.weight'), wb.f(t + '.norm3.bias'))
        g = ops.linear(n3, wb.w(t + '.ff.net.0.proj.weight'), wb.f(t + '.ff.net.0.proj.bias'))
        g = ops.geglu(g)
        tok = ops.linear(g, wb.w(t + '.ff.net.2.weight'), wb.f(t + '.ff.net.2.bias'), residual=tok)
        out = ops.linear(tok, wb.w(n + '.proj_out.weight'), wb.f(n + '.proj_out.bias'),
                         residual=x.view(B, H * W, C))
        return out.view(B, H, W, C)

    def _run(self, layers, h, rowvecs, ctx_kv):
        wb = self.wb
        for l in layers:
            kind, name = l[0], l[1]
            n = self.P + name
            if kind == 'conv':
                h = ops.conv2d(h, wb.w(n + '.weight'), wb.f(n + '.bias'))
            elif kind == 'res':
                h = self._res(name, h, rowvecs)
            elif kind == 'st':
                h = self._st(name, h, l[3], ctx_kv[name])
            elif kind == 'down':
                h = ops.conv2d(h, wb.w(n + '.op.weight'), wb.f(n + '.op.bias'), stride=2)
            elif kind == 'up':      # nearest x2 folded into the conv's gather
                h = ops.conv2d(h, wb.w(n + '.conv.weight'), wb.f(n + '.conv.bias'), ups=True)
        return h

    def forward(self, x, rowvecs, ctx_kv, out=None):
        """x [B,h,w,Cpad] compute dtype -> eps [B,h,w,4] fp32 (3 channels + zero pad)."""
        hs = []
        h = x
        for blk in self.plan['input']:
            h = self._run(blk, h, rowvecs, ctx_kv)
            hs.append(h)
        h = self._run(self.plan['middle'], h, rowvecs, ctx_kv)
        for blk in self.plan['output']:
            h = self._run(blk, ops.concat_channels(h, hs.pop()), rowvecs, ctx_kv)
        wb, P = self.wb, self.P
        h = ops.group_norm(h, wb.f(P + 'out.0.weight'), wb.f(P + 'out.0.bias'), eps=1e-5, act='silu')
        return ops.conv2d(h, wb.w(P + 'out.2.weight'), wb.f(P + 'out.2.bias'),
                          out_dtype=torch.float32, ldc=4, out=out)


# ------------------------------------------------------------------------------------------
# a6/a14/a15: VQ-VAE (VQVAE.py:94-114, 183-194; modules.py:239-261, 338-362)
# ------------------------------------------------------------------------------------------
def _vae_res(wb, n, x):
    h = ops.group_norm(x, wb.f(n + '.norm1.weight'), wb.f(n + '.norm1.bias'), eps=1e-6, act='silu')
    h = ops.conv2d(h, wb.w(n + '.conv1.weight'), wb.f(n + '.conv1.bias'))
    h = ops.group_norm(h, wb.f(n + '.norm2.weight'), wb.f(n + '.norm2.bias'), eps=1e-6, act='silu')
    skip = x
    if (n + '.nin_shortcut.weight') in wb.t:
        skip = ops.conv2d(x, wb.w(n + '.nin_shortcut.weight'), wb.f(n + '.nin_shortcut.bias'),
                          kh=1, kw=1, pad=(0, 0, 0, 0))
    return ops.conv2d(h, wb.w(n + '.conv2.weight'), wb.f(n + '.conv2.bias'), residual=skip)


def _vae_attn(wb, n, x):
    """Single-head attention over h*w tokens with head dim C (modules.py:130-154), as GEMMs:
    S = scale * Q K^T (batched), row softmax, O = P V computed as P @ (V^T)^T."""
    B, H, W, C = x.shape
    S = H * W
    h = ops.group_norm(x, wb.f(n + '.norm.weight'), wb.f(n + '.norm.bias'), eps=1e-6).view(B, S, C)
    wqk = wb.fused(n + '/qk', [n + '.q.weight', n + '.k.weight'])
    bqk = wb.fused_f32(n + '/qk_b', [n + '.q.bias', n + '.k.bias'])
    qk = ops.linear(h, wqk, bqk)                                # [B,S,2C]
    # V^T [B,C,S] = Wv [C,Cin] @ h[b]^T : batched GEMM with the weight as the row operand
    vt = torch.empty((B, C, S), dtype=x.dtype, device=x.device)
    wv = wb.w(n + '.v.weight')
    ops.bmm_nt(wv.unsqueeze(0).expand(B, C, C), h, vt, bias_m=wb.f(n + '.v.bias'))
    sc = torch.empty((B, S, S), dtype=x.dtype, device=x.device)
    ops.bmm_nt(qk[..., :C], qk[..., C:], sc)
    ops.softmax_rows_(sc, scale=float(C) ** -0.5)
    o = torch.empty((B, S, C), dtype=x.dtype, device=x.device)
    ops.bmm_nt(sc, vt, o)
    out = ops.linear(o, wb.w(n + '.proj_out.weight'), wb.f(n + '.proj_out.bias'),
                     residual=x.view(B, S, C))
    return out.view(B, H, W, C)


def vae_encode(wb, img_nhwc, ed, prefix='dm_decoder.vae.vqvae', scale_factor=1.0):
    """img [B,H,W,Cpad] -> x0 [B,h,w,4] fp32 (3 latent channels + zero pad)."""
    e = prefix + '.encoder'
    mult, nrb = tuple(ed['ch_mult']), ed['num_res_blocks']
    h = ops.conv2d(img_nhwc, wb.w(e + '.conv_in.weight'), wb.f(e + '.conv_in.bias'))
    for lvl in range(len(mult)):
        for b in range(nrb):
            h = _vae_res(wb, f'{e}.down.{lvl}.block.{b}', h)
        if lvl != len(mult) - 1:       # asymmetric (0,1,0,1) zero pad + stride 2, pad handled in-kernel
            h = ops.conv2d(h, wb.w(f'{e}.down.{lvl}.downsample.conv.weight'),
                           wb.f(f'{e}.down.{lvl}.downsample.conv.bias'), stride=2, pad=(0, 1, 0, 1))
    h = _vae_res(wb, e + '.mid.block_1', h)
    h = _vae_attn(wb, e + '.mid.attn_1', h)
    h = _vae_res(wb, e + '.mid.block_2', h)
    h = ops.group_norm(h, wb.f(e + '.norm_out.weight'), wb.f(e + '.norm_out.bias'), eps=1e-6,
                       act='silu')
    vec = ops.vec_of(wb.dtype)
    h = ops.conv2d(h, wb.w(e + '.conv_out.weight'), wb.f(e + '.conv_out.bias'), ldc=vec)
    z = ops.conv2d(h, wb.w(prefix + '.quant_conv.weight'), wb.f(prefix + '.quant_conv.bias'),
                   kh=1, kw=1, pad=(0, 0, 0, 0), out_dtype=torch.float32, ldc=4)
    if scale_factor != 1.0:
        z = ops.lincomb(1.0, z, div=scale_factor)
    return z


def vae_decode(wb, z, ed, prefix='dm_decoder.vae.vqvae', scale_factor=1.0, quantize=True):
    """z [B,h,w,4] fp32 latent -> image [B,H,W,4] fp32 (VQVAEWrapper.decode, VQVAE.py:186-190)."""
    d = prefix + '.decoder'
    mult, nrb = tuple(ed['ch_mult']), ed['num_res_blocks']
    if quantize:
        _, z = ops.vq_nearest(z, wb.f(prefix + '.quantize.embedding.weight'), scale=scale_factor,
                              want_idx=False)
        if scale_factor != 1.0:      # vq_nearest returns zq / scale; decode wants zq
            z = ops.lincomb(scale_factor, z)
    vec = ops.vec_of(wb.dtype)
    zc = ops.cast2d(z, wb.dtype, cols=3, ldd=vec)
    h = ops.conv2d(zc, wb.w(prefix + '.post_quant_conv.weight'),
                   wb.f(prefix + '.post_quant_conv.bias'), kh=1, kw=1, pad=(0, 0, 0, 0), ldc=vec)
    h = ops.conv2d(h, wb.w(d + '.conv_in.weight'), wb.f(d + '.conv_in.bias'))
    h = _vae_res(wb, d + '.mid.block_1', h)
    h = _vae_attn(wb, d + '.mid.attn_1', h)
    h = _vae_res(wb, d + '.mid.block_2', h)
    for lvl in reversed(range(len(mult))):
        for b in range(nrb + 1):
            h = _vae_res(wb, f'{d}.up.{lvl}.block.{b}', h)
        if lvl != 0:
            h = ops.conv2d(h, wb.w(f'{d}.up.{lvl}.upsample.conv.weight'),
                           wb.f(f'{d}.up.{lvl}.upsample.conv.bias'), ups=True)
    h = ops.group_norm(h, wb.f(d + '.norm_out.weight'), wb.f(d + '.norm_out.bias'), eps=1e-6,
                       act='silu')
    return ops.conv2d(h, wb.w(d + '.conv_out.weight'), wb.f(d + '.conv_out.bias'),
                      out_dtype=torch.float32, ldc=4)
