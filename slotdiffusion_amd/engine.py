"""Forward programs of the hot path on the libsdmi kernels (NHWC, fp32 or bf16 compute).

The networks are executed from the flat checkpoint-key dict (see spec.py / module.py): each
function below is the MI355X-side restatement of one reference `forward`, built from fused
kernel calls instead of nn.Module graphs.  `K` is a kernel provider (kern.Kern for inference,
kern.KernGrad for training -- same program, autograd-recording kernels).  Reference call sites
are cited per function.
"""
import torch

from . import ops, policy, spec

_DEFER_SPLITK = bool(policy.flag('DEFER_SPLITK'))


# ------------------------------------------------------------------------------------------
# a1/a2: ResNet-18(GN) + SoftPositionEmbed + encoder head     (resnet.py:294-312,
#        img_based/models/slot_attention.py:305-316, models/utils.py:60-63)
# ------------------------------------------------------------------------------------------
def resnet_encoder(K, x, plan, prefix='encoder'):
    """x [B,H,W,Cpad] compute dtype -> [B,H/4,W/4,256]."""
    h = K.conv(x, f'{prefix}.conv1.weight')
    h = K.gn(h, f'{prefix}.bn1', eps=1e-5, act='relu')
    for blk, cin, cout, stride, has_ds in plan:
        b = f'{prefix}.{blk}'
        # h feeds conv1 and the identity branch: the branch's gradient is summed in conv1's dgrad
        o, idt = K.conv_fan(h, f'{b}.conv1.weight', stride=stride)
        o = K.gn(o, f'{b}.bn1', eps=1e-5, act='relu')
        o = K.conv(o, f'{b}.conv2.weight')
        if has_ds:
            idt = K.conv(idt, f'{b}.downsample.0.weight', kh=1, kw=1, stride=stride, pad=(0, 0, 0, 0))
            idt = K.gn(idt, f'{b}.downsample.1', eps=1e-5)
        # relu(gn(conv2) + identity) fused in the GN apply kernel
        h = K.gn(o, f'{b}.bn2', eps=1e-5, act='relu', residual=idt)
    return h


def position_embedding(K, name='encoder_pos_embedding'):
    """Linear(4->C)(grid) -> fp32 [h*w, C] (input independent)."""
    wb = K.wb
    key = name + '/pos'
    if not K.training and key in wb.cache:
        return wb.cache[key]
    g = wb.t[f'{name}.grid'].reshape(-1, 4)
    pos = K.linear(g, f'{name}.dense.weight', f'{name}.dense.bias')
    if not K.training:
        wb.cache[key] = pos
    return pos


# ------------------------------------------------------------------------------------------
# frozen DINO ViT encoder (video_based/models/dino.py:21-60 -> transformers ViTModel: patch
# embedding, CLS + learnt positions, pre-LN blocks with biased q/k/v, GELU MLP, final LayerNorm)
# ------------------------------------------------------------------------------------------
def dino_encoder(K, img_nhwc, meta, prefix='encoder.dino'):
    """img [B,H,W,Cpad] compute dtype -> patch features [B, (H/p)*(W/p), hidden] (CLS dropped).
    No autograd: the ViT is frozen (dino.py:39-41).  Attention over the 785 tokens (head dim 64): bf16 ->
    the matrix-core kernel with K/V chunked through LDS (one launch per layer); fp32 -> batched GEMMs
    per head (S = q k^T, row softmax, O = P v)."""
    wb = K.wb
    hid, heads, p_ = meta['hidden'], meta['heads'], meta['patch']
    hd = hid // heads
    dt = img_nhwc.dtype
    with torch.no_grad():
        Kf = K if not K.training else K.wb.model.K()          # frozen: plain launches
        e = prefix + '.embeddings'
        pat = Kf.conv(img_nhwc, e + '.patch_embeddings.projection.weight',
                      e + '.patch_embeddings.projection.bias', kh=p_, kw=p_, stride=p_, pad=(0, 0, 0, 0))
        B, gh, gw, _ = pat.shape
        S = gh * gw + 1
        x = ops.zeros((B, S, hid), dt, pat.device)                                        # token 0 = 0
        ops.cast2d(pat.view(B, gh * gw * hid), dt, out=x.view(B, S * hid)[:, hid:], ldd=S * hid)   # 1..
        key = prefix + '/pos+cls'            # positions with the CLS token folded into row 0 (frozen)
        if key not in wb.cache:
            pc = wb.f(e + '.position_embeddings').view(S, hid).clone()
            ops.lincomb(1.0, pc[0], 1.0, wb.f(e + '.cls_token').view(hid), out=pc[0])
            wb.cache[key] = pc
        x = ops.add_pos(x, wb.cache[key])
        for i in range(meta['layers']):
            l = f'{prefix}.encoder.layer.{i}'
            a = l + '.attention.attention'
            h = Kf.ln(x, l + '.layernorm_before', eps=1e-12)
            qkv = Kf.linear(h, (a + '.query.weight', a + '.key.weight', a + '.value.weight'),
                            (a + '.query.bias', a + '.key.bias', a + '.value.bias'))
            att = ops.attention if (dt == torch.bfloat16 and hd == 64) else ops.attention_long
            ctx = att(qkv[..., :hid], qkv[..., hid:2 * hid], qkv[..., 2 * hid:], heads, head_dim=hd)
            x = Kf.linear(ctx, l + '.attention.output.dense.weight', l + '.attention.output.dense.bias',
                          residual=x)
            h = Kf.ln(x, l + '.layernorm_after', eps=1e-12)
            h = Kf.linear(h, l + '.intermediate.dense.weight', l + '.intermediate.dense.bias', act='gelu')
            x = Kf.linear(h, l + '.output.dense.weight', l + '.output.dense.bias', residual=x)
        x = Kf.ln(x, prefix + '.layernorm', eps=1e-12)
        return x[:, 1:, :].contiguous(), (gh, gw)


def encoder_out(K, img_nhwc, plan):
    """-> tokens [B, h*w, enc_out] in compute dtype."""
    if isinstance(plan, dict):            # DINO ViT features (already [B, h*w, C] tokens)
        feat, (h, w) = dino_encoder(K, img_nhwc, plan)
        B, _, C = feat.shape
    else:
        feat = resnet_encoder(K, img_nhwc, plan)
        B, h, w, C = feat.shape
    tok = K.add_pos(feat.view(B, h * w, C), position_embedding(K))
    tok = K.ln(tok, 'encoder_out_layer.0')
    tok = K.linear(tok, 'encoder_out_layer.1.weight', 'encoder_out_layer.1.bias', act='relu')
    return K.linear(tok, 'encoder_out_layer.3.weight', 'encoder_out_layer.3.bias')


# ------------------------------------------------------------------------------------------
# a3: Slot Attention with mask (img_based/models/sa_diffusion.py:16-70)
# ------------------------------------------------------------------------------------------
def slot_attention(K, tokens, slots_init, iters, eps, name='slot_attention'):
    """tokens [B,M,Cin]; slots_init [N,D] or [B,N,D] fp32 -> slots [B,N,D] fp32, seg [B,M,N]."""
    x = K.ln(tokens, f'{name}.norm_inputs')
    kv = K.linear(x, (f'{name}.project_k.weight', f'{name}.project_v.weight'))     # [B,M,2D]
    return K.slot_attention(kv, slots_init, name, iters, eps)


# ------------------------------------------------------------------------------------------
# a5: SAVi slot transition = nn.TransformerEncoder, norm_first (video_based/models/predictor.py:20-44)
# ------------------------------------------------------------------------------------------
def transformer_predictor(K, x, num_layers, num_heads, name='predictor'):
    """x [B,N,D] fp32 slots -> [B,N,D]; pre-LN blocks, ReLU FFN, head_dim D/heads (48)."""
    D = x.shape[-1]
    hd = D // num_heads
    for i in range(num_layers):
        l = f'{name}.transformer_encoder.layers.{i}'
        n1, xr = K.ln_fan(x, f'{l}.norm1')
        qkv = K.linear(n1, f'{l}.self_attn.in_proj_weight', f'{l}.self_attn.in_proj_bias')
        a = K.attn_self(qkv, num_heads, hd)
        x = K.linear_drop_res(a, f'{l}.self_attn.out_proj.weight', f'{l}.self_attn.out_proj.bias', xr)
        n2, xr = K.ln_fan(x, f'{l}.norm2')
        h = K.dropout(K.linear(n2, f'{l}.linear1.weight', f'{l}.linear1.bias', act='relu'), site='pred')
        x = K.linear_drop_res(h, f'{l}.linear2.weight', f'{l}.linear2.bias', xr)
    return x


# ------------------------------------------------------------------------------------------
# a9-a11: LDM UNet (unet.py:551-576, 271-285; attention.py:297-308, 247-251, 182-206)
# ------------------------------------------------------------------------------------------
class UNetRunner:
    """Block plan + per-call helpers for one UNet."""

    def __init__(self, cfg, prefix='dm_decoder.model.diffusion_model'):
        self.cfg, self.P = cfg, prefix + '.'
        self.plan = spec.unet_plan(cfg)
        self.mc = cfg['model_channels']
        blocks = self.plan['input'] + [self.plan['middle']] + self.plan['output']
        self.res_names = [(l[1], l[3]) for blk in blocks for l in blk if l[0] == 'res']
        self.emb_off = {}
        o = 0
        for n, c in self.res_names:
            self.emb_off[n] = (o, c)
            o += c
        self.emb_total = o
        self.st_names = [l[1] for blk in blocks for l in blk if l[0] == 'st']
        self.heads_of = {l[1]: l[3] for blk in blocks for l in blk if l[0] == 'st'}

    # -- per-call invariants -------------------------------------------------------------
    def time_rowvecs(self, K, t):
        """t [B] fp32 -> fp32 [B, sum(Cout)]: every ResBlock's Linear(SiLU(emb)) in ONE GEMM."""
        P = self.P
        dt = K.wb.dtype
        e = ops.timestep_embedding(t, self.mc)
        e = K.cast(e, dt)
        e = K.linear(e, P + 'time_embed.0.weight', P + 'time_embed.0.bias', act='silu')
        # SiLU(emb) is what every ResBlock consumes -> applied once here
        e = K.linear(e, P + 'time_embed.2.weight', P + 'time_embed.2.bias', act='silu')
        wn = tuple(P + n + '.emb_layers.1.weight' for n, _ in self.res_names)
        bn = tuple(P + n + '.emb_layers.1.bias' for n, _ in self.res_names)
        return K.linear(e, wn, bn, out_dtype=torch.float32)

    def context_kv(self, K, ctx):
        """ctx [B,N,Dc] compute dtype -> {st name: kv [B,N,2C]}; constant across all NFEs."""
        names = [(self.P + n + '.transformer_blocks.0.attn2.to_k.weight',
                  self.P + n + '.transformer_blocks.0.attn2.to_v.weight') for n in self.st_names]
        kvs = dict(zip(self.st_names, K.linear_multi(ctx, names)))
        if hasattr(K, 'cross_prepare') and not K.training:       # inference: fold the slots into the weights
            for n in self.st_names:
                fold = K.cross_prepare(kvs[n], self.P + n + '.transformer_blocks.0', self.heads_of[n])
                if fold is not None:
                    kvs[n] = {'kv': kvs[n], 'fold': fold}
        return kvs

    # -- blocks ---------------------------------------------------------------------------
    def _res(self, K, name, x, rowvecs, want_cat=False):
        """-> (block output, alias of x for the UNet skip-concat or None).  x has up to three
        consumers (GroupNorm, skip branch, skip-concat): the aliases route their gradients into
        the GroupNorm backward kernel."""
        n = self.P + name
        rv = rowvecs[name]                         # strided view; the kernel takes its row pitch
        outs = K.gn_fan(x, n + '.in_layers.0', eps=1e-5, act='silu', n_alias=2 if want_cat else 1,
                        for_conv=n + '.in_layers.2.weight')
        h, skip = outs[0], outs[1]
        h = K.conv(h, n + '.in_layers.2.weight', n + '.in_layers.2.bias', rowvec=rv)
        h = K.gn(h, n + '.out_layers.0', eps=1e-5, act='silu', dropout='unet',      # (dropout: training only)
                 for_conv=n + '.out_layers.3.weight', rowsum_of=rv)
        out = K.res_tail(h, n, skip)
        return out, (outs[2] if want_cat else None)

    def _st(self, K, name, x, heads, kv):
        n = self.P + name
        B, H, W, C = x.shape
        if not K.training and hasattr(K, 'st_fused'):     # bf16 inference: the block as two launches
            out = K.st_fused(x, n, heads, kv)
            if out is not None:
                return out
        if K.training and hasattr(K, 'st_train'):         # bf16 training: fused forward that keeps what backward reads
            out = K.st_train(x, n, heads, kv)
            if out is not None:
                return out
        # (every residual branch takes an alias of the normalised tensor: its gradient is summed
        # inside the norm's backward kernel)
        h, xres = K.gn_fan(x, n + '.norm', eps=1e-6)
        tok = K.linear(h.view(B, H * W, C), n + '.proj_in.weight', n + '.proj_in.bias')
        t = n + '.transformer_blocks.0'
        # self attention
        qkv, tres = K.ln_linear_fan(tok, t + '.norm1', (t + '.attn1.to_q.weight', t + '.attn1.to_k.weight',
                                                        t + '.attn1.to_v.weight'))
        a = K.attn_self(qkv, heads)
        tok = K.linear(a, t + '.attn1.to_out.0.weight', t + '.attn1.to_out.0.bias', residual=tres)
        # slot cross attention (K/V precomputed per sample call)
        tok = K.cross_block(tok, t, kv, heads)
        # GEGLU feed-forward
        g, tres = K.ln_linear_fan(tok, t + '.norm3', t + '.ff.net.0.proj.weight', t + '.ff.net.0.proj.bias',
                                  geglu=True)
        out = K.ff_out_proj(g, tres, xres.view(B, H * W, C), t, n)
        return out.view(B, H, W, C)

    def _run(self, K, layers, h, rowvecs, ctx_kv, want_cat=False):
        """-> (output, alias of the block INPUT for the skip-concat when want_cat, else None)."""
        cat = None
        for i, l in enumerate(layers):
            kind, name = l[0], l[1]
            n = self.P + name
            fan = want_cat and i == 0
            if kind == 'conv':
                h = K.conv(h, n + '.weight', n + '.bias')
            elif kind == 'res':
                h, c = self._res(K, name, h, rowvecs, want_cat=fan)
                cat = c if fan else cat
            elif kind == 'st':
                h = self._st(K, name, h, l[3], ctx_kv[name])
            elif kind == 'down':
                if fan:
                    h, cat = K.conv_fan(h, n + '.op.weight', n + '.op.bias', stride=2)
                else:
                    h = K.conv(h, n + '.op.weight', n + '.op.bias', stride=2)
            elif kind == 'up':      # nearest x2 folded into the conv's gather
                h = K.conv(h, n + '.conv.weight', n + '.conv.bias', ups=True)
        return h, cat

    def forward(self, K, x, rowvecs, ctx_kv, zero_pad=True):
        """x [B,h,w,Cpad] compute dtype -> eps [B,h,w,4] fp32 (3 channels + zero pad; zero_pad=False leaves the pad
        channel unwritten for a reader that takes three channels only)."""
        if not K.training and _DEFER_SPLITK:
            # inference only: a split-K convolution leaves its second stage to the GroupNorm behind it (ops.defer_splitk).
            # (The training forward never defers: autograd keeps tensors alive past the hook that finishes them; the
            # round-4 experiment knob measured 0.03 ms and was removed.)
            with ops.defer_splitk():
                return self._forward(K, x, rowvecs, ctx_kv, zero_pad)
        return self._forward(K, x, rowvecs, ctx_kv, zero_pad)

    def _forward(self, K, x, rowvecs, ctx_kv, zero_pad=True):
        names = [n for n, _ in self.res_names]
        rowvecs = dict(zip(names, K.rowvec_slices(rowvecs, [self.emb_off[n] for n in names])))
        hs = []
        h = x
        # every input-block output has two consumers, the next block and a skip-concat: the next
        # block's first layer hands back an alias of it for the concat (gradients meet in-kernel)
        for i, blk in enumerate(self.plan['input']):
            h, cat = self._run(K, blk, h, rowvecs, ctx_kv, want_cat=i > 0)
            if cat is not None:
                hs[-1] = cat
            hs.append(h)
        h, cat = self._run(K, self.plan['middle'], h, rowvecs, ctx_kv, want_cat=True)
        if cat is not None:
            hs[-1] = cat
        for blk in self.plan['output']:
            h, _ = self._run(K, blk, K.concat(h, hs.pop()), rowvecs, ctx_kv)
        P = self.P
        h = K.gn(h, P + 'out.0', eps=1e-5, act='silu')
        if not zero_pad and not K.training:
            return K.conv(h, P + 'out.2.weight', P + 'out.2.bias', out_dtype=torch.float32, ldc=4, zero_pad=False)
        return K.conv(h, P + 'out.2.weight', P + 'out.2.bias', out_dtype=torch.float32, ldc=4)


# ------------------------------------------------------------------------------------------
# a6/a14/a15: VQ-VAE (VQVAE.py:94-114, 183-194; modules.py:239-261, 338-362) -- frozen, no grad
# ------------------------------------------------------------------------------------------
def _vae_res(K, n, x):
    h, skip = K.gn_fan(x, n + '.norm1', eps=1e-6, act='silu')
    h = K.conv(h, n + '.conv1.weight', n + '.conv1.bias')
    h = K.gn(h, n + '.norm2', eps=1e-6, act='silu')
    if (n + '.nin_shortcut.weight') in K.wb.t:
        skip = K.conv(skip, n + '.nin_shortcut.weight', n + '.nin_shortcut.bias', kh=1, kw=1,
                      pad=(0, 0, 0, 0))
    return K.conv(h, n + '.conv2.weight', n + '.conv2.bias', residual=skip)


def _vae_attn(K, n, x):
    """Single-head attention over h*w tokens with head dim C (modules.py:130-154), as GEMMs:
    S = scale * Q K^T (batched), row softmax, O = P V computed as P @ (V^T)^T."""
    wb = K.wb
    B, H, W, C = x.shape
    S = H * W
    if getattr(K, 'training', False):        # autograd form (VQ-VAE stage-1 training)
        h, xres = K.gn_fan(x, n + '.norm', eps=1e-6)
        qkv = K.linear(h.view(B, S, C), (n + '.q.weight', n + '.k.weight', n + '.v.weight'),
                       (n + '.q.bias', n + '.k.bias', n + '.v.bias'))
        o = K.vae_attn_core(qkv)
        out = K.linear(o, n + '.proj_out.weight', n + '.proj_out.bias', residual=xres.view(B, S, C))
        return out.view(B, H, W, C)
    h = K.gn(x, n + '.norm', eps=1e-6).view(B, S, C)
    qk = K.linear(h, (n + '.q.weight', n + '.k.weight'), (n + '.q.bias', n + '.k.bias'))
    # V^T [B,C,S] = Wv [C,Cin] @ h[b]^T : batched GEMM with the weight as the row operand
    vt = torch.empty((B, C, S), dtype=x.dtype, device=x.device)
    wv = wb.w(n + '.v.weight', x.dtype)
    ops.bmm_nt(wv.unsqueeze(0).expand(B, C, C), h, vt, bias_m=wb.f(n + '.v.bias'))
    sc = torch.empty((B, S, S), dtype=x.dtype, device=x.device)
    ops.bmm_nt(qk[..., :C], qk[..., C:], sc)
    ops.softmax_rows_(sc, scale=float(C) ** -0.5)
    o = torch.empty((B, S, C), dtype=x.dtype, device=x.device)
    ops.bmm_nt(sc, vt, o)
    out = K.linear(o, n + '.proj_out.weight', n + '.proj_out.bias', residual=x.view(B, S, C))
    return out.view(B, H, W, C)


def _pj(prefix, name):
    return f'{prefix}.{name}' if prefix else name


def vae_encode(K, img_nhwc, ed, prefix='dm_decoder.vae.vqvae', scale_factor=1.0):
    """img [B,H,W,Cpad] -> x0 [B,h,w,4] fp32 (3 latent channels + zero pad)."""
    e = _pj(prefix, 'encoder')
    mult, nrb = tuple(ed['ch_mult']), ed['num_res_blocks']
    h = K.conv(img_nhwc, e + '.conv_in.weight', e + '.conv_in.bias')
    for lvl in range(len(mult)):
        for b in range(nrb):
            h = _vae_res(K, f'{e}.down.{lvl}.block.{b}', h)
        if lvl != len(mult) - 1:       # asymmetric (0,1,0,1) zero pad + stride 2, pad handled in-kernel
            h = K.conv(h, f'{e}.down.{lvl}.downsample.conv.weight',
                       f'{e}.down.{lvl}.downsample.conv.bias', stride=2, pad=(0, 1, 0, 1))
    h = _vae_res(K, e + '.mid.block_1', h)
    h = _vae_attn(K, e + '.mid.attn_1', h)
    h = _vae_res(K, e + '.mid.block_2', h)
    h = K.gn(h, e + '.norm_out', eps=1e-6, act='silu')
    vec = ops.vec_of(h.dtype)
    h = K.conv(h, e + '.conv_out.weight', e + '.conv_out.bias', ldc=vec)
    z = K.conv(h, _pj(prefix, 'quant_conv.weight'), _pj(prefix, 'quant_conv.bias'), kh=1, kw=1,
               pad=(0, 0, 0, 0), out_dtype=torch.float32, ldc=4)
    if scale_factor != 1.0:
        z = ops.lincomb(1.0, z, div=scale_factor)
    return z


def vae_decode(K, z, ed, prefix='dm_decoder.vae.vqvae', scale_factor=1.0, quantize=True):
    """z [B,h,w,4] fp32 latent -> image [B,H,W,4] fp32 (VQVAEWrapper.decode, VQVAE.py:186-190)."""
    wb = K.wb
    d = _pj(prefix, 'decoder')
    mult, nrb = tuple(ed['ch_mult']), ed['num_res_blocks']
    if quantize:
        _, z = ops.vq_nearest(z, wb.f(_pj(prefix, 'quantize.embedding.weight')), scale=scale_factor,
                              want_idx=False)
        if scale_factor != 1.0:      # vq_nearest returns zq / scale; decode wants zq
            z = ops.lincomb(scale_factor, z)
    vec = ops.vec_of(wb.dtype)
    zc = K.cast_pad(z, wb.dtype, 3, vec)
    h = K.conv(zc, _pj(prefix, 'post_quant_conv.weight'), _pj(prefix, 'post_quant_conv.bias'), kh=1, kw=1,
               pad=(0, 0, 0, 0), ldc=vec)
    h = K.conv(h, d + '.conv_in.weight', d + '.conv_in.bias')
    h = _vae_res(K, d + '.mid.block_1', h)
    h = _vae_attn(K, d + '.mid.attn_1', h)
    h = _vae_res(K, d + '.mid.block_2', h)
    for lvl in reversed(range(len(mult))):
        for b in range(nrb + 1):
            h = _vae_res(K, f'{d}.up.{lvl}.block.{b}', h)
        if lvl != 0:
            h = K.conv(h, f'{d}.up.{lvl}.upsample.conv.weight', f'{d}.up.{lvl}.upsample.conv.bias',
                       ups=True)
    h = K.gn(h, d + '.norm_out', eps=1e-6, act='silu')
    return K.conv(h, d + '.conv_out.weight', d + '.conv_out.bias', out_dtype=torch.float32, ldc=4)


# ------------------------------------------------------------------------------------------
# a16: plain-SA spatial-broadcast decoder (img_based/models/slot_attention.py:343-364)
# ------------------------------------------------------------------------------------------
def sa_decode(K, slots, dec_plan, dec_resolution, dtype):
    """slots [B,N,D] fp32 -> recon [B,H,W,4] fp32 (channel 3 = 0), masks [B,N,H*W] fp32,
    o [B*N,H,W,ld] (per-slot rgb + alpha logit, compute dtype)."""
    B, N, D = slots.shape
    h, w = dec_resolution
    pos = position_embedding(K, 'decoder_pos_embedding')                 # [h*w, D] fp32
    x = K.broadcast_pos(slots.reshape(B * N, D), pos, dtype).view(B * N, h, w, D)
    for i, (kind, cin, cout, k, stride) in enumerate(dec_plan):
        if kind == 'deconv':
            x = K.deconv(x, f'decoder.{i}.0.weight', f'decoder.{i}.0.bias', k=k, stride=stride,
                         pad=k // 2, act='relu')
        else:
            vec = 8 if dtype == torch.bfloat16 else 4
            x = K.conv(x, f'decoder.{i}.weight', f'decoder.{i}.bias', kh=k, kw=k, stride=1,
                       pad=(k // 2,) * 4, ldc=(cout + vec - 1) // vec * vec)
    recon, masks = K.sa_combine(x, B, N)
    return recon, masks, x
