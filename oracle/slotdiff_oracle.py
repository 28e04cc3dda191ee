"""CPU oracle for the SlotDiffusion hot path -- TEST INFRASTRUCTURE ONLY.

A from-scratch, functional, fp32 torch-CPU restatement of the reference's
algorithm for the path named by BASELINE.json.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file; the product (``slotdiffusion_amd/``) never does.

Parity pinning: ``tests/test_oracle_golden.py`` checks every function here
against tensors captured from the *real* reference (imported with stubs in the
build container by ``tools/gen_golden.py``) and stored under ``tests/golden/``.

All functions take ``W``: a flat ``{checkpoint key: fp32 CPU tensor}`` dict in
the reference's NCHW/[out,in] conventions, so the oracle is independent of the
product's NHWC engine.  Each function cites the reference lines it follows.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------
def _gn(W, name, x, eps, groups=32):
    return F.group_norm(x, groups, W[name + '.weight'], W[name + '.bias'], eps)


def _ln(W, name, x):
    return F.layer_norm(x, (x.shape[-1],), W[name + '.weight'], W[name + '.bias'], 1e-5)


def _lin(W, name, x):
    return F.linear(x, W[name + '.weight'], W.get(name + '.bias'))


def _conv(W, name, x, stride=1, padding=1):
    return F.conv2d(x, W[name + '.weight'], W.get(name + '.bias'), stride=stride, padding=padding)


def _swish(x):
    return x * torch.sigmoid(x)


# ---------------------------------------------------------------------------
# a1/a2: ResNet-18(GN) slot encoder + position embedding + head
# ---------------------------------------------------------------------------
def resnet_encoder(W, img, plan, prefix='encoder'):
    """video_based/models/resnet.py:294-312 (BasicBlock 73-89); GroupNorm eps 1e-5."""
    x = _conv(W, f'{prefix}.conv1', img)
    x = F.relu(_gn(W, f'{prefix}.bn1', x, 1e-5))
    for blk, cin, cout, stride, has_ds in plan:
        b = f'{prefix}.{blk}'
        out = _conv(W, f'{b}.conv1', x, stride=stride)
        out = F.relu(_gn(W, f'{b}.bn1', out, 1e-5))
        out = _conv(W, f'{b}.conv2', out)
        out = _gn(W, f'{b}.bn2', out, 1e-5)
        idt = x
        if has_ds:
            idt = _conv(W, f'{b}.downsample.0', x, stride=stride, padding=0)
            idt = _gn(W, f'{b}.downsample.1', idt, 1e-5)
        x = F.relu(out + idt)
    return x


def dino_vit_forward(W, img, meta, prefix='encoder.dino'):
    """DINOEncoder.forward (video_based/models/dino.py:43-55): transformers' ViTModel (third party,
    pinned 4.27.4 by the reference; the architecture is restated from its published definition and
    pinned by a fixture generated with the installed transformers' ViTModel class on the same weights):
    patch embedding (conv p x p, stride p), [CLS] + learnt positions, pre-LN blocks -- biased q/k/v,
    softmax(q k^T / sqrt(d)) v, output projection, GELU(erf) MLP -- final LayerNorm, eps 1e-12, CLS
    dropped, -> [B, hidden, H/p, W/p]."""
    p, hid, heads = prefix, meta['hidden'], meta['heads']
    hd = hid // heads
    e = p + '.embeddings'
    x = F.conv2d(img, W[e + '.patch_embeddings.projection.weight'], W[e + '.patch_embeddings.projection.bias'],
                 stride=meta['patch'])
    B, _, gh, gw = x.shape
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([W[e + '.cls_token'].expand(B, -1, -1), x], 1) + W[e + '.position_embeddings']
    S = x.shape[1]
    ln = lambda name, t: F.layer_norm(t, (hid,), W[name + '.weight'], W[name + '.bias'], 1e-12)
    for i in range(meta['layers']):
        l = f'{p}.encoder.layer.{i}'
        h = ln(l + '.layernorm_before', x)
        sp = lambda t: t.view(B, S, heads, hd).permute(0, 2, 1, 3)
        q, k, v = (sp(_lin(W, f'{l}.attention.attention.{n}', h)) for n in ('query', 'key', 'value'))
        att = (torch.einsum('bhid,bhjd->bhij', q, k) * hd ** -0.5).softmax(-1)
        ctx = torch.einsum('bhij,bhjd->bhid', att, v).permute(0, 2, 1, 3).reshape(B, S, hid)
        x = x + _lin(W, l + '.attention.output.dense', ctx)
        h = F.gelu(_lin(W, l + '.intermediate.dense', ln(l + '.layernorm_after', x)))
        x = x + _lin(W, l + '.output.dense', h)
    x = ln(p + '.layernorm', x)[:, 1:, :]
    return x.reshape(B, gh, gw, hid).permute(0, 3, 1, 2)


def encoder_out(W, img, plan):
    """img_based/models/slot_attention.py:305-316 + models/utils.py:60-63."""
    feat = dino_vit_forward(W, img, plan) if isinstance(plan, dict) else resnet_encoder(W, img, plan)
    pos = _lin(W, 'encoder_pos_embedding.dense', W['encoder_pos_embedding.grid'])   # [1,h,w,C]
    feat = feat + pos.permute(0, 3, 1, 2)
    x = feat.flatten(2, 3).permute(0, 2, 1)                                        # [B,HW,C]
    x = _ln(W, 'encoder_out_layer.0', x)
    x = F.relu(_lin(W, 'encoder_out_layer.1', x))
    return _lin(W, 'encoder_out_layer.3', x)


# ---------------------------------------------------------------------------
# a3: Slot Attention with mask
# ---------------------------------------------------------------------------
def gru_cell(W, name, x, h):
    """torch.nn.GRUCell semantics: gate order (r, z, n)."""
    gi = F.linear(x, W[name + '.weight_ih'], W[name + '.bias_ih'])
    gh = F.linear(h, W[name + '.weight_hh'], W[name + '.bias_hh'])
    i_r, i_z, i_n = gi.chunk(3, -1)
    h_r, h_z, h_n = gh.chunk(3, -1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1. - z) * n + z * h


def slot_attention(W, x, slots, num_iterations, eps=1e-6, name='slot_attention'):
    """img_based/models/sa_diffusion.py:16-70.

    x [B,M,Cin], slots [B,N,D] -> slots [B,N,D], seg_mask [B,N,M] (softmax over
    slots of the LAST iteration, before +eps / token renormalisation).
    """
    B, N, D = slots.shape
    x = _ln(W, f'{name}.norm_inputs', x)
    k = _lin(W, f'{name}.project_k', x)
    v = _lin(W, f'{name}.project_v', x)
    scale = D ** -0.5
    seg = None
    for it in range(num_iterations):
        prev = slots
        q = _lin(W, f'{name}.project_q.1', _ln(W, f'{name}.project_q.0', slots))
        logits = scale * torch.einsum('bmc,bnc->bmn', k, q)
        attn = F.softmax(logits, dim=-1)                       # over slots
        if it == num_iterations - 1:
            seg = attn.detach().clone().permute(0, 2, 1)
        attn = attn + eps
        attn = attn / attn.sum(dim=1, keepdim=True)            # over tokens
        upd = torch.einsum('bmn,bmc->bnc', attn, v)
        slots = gru_cell(W, f'{name}.gru', upd.reshape(B * N, D), prev.reshape(B * N, D))
        slots = slots.view(B, N, D)
        h = F.relu(_lin(W, f'{name}.mlp.1', _ln(W, f'{name}.mlp.0', slots)))
        slots = slots + _lin(W, f'{name}.mlp.3', h)
    return slots, seg


def sa_encode(W, img, plan, num_iterations, training, eps=1e-6):
    """img_based/models/sa_diffusion.py:155-183. Returns slots [B,N,D], masks [B,N,h,w]."""
    B, _, H, Wd = img.shape
    x = encoder_out(W, img, plan)
    hw = W['encoder_pos_embedding.grid'].shape[1:3]
    init = W['init_latents'].repeat(B, 1, 1)
    slots, masks = slot_attention(W, x, init, num_iterations, eps)
    N = slots.shape[1]
    masks = masks.unflatten(-1, tuple(hw))
    if not training and tuple(hw) != (H, Wd):
        m = masks.flatten(0, 1).unsqueeze(1)
        m = F.interpolate(m, (H, Wd), mode='bilinear', align_corners=False)
        masks = m.squeeze(1).unflatten(0, (B, N))
    return slots, masks



# ---------------------------------------------------------------------------
# a4/a5: video model -- transformer slot predictor + recurrence over frames
# ---------------------------------------------------------------------------
def transformer_predictor(W, x, num_layers, num_heads, name='predictor'):
    """video_based/models/predictor.py:20-44: nn.TransformerEncoder, norm_first=True, ReLU FFN,
    no final norm; dropout off (eval / parity).  x [B,N,D]."""
    B, N, D = x.shape
    hd = D // num_heads
    for i in range(num_layers):
        l = f'{name}.transformer_encoder.layers.{i}'
        h = _ln(W, f'{l}.norm1', x)
        qkv = F.linear(h, W[f'{l}.self_attn.in_proj_weight'], W[f'{l}.self_attn.in_proj_bias'])
        q, k, v = qkv.chunk(3, -1)
        sp = lambda t: t.view(B, N, num_heads, hd).permute(0, 2, 1, 3)
        att = (torch.einsum('bhid,bhjd->bhij', sp(q), sp(k)) * hd ** -0.5).softmax(-1)
        a = torch.einsum('bhij,bhjd->bhid', att, sp(v)).permute(0, 2, 1, 3).reshape(B, N, D)
        x = x + _lin(W, f'{l}.self_attn.out_proj', a)
        h = F.relu(_lin(W, f'{l}.linear1', _ln(W, f'{l}.norm2', x)))
        x = x + _lin(W, f'{l}.linear2', h)
    return x


def savi_encode(W, img, plan, num_iterations, training, pred_layers, pred_heads, eps=1e-6):
    """video_based/models/savi_diffusion.py:169-216.  img [B,T,3,H,W] ->
    slots [B,T,N,D], masks [B,T,N,h,w] (train) / [B,T,N,H,W] (eval)."""
    B, T, _, H, Wd = img.shape
    x = encoder_out(W, img.flatten(0, 1), plan).unflatten(0, (B, T))
    hw = W['encoder_pos_embedding.grid'].shape[1:3]
    init = W['init_latents'].repeat(B, 1, 1)
    prev, all_s, all_m = None, [], []
    for t in range(T):
        lat = init if prev is None else transformer_predictor(W, prev, pred_layers, pred_heads)
        s, m = slot_attention(W, x[:, t], lat, num_iterations, eps)
        all_s.append(s)
        all_m.append(m)
        prev = s
    slots = torch.stack(all_s, 1)
    masks = torch.stack(all_m, 1).unflatten(-1, tuple(hw))
    N = slots.shape[2]
    if not training and tuple(hw) != (H, Wd):
        m = masks.flatten(0, 2).unsqueeze(1)
        m = F.interpolate(m, (H, Wd), mode='bilinear', align_corners=False)
        masks = m.squeeze(1).unflatten(0, (B, T, N))
    return slots, masks

# ---------------------------------------------------------------------------
# a9-a11: LDM UNet
# ---------------------------------------------------------------------------
def timestep_embedding(t, dim, max_period=10000):
    """unet/utils.py:70-92; [cos | sin], fractional t allowed."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _attention(q, k, v, heads):
    """attention.py:182-206: softmax(q k^T * d^-0.5) v, heads split on channels."""
    B, S, C = q.shape
    d = C // heads
    qh = q.view(B, S, heads, d).permute(0, 2, 1, 3)
    kh = k.view(B, k.shape[1], heads, d).permute(0, 2, 1, 3)
    vh = v.view(B, v.shape[1], heads, d).permute(0, 2, 1, 3)
    sim = torch.einsum('bhid,bhjd->bhij', qh, kh) * (d ** -0.5)
    att = sim.softmax(dim=-1)
    out = torch.einsum('bhij,bhjd->bhid', att, vh)
    return out.permute(0, 2, 1, 3).reshape(B, S, C)


def _spatial_transformer(W, name, x, ctx, heads):
    """attention.py:297-308 / 247-251; GroupNorm eps 1e-6; exact-erf GELU in GEGLU."""
    B, C, H, Wd = x.shape
    h = _gn(W, f'{name}.norm', x, 1e-6)
    h = _conv(W, f'{name}.proj_in', h, padding=0)
    tok = h.flatten(2).permute(0, 2, 1)
    t = f'{name}.transformer_blocks.0'
    n1 = _ln(W, f'{t}.norm1', tok)
    a = _attention(_lin(W, f'{t}.attn1.to_q', n1), _lin(W, f'{t}.attn1.to_k', n1),
                   _lin(W, f'{t}.attn1.to_v', n1), heads)
    tok = _lin(W, f'{t}.attn1.to_out.0', a) + tok
    n2 = _ln(W, f'{t}.norm2', tok)
    a = _attention(_lin(W, f'{t}.attn2.to_q', n2), _lin(W, f'{t}.attn2.to_k', ctx),
                   _lin(W, f'{t}.attn2.to_v', ctx), heads)
    tok = _lin(W, f'{t}.attn2.to_out.0', a) + tok
    n3 = _ln(W, f'{t}.norm3', tok)
    xg, gate = _lin(W, f'{t}.ff.net.0.proj', n3).chunk(2, dim=-1)
    tok = _lin(W, f'{t}.ff.net.2', xg * F.gelu(gate)) + tok
    h = tok.permute(0, 2, 1).reshape(B, C, H, Wd)
    return _conv(W, f'{name}.proj_out', h, padding=0) + x


def _resblock(W, name, x, emb, dropout_mask=None):
    """unet.py:271-285; GroupNorm32 eps 1e-5; dropout applied as an explicit mask (or off)."""
    h = _conv(W, f'{name}.in_layers.2', F.silu(_gn(W, f'{name}.in_layers.0', x, 1e-5)))
    h = h + _lin(W, f'{name}.emb_layers.1', F.silu(emb))[:, :, None, None]
    h = F.silu(_gn(W, f'{name}.out_layers.0', h, 1e-5))
    if dropout_mask is not None:
        h = h * dropout_mask
    h = _conv(W, f'{name}.out_layers.3', h)
    if f'{name}.skip_connection.weight' in W:
        x = _conv(W, f'{name}.skip_connection', x, padding=0)
    return x + h


def unet_forward(W, plan, x, t, ctx, prefix='dm_decoder.model.diffusion_model', mc=128):
    """unet.py:551-576 driven by the block plan (spec.unet_plan)."""
    P = prefix + '.'
    emb = timestep_embedding(t, mc)
    emb = _lin(W, P + 'time_embed.2', F.silu(_lin(W, P + 'time_embed.0', emb)))

    def run(layers, h):
        for l in layers:
            kind, name = l[0], P + l[1]
            if kind == 'conv':
                h = _conv(W, name, h)
            elif kind == 'res':
                h = _resblock(W, name, h, emb)
            elif kind == 'st':
                h = _spatial_transformer(W, name, h, ctx, l[3])
            elif kind == 'down':
                h = _conv(W, name + '.op', h, stride=2)
            elif kind == 'up':
                h = _conv(W, name + '.conv', F.interpolate(h, scale_factor=2, mode='nearest'))
        return h

    hs = []
    h = x
    for blk in plan['input']:
        h = run(blk, h)
        hs.append(h)
    h = run(plan['middle'], h)
    for blk in plan['output']:
        h = run(blk, torch.cat([h, hs.pop()], dim=1))
    h = F.silu(_gn(W, P + 'out.0', h, 1e-5))
    return _conv(W, P + 'out.2', h)


# ---------------------------------------------------------------------------
# a6/a14/a15: VQ-VAE encode / quantize / decode
# ---------------------------------------------------------------------------
def _pj(prefix, name):
    return f'{prefix}.{name}' if prefix else name


def _vae_res(W, name, x):
    """vqvae/modules.py:95-110 (GroupNorm eps 1e-6, swish, dropout 0)."""
    h = _conv(W, f'{name}.conv1', _swish(_gn(W, f'{name}.norm1', x, 1e-6)))
    h = _conv(W, f'{name}.conv2', _swish(_gn(W, f'{name}.norm2', h, 1e-6)))
    if f'{name}.nin_shortcut.weight' in W:
        x = _conv(W, f'{name}.nin_shortcut', x, padding=0)
    return x + h


def _vae_attn(W, name, x):
    """vqvae/modules.py:130-154: single-head attention over h*w tokens, scale C^-0.5."""
    B, C, H, Wd = x.shape
    h = _gn(W, f'{name}.norm', x, 1e-6)
    q = _conv(W, f'{name}.q', h, padding=0).flatten(2).permute(0, 2, 1)     # [B,S,C]
    k = _conv(W, f'{name}.k', h, padding=0).flatten(2)                      # [B,C,S]
    v = _conv(W, f'{name}.v', h, padding=0).flatten(2)                      # [B,C,S]
    w = torch.bmm(q, k) * (int(C) ** -0.5)
    w = F.softmax(w, dim=2)
    o = torch.bmm(v, w.permute(0, 2, 1)).reshape(B, C, H, Wd)
    return x + _conv(W, f'{name}.proj_out', o, padding=0)


def vae_encode(W, img, ed, prefix='dm_decoder.vae.vqvae', scale_factor=1.0):
    """VQVAE.py:94-99, 183-184; modules.py:239-261; asymmetric (0,1,0,1) pad before stride-2."""
    e = _pj(prefix, 'encoder')
    mult, nrb = tuple(ed['ch_mult']), ed['num_res_blocks']
    h = _conv(W, f'{e}.conv_in', img)
    for lvl in range(len(mult)):
        for b in range(nrb):
            h = _vae_res(W, f'{e}.down.{lvl}.block.{b}', h)
        if lvl != len(mult) - 1:
            h = _conv(W, f'{e}.down.{lvl}.downsample.conv', F.pad(h, (0, 1, 0, 1)), stride=2,
                      padding=0)
    h = _vae_res(W, f'{e}.mid.block_1', h)
    h = _vae_attn(W, f'{e}.mid.attn_1', h)
    h = _vae_res(W, f'{e}.mid.block_2', h)
    h = _conv(W, f'{e}.conv_out', _swish(_gn(W, f'{e}.norm_out', h, 1e-6)))
    return _conv(W, _pj(prefix, 'quant_conv'), h, padding=0) / scale_factor


def vq_nearest(W, z, prefix='dm_decoder.vae.vqvae'):
    """quantize.py:85-94: expanded-square distance, argmin (first minimum) -> int64 [B,h,w]."""
    emb = W[_pj(prefix, 'quantize.embedding.weight')]
    zf = z.permute(0, 2, 3, 1).contiguous().view(-1, emb.shape[1])
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1) - \
        2 * torch.einsum('bd,dn->bn', zf, emb.t())
    idx = torch.argmin(d, dim=1)
    return idx.view(z.shape[0], z.shape[2], z.shape[3])


def vq_quantize(W, z, prefix='dm_decoder.vae.vqvae', scale_factor=1.0):
    """VQVAE.py:192-194: z*s -> nearest code -> /s.  Returns (z_q [B,C,h,w], idx)."""
    zs = z * scale_factor
    idx = vq_nearest(W, zs, prefix)
    emb = W[_pj(prefix, 'quantize.embedding.weight')]
    zq = emb[idx].permute(0, 3, 1, 2).contiguous()
    zq = zs + (zq - zs).detach()  # straight-through form (quantize.py:107) -- keeps its rounding
    return zq / scale_factor, idx


def vae_decode_quant(W, zq, ed, prefix='dm_decoder.vae.vqvae'):
    """VQVAE.py:110-114; modules.py:338-362 (input already quantized)."""
    d = _pj(prefix, 'decoder')
    mult, nrb = tuple(ed['ch_mult']), ed['num_res_blocks']
    h = _conv(W, _pj(prefix, 'post_quant_conv'), zq, padding=0)
    h = _conv(W, f'{d}.conv_in', h)
    h = _vae_res(W, f'{d}.mid.block_1', h)
    h = _vae_attn(W, f'{d}.mid.attn_1', h)
    h = _vae_res(W, f'{d}.mid.block_2', h)
    for lvl in reversed(range(len(mult))):
        for b in range(nrb + 1):
            h = _vae_res(W, f'{d}.up.{lvl}.block.{b}', h)
        if lvl != 0:
            h = _conv(W, f'{d}.up.{lvl}.upsample.conv',
                      F.interpolate(h, scale_factor=2.0, mode='nearest'))
    return _conv(W, f'{d}.conv_out', _swish(_gn(W, f'{d}.norm_out', h, 1e-6)))


def vae_decode(W, z, ed, prefix='dm_decoder.vae.vqvae', scale_factor=1.0):
    """VQVAEWrapper.decode(quantize=True) (VQVAE.py:186-190)."""
    zq, _ = vq_quantize(W, z * scale_factor, prefix, 1.0)
    return vae_decode_quant(W, zq, ed, prefix)


def vqvae_forward(W, img, ed, prefix='', beta=0.25, percept_loss_w=0.):
    """Stand-alone VQVAE.forward + calc_eval_loss (VQVAE.py:116-146, quantize.py:80-123,
    loss.py:19-46 without the LPIPS term): -> dict(recon, token_id, quant_loss, recon_loss,
    recon_mse)."""
    z = vae_encode(W, img, ed, prefix)
    zq, idx = vq_quantize(W, z, prefix)
    emb = W[_pj(prefix, 'quantize.embedding.weight')]
    zq_raw = emb[idx].permute(0, 3, 1, 2)
    quant_loss = torch.mean((zq_raw.detach() - z) ** 2) + beta * torch.mean((zq_raw - z.detach()) ** 2)
    recon = vae_decode_quant(W, zq, ed, prefix)
    recon_loss = torch.abs(img - recon).mean() if percept_loss_w > 0 else F.mse_loss(img, recon)
    return dict(recon=recon, token_id=idx, quant_loss=quant_loss, recon_loss=recon_loss,
                recon_mse=F.mse_loss(recon, img))


# ---------------------------------------------------------------------------
# a7/a8: q-sample and the denoising loss (explicit t and noise)
# ---------------------------------------------------------------------------
def q_sample(W, x0, t, noise, prefix='dm_decoder'):
    """ddpm.py:161-165."""
    a = W[f'{prefix}.sqrt_alphas_bar'][t].view(-1, 1, 1, 1)
    s = W[f'{prefix}.sqrt_one_minus_alphas_bar'][t].view(-1, 1, 1, 1)
    return a * x0 + s * noise


def ldm_loss(W, plan, ed, img, slots, t, noise, mc=128, pred_target='eps'):
    """ldm.py:59-83 with t / noise supplied by the caller; target = noise ('eps'), x0 ('x0') or
    v = alpha_t * noise - sigma_t * x0 ('v', video_based/models/ddpm/ldm.py:74-78)."""
    with torch.no_grad():
        x0 = vae_encode(W, img, ed)
    xt = q_sample(W, x0, t, noise)
    pred = unet_forward(W, plan, xt, t, slots, mc=mc)
    if pred_target == 'eps':
        gt = noise
    elif pred_target == 'v':
        a = W['dm_decoder.sqrt_alphas_bar'][t].view(-1, 1, 1, 1)
        sg = W['dm_decoder.sqrt_one_minus_alphas_bar'][t].view(-1, 1, 1, 1)
        gt = a * noise - sg * x0
    else:
        gt = x0
    return F.mse_loss(pred, gt), pred, x0


# ---------------------------------------------------------------------------
# a16: plain-SA spatial-broadcast CNN decoder + reconstruction loss
# ---------------------------------------------------------------------------
def sa_decode(W, slots, dec_plan, dec_resolution):
    """img_based/models/slot_attention.py:343-364.  dec_plan = spec.sa_decoder_plan(...).
    -> recon [B,3,H,W], recons [B,N,3,H,W], masks [B,N,1,H,W].  The transposed convs follow nerv's
    deconv_norm_act (padding k//2, output_padding stride-1, ReLU; tools/ref_harness.py)."""
    B, N, D = slots.shape
    h, w = dec_resolution
    x = slots.reshape(B * N, D, 1, 1).repeat(1, 1, h, w)
    pos = _lin(W, 'decoder_pos_embedding.dense', W['decoder_pos_embedding.grid'])
    x = x + pos.permute(0, 3, 1, 2)
    for i, (kind, cin, cout, k, stride) in enumerate(dec_plan):
        if kind == 'deconv':
            x = F.relu(F.conv_transpose2d(x, W[f'decoder.{i}.0.weight'], W[f'decoder.{i}.0.bias'],
                                          stride=stride, padding=k // 2, output_padding=stride - 1))
        else:
            x = F.conv2d(x, W[f'decoder.{i}.weight'], W[f'decoder.{i}.bias'])
    H, Wd = x.shape[-2:]
    out = x.view(B, N, 4, H, Wd)
    recons = out[:, :, :3]
    masks = F.softmax(out[:, :, 3:], dim=1)
    return (recons * masks).sum(1), recons, masks


def sa_forward_loss(W, img, plan, dec_plan, dec_resolution, num_iterations, eps=1e-6):
    """SA.forward + calc_train_loss (slot_attention.py:318-375): -> loss, recon, masks, slots."""
    slots, _ = sa_encode(W, img, plan, num_iterations, training=True, eps=eps)
    recon, recons, masks = sa_decode(W, slots, dec_plan, dec_resolution)
    return F.mse_loss(recon, img), recon, masks, slots


def savi_forward_loss(W, img, plan, dec_plan, dec_resolution, num_iterations, pred_layers, pred_heads,
                      eps=1e-6):
    """SAVi._forward + calc_train_loss (video_based/models/savi.py:445-508): img [B,T,3,H,W] ->
    loss, recon [B,T,3,H,W], masks [B,T,N,1,H,W], slots [B,T,N,D]."""
    B, T = img.shape[:2]
    slots, _ = savi_encode(W, img, plan, num_iterations, True, pred_layers, pred_heads, eps)
    recon, recons, masks = sa_decode(W, slots.flatten(0, 1), dec_plan, dec_resolution)
    recon, masks = recon.unflatten(0, (B, T)), masks.unflatten(0, (B, T))
    return F.mse_loss(recon, img), recon, masks, slots


# ---------------------------------------------------------------------------
# a12/a13: DPM-Solver++ (singlestep, order 3, time_uniform) on the discrete schedule
# ---------------------------------------------------------------------------
class NoiseScheduleDiscrete:
    """dpm_solver.py:160-235 ('discrete' branch), fp32 like the reference."""

    def __init__(self, betas):
        self.log_alpha = (0.5 * torch.log(1 - betas).cumsum(dim=0)).float()       # [N]
        self.total_N = len(self.log_alpha)
        self.t_array = torch.linspace(0., 1., self.total_N + 1)[1:].float()
        self.T = 1.

    @staticmethod
    def _interp(x, xp, yp):
        """Piecewise-linear f(x) through (xp, yp), linear extrapolation outside (11-50)."""
        K = xp.shape[0]
        idx = torch.searchsorted(xp, x, right=False)          # #{xp < x}
        lo = torch.where(idx == 0, torch.zeros_like(idx),
                         torch.where(idx == K, torch.full_like(idx, K - 2), idx - 1))
        x0, x1, y0, y1 = xp[lo], xp[lo + 1], yp[lo], yp[lo + 1]
        return y0 + (x - x0) * (y1 - y0) / (x1 - x0)

    def log_mean_coeff(self, t):
        return self._interp(t.reshape(-1), self.t_array, self.log_alpha)

    def alpha(self, t):
        return torch.exp(self.log_mean_coeff(t))

    def std(self, t):
        return torch.sqrt(1. - torch.exp(2. * self.log_mean_coeff(t)))

    def lam(self, t):
        lm = self.log_mean_coeff(t)
        return lm - 0.5 * torch.log(1. - torch.exp(2. * lm))

    def inverse_lambda(self, lamb):
        la = -0.5 * torch.logaddexp(torch.zeros((1,)), -2. * lamb)
        return self._interp(la.reshape(-1), torch.flip(self.log_alpha, [0]),
                            torch.flip(self.t_array, [0]))


def dpm_orders_and_timesteps(steps, order, t_T, t_0):
    """dpm_solver.py:574-631 (order 3, skip_type 'time_uniform')."""
    assert order == 3
    K = steps // 3 + 1
    if steps % 3 == 0:
        orders = [3] * (K - 2) + [2, 1]
    elif steps % 3 == 1:
        orders = [3] * (K - 1) + [1]
    else:
        orders = [3] * (K - 1) + [2]
    ts = torch.linspace(t_T, t_0, steps + 1)
    outer = ts[torch.cumsum(torch.tensor([0] + orders), 0)]
    return outer, orders


def dpm_step_coeffs(ns, s, t, order):
    """Scalar coefficients of one singlestep DPM-Solver++ update (dpm_solver.py:639-831,
    'dpmsolver++' / solver_type 'dpmsolver' branches; r1, r2 from 1319-1323)."""
    inner = torch.linspace(s.item(), t.item(), order + 1)
    lam_in = ns.lam(inner)
    h_in = lam_in[-1] - lam_in[0]
    r1 = None if order <= 1 else (lam_in[1] - lam_in[0]) / h_in
    r2 = None if order <= 2 else (lam_in[2] - lam_in[0]) / h_in
    s1d, t1d = s.reshape(1), t.reshape(1)
    lam_s, lam_t = ns.lam(s1d), ns.lam(t1d)
    h = lam_t - lam_s
    c = dict(order=order, s=s1d, t=t1d, h=h, sigma_s=ns.std(s1d), sigma_t=ns.std(t1d),
             alpha_t=torch.exp(ns.log_mean_coeff(t1d)), phi_1=torch.expm1(-h))
    if order >= 2:
        if r1 is None:
            r1 = 0.5
        s1 = ns.inverse_lambda(lam_s + r1 * h)
        c.update(r1=r1, s1=s1, sigma_s1=ns.std(s1), alpha_s1=torch.exp(ns.log_mean_coeff(s1)),
                 phi_11=torch.expm1(-r1 * h))
    if order >= 3:
        s2 = ns.inverse_lambda(lam_s + r2 * h)
        c.update(r2=r2, s2=s2, sigma_s2=ns.std(s2), alpha_s2=torch.exp(ns.log_mean_coeff(s2)),
                 phi_12=torch.expm1(-r2 * h), phi_22=torch.expm1(-r2 * h) / (r2 * h) + 1.,
                 phi_2=torch.expm1(-h) / h + 1.)
    return c


def dpm_solver_sample(eps_fn, quantize_fn, betas, x, steps=20, order=3, trace=None,
                      model_type='noise'):
    """DPM_Solver.sample(method='singlestep', skip 'time_uniform') (dpm_solver.py:1310-1328).

    eps_fn(x, t_input[B]) -> eps prediction with t_input = (t - 1/N) * 1000 (345-346);
    quantize_fn(x0) -> VQ-denoised x0 (523-534, vq_denoised=True).
    """
    ns = NoiseScheduleDiscrete(betas)
    B = x.shape[0]
    t_0, t_T = 1. / ns.total_N, ns.T
    outer, orders = dpm_orders_and_timesteps(steps, order, t_T, t_0)

    def data_pred(xc, tc):
        t_in = (tc.expand(B) - 1. / ns.total_N) * 1000.
        eps = eps_fn(xc, t_in)
        if model_type == 'x_start':          # model_wrapper.noise_pred_fn (dpm_solver.py:358-361)
            eps = (xc - ns.alpha(tc) * eps) / ns.std(tc)
        elif model_type == 'v':              # dpm_solver.py:362-365
            eps = ns.alpha(tc) * eps + ns.std(tc) * xc
        x0 = (xc - ns.std(tc) * eps) / ns.alpha(tc)
        return quantize_fn(x0)

    for i, od in enumerate(orders):
        c = dpm_step_coeffs(ns, outer[i], outer[i + 1], od)
        m_s = data_pred(x, c['s'])
        if od == 1:
            x = c['sigma_t'] / c['sigma_s'] * x - c['alpha_t'] * c['phi_1'] * m_s
        elif od == 2:
            x_s1 = (c['sigma_s1'] / c['sigma_s']) * x - (c['alpha_s1'] * c['phi_11']) * m_s
            m_s1 = data_pred(x_s1, c['s1'])
            x = (c['sigma_t'] / c['sigma_s']) * x - (c['alpha_t'] * c['phi_1']) * m_s \
                - (0.5 / c['r1']) * (c['alpha_t'] * c['phi_1']) * (m_s1 - m_s)
        else:
            x_s1 = (c['sigma_s1'] / c['sigma_s']) * x - (c['alpha_s1'] * c['phi_11']) * m_s
            m_s1 = data_pred(x_s1, c['s1'])
            x_s2 = (c['sigma_s2'] / c['sigma_s']) * x - (c['alpha_s2'] * c['phi_12']) * m_s \
                + c['r2'] / c['r1'] * (c['alpha_s2'] * c['phi_22']) * (m_s1 - m_s)
            m_s2 = data_pred(x_s2, c['s2'])
            x = (c['sigma_t'] / c['sigma_s']) * x - (c['alpha_t'] * c['phi_1']) * m_s \
                + (1. / c['r2']) * (c['alpha_t'] * c['phi_2']) * (m_s2 - m_s)
        if trace is not None:
            trace.append(x.clone())
    return x


def ldm_sample(W, plan, ed, slots, x_T, steps=20, mc=128, trace=None):
    """CondDDPM.generate_imgs(use_dpm=True) + LDM.log_images decode (cond_ddpm.py:155-193,
    ldm.py:122-129). Returns (latent x_0 [B,3,h,w], decoded samples [B,3,H,W])."""
    betas = W['dm_decoder.betas']
    eps_fn = lambda xc, t_in: unet_forward(W, plan, xc, t_in, slots, mc=mc)
    q_fn = lambda x0: vq_quantize(W, x0)[0]
    x = dpm_solver_sample(eps_fn, q_fn, betas, x_T, steps=steps, order=3, trace=trace)
    return x, vae_decode(W, x, ed)


# ---------------------------------------------------------------------------
# 8(f) row 2: ancestral sampler (img_based/models/ddpm/cond_ddpm.py:55-132, ddpm.py:167-180)
# ---------------------------------------------------------------------------
def p_mean(W, model_out, x, t, quantize_fn, pred_target='eps', clip_denoised=False,
           prefix='dm_decoder'):
    """_p_mean_variance: x0 estimate (from eps, or the model output itself for 'x0'), optional
    clamp, VQ denoising, then the posterior mean / log-variance of q(x_{t-1} | x_t, x0)."""
    g = lambda k: W[f'{prefix}.{k}'][t].view(-1, 1, 1, 1)
    if pred_target == 'eps':
        x0 = g('sqrt_recip_alphas_bar') * x - g('sqrt_recipm1_alphas_bar') * model_out
    elif pred_target == 'v':                 # video_based/models/ddpm/cond_ddpm.py:63-67
        x0 = g('sqrt_alphas_bar') * x - g('sqrt_one_minus_alphas_bar') * model_out
    else:
        x0 = model_out
    if clip_denoised:
        x0 = x0.clamp(-1., 1.)
    if quantize_fn is not None:
        x0 = quantize_fn(x0)
    mean = g('posterior_mean_coef1') * x0 + g('posterior_mean_coef2') * x
    return mean, g('posterior_log_variance_clipped')


def p_sample(W, model_fn, x, t, noise, quantize_fn, pred_target='eps', clip_denoised=False):
    """_p_sample: x_{t-1} = mean + [t > 0] * exp(0.5 * logvar) * noise."""
    mean, logvar = p_mean(W, model_fn(x, t), x, t, quantize_fn, pred_target, clip_denoised)
    mask = (1 - (t == 0).float()).view(-1, 1, 1, 1)
    return mean + mask * (0.5 * logvar).exp() * noise


def ancestral_sample(W, model_fn, quantize_fn, x, noise_fn, num_timesteps, log_every_t=100,
                     pred_target='eps'):
    """_sample_x0_from_noise: t = T-1 .. 0; intermediates = [x_T] + states at i % log_every_t == 0
    or i == T-1."""
    inter = [x]
    B = x.shape[0]
    for i in reversed(range(num_timesteps)):
        t = torch.full((B,), i, dtype=torch.long)
        x = p_sample(W, model_fn, x, t, noise_fn(x.shape), quantize_fn, pred_target)
        if i % log_every_t == 0 or i == num_timesteps - 1:
            inter.append(x)
    return x, torch.stack(inter, 0)


# ---------------------------------------------------------------------------
# 8(f) row 2: DDIM sampler (video_based/models/ddpm/ddim.py:90-218; utils.py:50-97)
# ---------------------------------------------------------------------------
def ddim_schedule(alphas_bar, steps, eta=0.):
    """uniform discretisation: timesteps = range(0, T, T // steps) + 1; a_t, a_prev, sigma_t."""
    T = alphas_bar.shape[0]
    ts = torch.arange(0, T, T // steps) + 1
    a = alphas_bar[ts]
    a_prev = torch.cat([alphas_bar[:1], alphas_bar[ts[:-1]]])
    sig = eta * torch.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
    return ts, a, a_prev, sig


def ddim_sample(eps_fn, quantize_fn, alphas_bar, x, steps, eta=0., log_every_t=100, noise_fn=None):
    """DDIMSampler._sample_x0_from_noise / _p_sample_ddim for pred_target='eps', vq_denoised.
    Returns (x_0, intermediates as the reference logs them)."""
    ts, a, a_prev, sig = ddim_schedule(alphas_bar, steps, eta)
    som = torch.sqrt(1. - a)
    B = x.shape[0]
    n = ts.shape[0]
    inter = [x]
    for i, step in enumerate(torch.flip(ts, (0,))):
        index = n - i - 1
        t = torch.full((B,), int(step), dtype=torch.long)
        e_t = eps_fn(x, t)
        pred_x0 = (x - som[index] * e_t) / a[index].sqrt()
        pred_x0 = quantize_fn(pred_x0)
        dir_xt = (1. - a_prev[index] - sig[index] ** 2).sqrt() * e_t
        noise = sig[index] * (noise_fn(x.shape) if noise_fn is not None else torch.zeros_like(x))
        x = a_prev[index].sqrt() * pred_x0 + dir_xt + noise
        if index % log_every_t == 0 or index == n - 1:
            inter.append(x)
    return x, torch.stack(inter, 0)


# ---------------------------------------------------------------------------
# a17: optimiser step restated (Adam, two groups, global-norm clip)
# ---------------------------------------------------------------------------
def clip_and_adam(params, grads, m, v, step, lrs, clip, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) after clip_grad_norm_(max_norm=clip)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(clip / (total + 1e-6), max=1.0)
    for p, g, mi, vi, lr in zip(params, grads, m, v, lrs):
        g = g * coef
        mi.mul_(b1).add_(g, alpha=1 - b1)
        vi.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (vi.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(mi, denom, value=-lr / bc1)
    return total


# ---------------------------------------------------------------------------
# metrics used by the acceptance checks
# ---------------------------------------------------------------------------
def ema_update(shadow, params, num_updates, decay=0.9999):
    """LitEma.forward (ddpm/ema.py:29-52): returns the new num_updates; shadows updated in place.
    The reference keeps `decay` as a float32 buffer and `num_updates` as an int tensor, so the warm-up
    ratio and 1 - decay are float32 values; restated the same way (pinned by tests/golden/ema_lit.npz)."""
    d = torch.tensor(decay, dtype=torch.float32)
    if num_updates >= 0:
        num_updates += 1
        n = torch.tensor(num_updates, dtype=torch.int32)
        d = torch.minimum(d, (1 + n) / (10 + n))
    omd = 1.0 - d
    for s, p in zip(shadow, params):
        s.sub_(omd * (s - p))
    return num_updates


def psnr(a, b):
    """10*log10(1/mse) on [0,1]-mapped images (eval_utils.py:95-101 via skimage, data_range=1)."""
    a01, b01 = (a * 0.5 + 0.5).clamp(0, 1), (b * 0.5 + 0.5).clamp(0, 1)
    mse = ((a01 - b01) ** 2).flatten(1).mean(1)
    return 10. * torch.log10(1. / mse)


# ---------------------------------------------------------------------------
# 8(f) row 3: evaluation metrics (video_based/models/eval_utils.py:75-92, 119-333)
# ---------------------------------------------------------------------------
def adjusted_rand_index(true_ids, pred_ids, ignore_background=False):
    """eval_utils.py:119-176 (one_hot + einsum form): ARI per batch element, float32 [B]."""
    if true_ids.dim() == 3:
        true_ids = true_ids.unsqueeze(1)
    if pred_ids.dim() == 3:
        pred_ids = pred_ids.unsqueeze(1)
    true_oh = F.one_hot(true_ids).float()
    pred_oh = F.one_hot(pred_ids).float()
    if ignore_background:
        true_oh = true_oh[..., 1:]
    N = torch.einsum('bthwc,bthwk->bck', true_oh, pred_oh)
    A = N.sum(-1)
    Bc = N.sum(-2)
    num_points = A.sum(1)
    rindex = torch.sum(N * (N - 1), dim=[1, 2])
    aindex = torch.sum(A * (A - 1), dim=1)
    bindex = torch.sum(Bc * (Bc - 1), dim=1)
    expected = aindex * bindex / torch.clamp(num_points * (num_points - 1), min=1)
    max_r = (aindex + bindex) / 2
    den = max_r - expected
    ari = (rindex - expected) / den
    return torch.where(den != 0, ari, torch.tensor(1.).type_as(ari))


def _pair_iou(gt_mask, pred_mask, ignore_background):
    true_oh = F.one_hot(gt_mask).float()
    if ignore_background:
        true_oh = true_oh[..., 1:]
    pred_oh = F.one_hot(pred_mask).float()
    inter = (true_oh[:, :, None] * pred_oh[:, None, :]).sum(0)
    union = true_oh.sum(0)[:, None] + pred_oh.sum(0)[None] - inter
    return (inter / (union + 1e-8)).numpy()


def hungarian_miou(gt_mask, pred_mask, ignore_background=True):
    """eval_utils.py:238-263, masks [H*W] after argmax."""
    import numpy as np
    from scipy.optimize import linear_sum_assignment
    if gt_mask.max().item() == 0 and ignore_background:
        return np.nan
    iou = _pair_iou(gt_mask, pred_mask, ignore_background)
    N, M = iou.shape
    r, c = linear_sum_assignment(iou, maximize=True)
    return iou[r, c].mean() if M >= N else iou[r, c].sum() / float(N)


def mean_best_overlap(gt_mask, pred_mask):
    """eval_utils.py:266-290."""
    import numpy as np
    if gt_mask.max().item() == 0:
        return np.nan
    return _pair_iou(gt_mask, pred_mask, True).max(1).mean()


def seg_metrics(gt, pred):
    """ARI / FG-ARI / mIoU / FG-mIoU / mBO of [B,H,W] id maps, as the *_metric wrappers
    (eval_utils.py:179-190, 293-333)."""
    import numpy as np
    g, q = gt.flatten(1, 2), pred.flatten(1, 2)
    B = gt.shape[0]
    return dict(
        ari=adjusted_rand_index(gt, pred, False).mean().item(),
        fari=adjusted_rand_index(gt, pred, True).mean().item(),
        miou=np.nanmean([hungarian_miou(g[i], q[i], False) for i in range(B)]),
        fmiou=np.nanmean([hungarian_miou(g[i], q[i], True) for i in range(B)]),
        mbo=np.nanmean([mean_best_overlap(g[i], q[i]) for i in range(B)]))


def mse_metric(x, y):
    """eval_utils.py:75-78 (numpy arrays or tensors in [0,1], [B,3,H,W])."""
    d = (x.double() - y.double()) ** 2
    return float(d.sum((-1, -2, -3)).mean())


def psnr_metric(x, y):
    """eval_utils.py:81-92 with skimage's definition restated (skimage is absent here):
    peak_signal_noise_ratio(a, b, data_range=1) = 10 log10(1 / mean((a-b)^2)), float64."""
    d = ((x.double() - y.double()) ** 2).flatten(1).mean(1)
    return float((10.0 * torch.log10(1.0 / d)).mean())


def ssim_metric(x, y):
    """eval_utils.ssim_metric (video_based/models/eval_utils.py:91-106): mean over images of
    skimage.metrics.structural_similarity(x*255, y*255, channel_axis=0, gaussian_weights=True,
    sigma=1.5, use_sample_covariance=False, data_range=255).  skimage v0.19 (environment.yml) is absent
    offline -> PARITY UNPINNED for this function; its documented algorithm (Wang et al. 2004 as
    skimage implements it) is restated on scipy.ndimage.gaussian_filter, the very filter skimage
    calls: float64, truncate=3.5 (radius 5 -> 11 taps), K1 = 0.01, K2 = 0.03, cov_norm = 1, SSIM map
    averaged over the interior left by cropping (win_size - 1) // 2 = 5 pixels, channels averaged."""
    import numpy as np
    from scipy.ndimage import gaussian_filter
    x = np.asarray(x, dtype=np.float64) * 255.
    y = np.asarray(y, dtype=np.float64) * 255.
    L = 255.
    C1, C2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    flt = lambda a: gaussian_filter(a, sigma=1.5, truncate=3.5, mode='reflect')
    vals = []
    for i in range(x.shape[0]):
        ch = []
        for c in range(x.shape[1]):
            a, b = x[i, c], y[i, c]
            ux, uy = flt(a), flt(b)
            vx, vy, vxy = flt(a * a) - ux * ux, flt(b * b) - uy * uy, flt(a * b) - ux * uy
            S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
            ch.append(S[5:-5, 5:-5].mean())
        vals.append(np.mean(ch))
    return float(np.mean(vals))


def shuffle_slots(slots):
    """video_based/test_comp_gen.py:25-31, out of place: slot i of item b comes from item (b + i) % B."""
    out = slots.clone()
    N = slots.shape[-2]
    for i in range(N):
        if slots.dim() == 4:
            out[:, :, i] = torch.cat([slots[i:, :, i], slots[:i, :, i]])
        else:
            out[:, i] = torch.cat([slots[i:, i], slots[:i, i]])
    return out
