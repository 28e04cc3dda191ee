"""Flat parameter specifications (checkpoint-key compatible with the reference).

The reference builds its models from nested ``nn.Module`` classes; the only
thing a drop-in replacement must preserve from that nesting is the *names and
shapes* of the tensors in ``state_dict()`` ("state_dict
key compatibility is part of the boundary").  Here every network is described
by a flat, ordered list of :class:`P` records -- name, shape, how to
initialise, whether it is a buffer -- produced by small generator functions.
The HIP engine (``slotdiffusion_amd/engine.py``) looks weights up by these
names; ``slotdiffusion_amd/module.py`` materialises them as a tree of bare
``nn.Module`` containers so ``state_dict()/load_state_dict()`` round-trip with
reference checkpoints.

Reference layouts restated (names only, no code shared):
  ResNet-18/GN encoder ...... video_based/models/resnet.py:150-312
  SoftPositionEmbed ......... video_based/models/utils.py:52-63
  SlotAttention ............. img_based/models/slot_attention.py:15-55
  UNetModel ................. video_based/models/unet/unet.py:366-549
  SpatialTransformer ........ video_based/models/unet/attention.py:209-295
  VQ-VAE Encoder/Decoder .... video_based/models/vqvae/modules.py:162-336
  VQVAE wrapper ............. video_based/models/vqvae/VQVAE.py:66-82
  DDPM schedule buffers ..... video_based/models/ddpm/ddpm.py:69-131
  TransformerPredictor ...... video_based/models/predictor.py:20-44
"""
from collections import namedtuple

# init kinds:
#  'lin'   torch default Linear/Conv init: U(+-1/sqrt(fan_in)) (weight and bias)
#  'kfo'   kaiming-normal, fan_out, relu gain (ResNet convs, resnet.py:238-241)
#  'one'/'zero'  constants;  'zlin' = zero-initialised layer (zero_module)
#  'n01'   N(0,1);  'vq' U(+-1/n_e);  'gru' U(+-1/sqrt(hidden)); 'xav' xavier-U
#  'buf:*' non-trainable buffers computed by module.py
P = namedtuple('P', 'name shape init fan_in trainable')


def _p(name, shape, init, fan_in=0, trainable=True):
    return P(name, tuple(int(s) for s in shape), init, int(fan_in), trainable)


def conv(name, cin, cout, k, bias=True, init='lin'):
    fi = cin * k * k
    out = [_p(f'{name}.weight', (cout, cin, k, k), init, fi)]
    if bias:
        out.append(_p(f'{name}.bias', (cout,), 'zero' if init == 'zlin' else 'lin', fi))
    return out


def linear(name, cin, cout, bias=True, init='lin'):
    out = [_p(f'{name}.weight', (cout, cin), init, cin)]
    if bias:
        out.append(_p(f'{name}.bias', (cout,), 'zero' if init in ('zlin', 'xav') else 'lin', cin))
    return out


def norm(name, c):
    return [_p(f'{name}.weight', (c,), 'one'), _p(f'{name}.bias', (c,), 'zero')]


# --------------------------------------------------------------------------
# slot encoder side
# --------------------------------------------------------------------------
def resnet18_gn(prefix='encoder', use_layer4=False):
    """GroupNorm ResNet-18, stride-1 3x3 stem, BasicBlock x2 per stage."""
    out = conv(f'{prefix}.conv1', 3, 64, 3, bias=False, init='kfo')
    out += norm(f'{prefix}.bn1', 64)
    cin = 64
    stages = [(64, 1), (128, 2), (256, 2)] + ([(512, 2)] if use_layer4 else [])
    for li, (planes, stride) in enumerate(stages, start=1):
        for bi in range(2):
            b = f'{prefix}.layer{li}.{bi}'
            s = stride if bi == 0 else 1
            out += conv(f'{b}.conv1', cin, planes, 3, bias=False, init='kfo')
            out += norm(f'{b}.bn1', planes)
            out += conv(f'{b}.conv2', planes, planes, 3, bias=False, init='kfo')
            out += norm(f'{b}.bn2', planes)
            if s != 1 or cin != planes:
                out += conv(f'{b}.downsample.0', cin, planes, 1, bias=False, init='kfo')
                out += norm(f'{b}.downsample.1', planes)
            cin = planes
    return out


def encoder_plan(resolution, enc_dict):
    """-> (plan for engine.encoder_out, visual resolution): the ResNet block list, or the DINO ViT's
    geometry dict (slot_attention.py:180-211)."""
    if enc_dict.get('dino', False):
        _, meta = dino_vit('encoder.dino', enc_dict.get('small_size', True), enc_dict['patch_size'],
                           resolution[0])
        return meta, tuple(r // enc_dict['patch_size'] for r in resolution)
    div = 8 if enc_dict['use_layer4'] else 4
    return resnet18_plan(enc_dict['use_layer4']), tuple(r // div for r in resolution)


def resnet18_plan(use_layer4=False):
    """Structural plan mirrored by engine + oracle: list of (block, cin, cout, stride, has_ds)."""
    plan = []
    cin = 64
    stages = [(64, 1), (128, 2), (256, 2)] + ([(512, 2)] if use_layer4 else [])
    for li, (planes, stride) in enumerate(stages, start=1):
        for bi in range(2):
            s = stride if bi == 0 else 1
            plan.append((f'layer{li}.{bi}', cin, planes, s, s != 1 or cin != planes))
            cin = planes
    return plan


def dino_vit(prefix='encoder.dino', small=True, patch=8, image=224):
    """Frozen DINO ViT of the DINOSAUR-style configs (video_based/models/dino.py:21-60 wraps
    transformers.ViTModel.from_pretrained('facebook/dino-vit{s,b}{8,16}')).  Key names and order are
    those of transformers 4.27.4's ViTModel (the version the reference pins, environment.yml:216) --
    recalled, not verifiable offline (the installed 5.x renames them; tools/gen_golden.py maps).  All
    tensors are frozen; the pooler exists in the checkpoint but is never evaluated."""
    hid, heads, mlp, layers = (384, 6, 1536, 12) if small else (768, 12, 3072, 12)
    ntok = (image // patch) ** 2 + 1
    fz = dict(trainable=False)
    out = [_p(f'{prefix}.embeddings.cls_token', (1, 1, hid), 'n01', **fz),
           _p(f'{prefix}.embeddings.position_embeddings', (1, ntok, hid), 'n01', **fz),
           _p(f'{prefix}.embeddings.patch_embeddings.projection.weight', (hid, 3, patch, patch), 'lin',
              3 * patch * patch, **fz),
           _p(f'{prefix}.embeddings.patch_embeddings.projection.bias', (hid,), 'lin', 3 * patch * patch, **fz)]

    def lin(name, cin, cout):
        return [_p(f'{name}.weight', (cout, cin), 'lin', cin, **fz), _p(f'{name}.bias', (cout,), 'lin', cin, **fz)]

    def ln(name):
        return [_p(f'{name}.weight', (hid,), 'one', **fz), _p(f'{name}.bias', (hid,), 'zero', **fz)]
    for i in range(layers):
        l = f'{prefix}.encoder.layer.{i}'
        for n in ('query', 'key', 'value'):
            out += lin(f'{l}.attention.attention.{n}', hid, hid)
        out += lin(f'{l}.attention.output.dense', hid, hid)
        out += lin(f'{l}.intermediate.dense', hid, mlp)
        out += lin(f'{l}.output.dense', mlp, hid)
        out += ln(f'{l}.layernorm_before') + ln(f'{l}.layernorm_after')
    out += ln(f'{prefix}.layernorm')
    out += lin(f'{prefix}.pooler.dense', hid, hid)
    return out, dict(hidden=hid, heads=heads, mlp=mlp, layers=layers, patch=patch, ntok=ntok)


def soft_pos_embed(name, hidden, res):
    return [_p(f'{name}.grid', (1, res[0], res[1], 4), 'buf:grid', trainable=False)] + \
        linear(f'{name}.dense', 4, hidden)


def encoder_head(name, cin, cout):
    return norm(f'{name}.0', cin) + linear(f'{name}.1', cin, cout) + linear(f'{name}.3', cout, cout)


def slot_attention(name, in_features, slot_size, mlp_hidden):
    d = slot_size
    out = norm(f'{name}.norm_inputs', in_features)
    out += norm(f'{name}.project_q.0', d)
    out += linear(f'{name}.project_q.1', d, d, bias=False)
    out += linear(f'{name}.project_k', in_features, d, bias=False)
    out += linear(f'{name}.project_v', in_features, d, bias=False)
    out += [_p(f'{name}.gru.weight_ih', (3 * d, d), 'gru', d),
            _p(f'{name}.gru.weight_hh', (3 * d, d), 'gru', d),
            _p(f'{name}.gru.bias_ih', (3 * d,), 'gru', d),
            _p(f'{name}.gru.bias_hh', (3 * d,), 'gru', d)]
    out += norm(f'{name}.mlp.0', d)
    out += linear(f'{name}.mlp.1', d, mlp_hidden)
    out += linear(f'{name}.mlp.3', mlp_hidden, d)
    return out


def transformer_predictor(name, d_model, num_layers, ffn_dim):
    out = []
    for i in range(num_layers):
        l = f'{name}.transformer_encoder.layers.{i}'
        out += [_p(f'{l}.self_attn.in_proj_weight', (3 * d_model, d_model), 'xav', d_model),
                _p(f'{l}.self_attn.in_proj_bias', (3 * d_model,), 'zero')]
        out += linear(f'{l}.self_attn.out_proj', d_model, d_model, init='lin')
        out[-1] = out[-1]._replace(init='zero')
        out += linear(f'{l}.linear1', d_model, ffn_dim)
        out += linear(f'{l}.linear2', ffn_dim, d_model)
        out += norm(f'{l}.norm1', d_model) + norm(f'{l}.norm2', d_model)
    return out


# --------------------------------------------------------------------------
# LDM UNet
# --------------------------------------------------------------------------
def _resblock(name, cin, cout, emb_ch):
    out = norm(f'{name}.in_layers.0', cin)
    out += conv(f'{name}.in_layers.2', cin, cout, 3)
    out += linear(f'{name}.emb_layers.1', emb_ch, cout)
    out += norm(f'{name}.out_layers.0', cout)
    out += conv(f'{name}.out_layers.3', cout, cout, 3, init='zlin')
    if cin != cout:
        out += conv(f'{name}.skip_connection', cin, cout, 1)
    return out


def _spatial_transformer(name, ch, ctx_dim):
    out = norm(f'{name}.norm', ch)
    out += conv(f'{name}.proj_in', ch, ch, 1)
    t = f'{name}.transformer_blocks.0'
    for a, kd in (('attn1', ch), ('attn2', ctx_dim)):
        if a == 'attn2':
            out += linear(f'{t}.ff.net.0.proj', ch, 8 * ch)
            out += linear(f'{t}.ff.net.2', 4 * ch, ch)
        out += linear(f'{t}.{a}.to_q', ch, ch, bias=False)
        out += linear(f'{t}.{a}.to_k', kd, ch, bias=False)
        out += linear(f'{t}.{a}.to_v', kd, ch, bias=False)
        out += linear(f'{t}.{a}.to_out.0', ch, ch)
    out += norm(f'{t}.norm1', ch) + norm(f'{t}.norm2', ch) + norm(f'{t}.norm3', ch)
    out += conv(f'{name}.proj_out', ch, ch, 1, init='zlin')
    return out


def unet_plan(cfg):
    """Block-level structure of the UNet shared by spec, engine and oracle.

    Returns dict(input=[...], middle=[...], output=[...]) where each entry is a
    list of layer tuples: ('conv', name, cin, cout) | ('res', name, cin, cout)
    | ('st', name, ch, heads) | ('down', name, ch) | ('up', name, ch).
    """
    mc = cfg['model_channels']
    mult = tuple(cfg['channel_mult'])
    nrb = cfg['num_res_blocks']
    att = tuple(cfg['attention_resolutions'])
    hc = cfg['num_head_channels']
    inp = [[('conv', 'input_blocks.0.0', cfg['in_channels'], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            i = len(inp)
            layers = [('res', f'input_blocks.{i}.0', ch, m * mc)]
            ch = m * mc
            if ds in att:
                layers.append(('st', f'input_blocks.{i}.1', ch, ch // hc))
            inp.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            i = len(inp)
            inp.append([('down', f'input_blocks.{i}.0', ch)])
            chans.append(ch)
            ds *= 2
    mid = [('res', 'middle_block.0', ch, ch), ('st', 'middle_block.1', ch, ch // hc),
           ('res', 'middle_block.2', ch, ch)]
    outp = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            j = len(outp)
            layers = [('res', f'output_blocks.{j}.0', ch + ich, mc * m)]
            ch = mc * m
            if ds in att:
                layers.append(('st', f'output_blocks.{j}.{len(layers)}', ch, ch // hc))
            if level and i == nrb:
                layers.append(('up', f'output_blocks.{j}.{len(layers)}', ch))
                ds //= 2
            outp.append(layers)
    return dict(input=inp, middle=mid, output=outp, out_ch=ch)


def unet(prefix, cfg):
    mc = cfg['model_channels']
    emb = 4 * mc
    ctx = cfg['context_dim']
    plan = unet_plan(cfg)
    out = linear(f'{prefix}.time_embed.0', mc, emb) + linear(f'{prefix}.time_embed.2', emb, emb)

    def emit(layers):
        o = []
        for l in layers:
            kind, name = l[0], f'{prefix}.{l[1]}'
            if kind == 'conv':
                o += conv(name, l[2], l[3], 3)
            elif kind == 'res':
                o += _resblock(name, l[2], l[3], emb)
            elif kind == 'st':
                o += _spatial_transformer(name, l[2], ctx)
            elif kind == 'down':
                o += conv(f'{name}.op', l[2], l[2], 3)
            elif kind == 'up':
                o += conv(f'{name}.conv', l[2], l[2], 3)
        return o

    for blk in plan['input']:
        out += emit(blk)
    out += emit(plan['middle'])
    for blk in plan['output']:
        out += emit(blk)
    out += norm(f'{prefix}.out.0', plan['out_ch'])
    out += conv(f'{prefix}.out.2', mc, cfg['out_channels'], 3, init='zlin')
    return out


# --------------------------------------------------------------------------
# VQ-VAE
# --------------------------------------------------------------------------
def _vae_resblock(name, cin, cout):
    out = norm(f'{name}.norm1', cin) + conv(f'{name}.conv1', cin, cout, 3)
    out += norm(f'{name}.norm2', cout) + conv(f'{name}.conv2', cout, cout, 3)
    if cin != cout:
        out += conv(f'{name}.nin_shortcut', cin, cout, 1)
    return out


def _vae_attn(name, c):
    out = norm(f'{name}.norm', c)
    for n in ('q', 'k', 'v', 'proj_out'):
        out += conv(f'{name}.{n}', c, c, 1)
    return out


def vqvae(prefix, ed, vq):
    ch, mult, nrb = ed['ch'], tuple(ed['ch_mult']), ed['num_res_blocks']
    zc = ed['z_channels']
    assert not ed.get('attn_resolutions'), 'attn_resolutions=[] in every LDM config'
    e = f'{prefix}.encoder'
    out = conv(f'{e}.conv_in', ed['in_channels'], ch, 3)
    in_mult = (1,) + mult
    bi = ch
    for lvl in range(len(mult)):
        bi, bo = ch * in_mult[lvl], ch * mult[lvl]
        for b in range(nrb):
            out += _vae_resblock(f'{e}.down.{lvl}.block.{b}', bi, bo)
            bi = bo
        if lvl != len(mult) - 1:
            out += conv(f'{e}.down.{lvl}.downsample.conv', bi, bi, 3)
    out += _vae_resblock(f'{e}.mid.block_1', bi, bi) + _vae_attn(f'{e}.mid.attn_1', bi)
    out += _vae_resblock(f'{e}.mid.block_2', bi, bi)
    out += norm(f'{e}.norm_out', bi) + conv(f'{e}.conv_out', bi, zc, 3)
    d = f'{prefix}.decoder'
    bi = ch * mult[-1]
    out += conv(f'{d}.conv_in', zc, bi, 3)
    out += _vae_resblock(f'{d}.mid.block_1', bi, bi) + _vae_attn(f'{d}.mid.attn_1', bi)
    out += _vae_resblock(f'{d}.mid.block_2', bi, bi)
    ups = {}
    for lvl in reversed(range(len(mult))):
        bo = ch * mult[lvl]
        o = []
        for b in range(nrb + 1):
            o += _vae_resblock(f'{d}.up.{lvl}.block.{b}', bi, bo)
            bi = bo
        if lvl != 0:
            o += conv(f'{d}.up.{lvl}.upsample.conv', bi, bi, 3)
        ups[lvl] = o
    for lvl in range(len(mult)):          # ModuleList.insert(0, ...) => ascending key order
        out += ups[lvl]
    out += norm(f'{d}.norm_out', bi) + conv(f'{d}.conv_out', bi, ed['out_ch'], 3)
    out += [_p(f'{prefix}.quantize.embedding.weight', (vq['n_embed'], vq['embed_dim']), 'vq',
               vq['n_embed'])]
    out += conv(f'{prefix}.quant_conv', zc, vq['embed_dim'], 1)
    out += conv(f'{prefix}.post_quant_conv', vq['embed_dim'], zc, 1)
    return [p._replace(trainable=False) for p in out]


def vqvae_model(ed, vq):
    """Stand-alone VQ-VAE (registry name 'VQVAE', video_based/models/vqvae/VQVAE.py:40-84): the
    same tensors as the LDM's frozen copy, without a prefix and trainable."""
    return [p._replace(name=p.name[2:], trainable=True) for p in vqvae('X', ed, vq)]


DDPM_BUFFERS = ('betas', 'alphas_bar', 'alphas_bar_prev', 'sqrt_alphas_bar',
                'sqrt_one_minus_alphas_bar', 'log_one_minus_alphas_bar',
                'sqrt_recip_alphas_bar', 'sqrt_recipm1_alphas_bar', 'posterior_variance',
                'posterior_log_variance_clipped', 'posterior_mean_coef1',
                'posterior_mean_coef2')


def ddpm_buffers(prefix, timesteps):
    return [_p(f'{prefix}.{n}', (timesteps,), f'buf:{n}', trainable=False) for n in DDPM_BUFFERS]


def ldm(prefix, dec_dict):
    """`dm_decoder` of SADiffusion / SAViDiffusion (reference ldm.py:21-57)."""
    dd = dec_dict['diffusion_dict']
    out = ddpm_buffers(prefix, dd.get('timesteps', 1000))
    out += unet(f'{prefix}.model.diffusion_model', dec_dict['unet_dict'])
    va = dec_dict['vae_dict']
    out += vqvae(f'{prefix}.vae.vqvae', va['enc_dec_dict'], va['vq_dict'])
    return out


# --------------------------------------------------------------------------
# whole models
# --------------------------------------------------------------------------
def sa_encoder_side(resolution, slot_dict, enc_dict):
    dino = bool(enc_dict.get('dino', False))
    assert dino or enc_dict.get('resnet') == 'resnet18', \
        'hot path covers the ResNet-18 and the frozen DINO ViT encoders (the plain nerv CNN encoder of ' \
        'the MOVi-Solid / MOVi-Tex configs is out of scope: SURVEY section 8(f) row 4)'
    d = slot_dict['slot_size']
    out = [_p('init_latents', (1, slot_dict['num_slots'], d), 'n01')]
    out += slot_attention('slot_attention', enc_dict['enc_out_channels'], d,
                          slot_dict['slot_mlp_size'])
    if dino:               # slot_attention.py:196-211: features [B, 384 | 768, H/patch, W/patch]
        patch = enc_dict['patch_size']
        assert patch in (8, 16) and resolution[0] == resolution[1]
        vit, meta = dino_vit('encoder.dino', enc_dict.get('small_size', True), patch, resolution[0])
        vis_ch, vis_res = meta['hidden'], tuple(r // patch for r in resolution)
        out += vit
        out += soft_pos_embed('encoder_pos_embedding', vis_ch, vis_res)
        out += encoder_head('encoder_out_layer', vis_ch, enc_dict['enc_out_channels'])
        return out, vis_ch, vis_res
    use_l4 = enc_dict['use_layer4']
    vis_ch = 512 if use_l4 else 256
    div = 8 if use_l4 else 4
    vis_res = tuple(r // div for r in resolution)
    out += resnet18_gn('encoder', use_l4)
    out += soft_pos_embed('encoder_pos_embedding', vis_ch, vis_res)
    out += encoder_head('encoder_out_layer', vis_ch, enc_dict['enc_out_channels'])
    return out, vis_ch, vis_res


def deconv(name, cin, cout, k):
    """nn.ConvTranspose2d(cin, cout, k): weight [cin, cout, k, k] (torch fan_in = cout*k*k)."""
    fi = cout * k * k
    return [_p(f'{name}.weight', (cin, cout, k, k), 'lin', fi), _p(f'{name}.bias', (cout,), 'lin', fi)]


def sa_decoder_plan(resolution, dec_dict):
    """[(kind, cin, cout, k, stride)] of the plain-SA CNN decoder (img_based/models/
    slot_attention.py:251-292): stride-2 transposed convs until the image resolution is reached,
    then stride 1; final 1x1 conv to rgb + alpha."""
    ch = list(dec_dict['dec_channels'])
    k = dec_dict['dec_ks']
    size = tuple(dec_dict['dec_resolution'])
    plan, stride = [], 2
    for i in range(len(ch) - 1):
        if size == tuple(resolution):
            stride = 1
        plan.append(('deconv', ch[i], ch[i + 1], k, stride))
        size = tuple((v - 1) * stride - 2 * (k // 2) + (k - 1) + (stride - 1) + 1 for v in size)
    assert size == tuple(resolution), 'decoder output does not match the image resolution'
    plan.append(('conv', ch[-1], 4, 1, 1))
    return plan


def sa_model(resolution, slot_dict, enc_dict, dec_dict):
    """Plain Slot Attention auto-encoder (registry name 'SA'), key order of the reference
    constructor (slot_attention.py:153-156)."""
    out, _, _ = sa_encoder_side(resolution, slot_dict, enc_dict)
    assert dec_dict.get('dec_norm', '') == '', 'decoder norm layers are not part of the path'
    assert dec_dict['dec_channels'][0] == slot_dict['slot_size']
    for i, (kind, cin, cout, k, _) in enumerate(sa_decoder_plan(resolution, dec_dict)):
        out += deconv(f'decoder.{i}.0', cin, cout, k) if kind == 'deconv' else conv(f'decoder.{i}', cin, cout, k)
    out += soft_pos_embed('decoder_pos_embedding', slot_dict['slot_size'], dec_dict['dec_resolution'])
    return out


def savi_model(resolution, slot_dict, enc_dict, dec_dict, pred_dict):
    """Video Slot Attention baseline (registry name 'SAVi', video_based/models/savi.py:117-170):
    the plain-SA tensors followed by the slot transition predictor."""
    pred = transformer_predictor('predictor', pred_dict['pred_slot_size'] if 'pred_slot_size'
                                 in pred_dict else slot_dict['slot_size'],
                                 pred_dict['pred_num_layers'], pred_dict['pred_ffn_dim'])
    return sa_model(resolution, slot_dict, enc_dict, dec_dict) + pred


def sa_diffusion(resolution, slot_dict, enc_dict, dec_dict):
    out, _, _ = sa_encoder_side(resolution, slot_dict, enc_dict)
    return out + ldm('dm_decoder', dec_dict)


def savi_diffusion(resolution, slot_dict, enc_dict, dec_dict, pred_dict):
    """Video model: same tensors + the slot transition predictor (savi.py:154-170 order)."""
    enc, _, _ = sa_encoder_side(resolution, slot_dict, enc_dict)
    head = [p for p in enc if p.name == 'init_latents' or p.name.startswith('slot_attention.')]
    rest = [p for p in enc if p not in head]
    pred = transformer_predictor('predictor', pred_dict['pred_slot_size'] if 'pred_slot_size'
                                 in pred_dict else slot_dict['slot_size'],
                                 pred_dict['pred_num_layers'], pred_dict['pred_ffn_dim'])
    return head + rest + ldm('dm_decoder', dec_dict) + pred
