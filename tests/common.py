"""Shared fixtures: CLEVRTex-7slot config dicts, golden loader, oracle weight dict."""
import gzip
import json
import os

import numpy as np
import torch

from slotdiffusion_amd import spec
from tests.detfill import det_value, is_buffer_name, make_inputs  # noqa: F401

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def clevrtex_cfg(num_slots=7):
    """Restates img_based/configs/sa_ldm/sa_ldm_clevrtex_params-res128.py (values only)."""
    res = (128, 128)
    d = 192
    return dict(
        resolution=res,
        slot_dict=dict(num_slots=num_slots, slot_size=d, slot_mlp_size=2 * d, num_iterations=3),
        enc_dict=dict(resnet='resnet18', use_layer4=False, enc_out_channels=d),
        dec_dict=dict(
            resolution=(32, 32),
            vae_dict=dict(
                vae_type='VQVAE',
                enc_dec_dict=dict(resolution=128, in_channels=3, z_channels=3, ch=64,
                                  ch_mult=[1, 2, 4], num_res_blocks=2, attn_resolutions=[],
                                  out_ch=3, dropout=0.0),
                vq_dict=dict(n_embed=4096, embed_dim=3, percept_loss_w=1.0),
                vqvae_ckp_path='./pretrained/vqvae_clevrtex_params-res128.pth'),
            unet_dict=dict(in_channels=3, model_channels=128, out_channels=3, num_res_blocks=2,
                           attention_resolutions=(8, 4, 2), dropout=0.1,
                           channel_mult=(1, 2, 3, 4), dims=2, use_checkpoint=False,
                           num_head_channels=32, resblock_updown=False, conv_resample=True,
                           transformer_depth=1, context_dim=d, n_embed=None),
            use_ema=False,
            diffusion_dict=dict(pred_target='eps', z_scale_factor=1., timesteps=1000,
                                beta_schedule='linear', linear_start=0.0015, linear_end=0.0195,
                                cosine_s=8e-3, log_every_t=200, logvar_init=0.),
            conditioning_key='crossattn', cond_stage_key='slots'),
        loss_dict=dict(use_denoise_loss=True))


def sa_plain_cfg(num_slots=7):
    """Restates img_based/configs/sa/sa_clevrtex_params-res128.py (values only; BASELINE.json
    config 0 asks 7 slots)."""
    d = 192
    return dict(resolution=(128, 128),
                slot_dict=dict(num_slots=num_slots, slot_size=d, slot_mlp_size=2 * d, num_iterations=3),
                enc_dict=dict(resnet='resnet18', use_layer4=False, enc_out_channels=d),
                dec_dict=dict(dec_channels=(d, 128, 128, 128, 128), dec_resolution=(8, 8), dec_ks=5,
                              dec_norm=''),
                loss_dict=dict(use_img_recon_loss=True))


def oracle_weights_sa(cfg):
    from slotdiffusion_amd.module import build_grid
    sp = spec.sa_model(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'])
    W = {}
    for i, p in enumerate(sp):
        W[p.name] = build_grid(p.shape[1:3]) if p.init == 'buf:grid' else det_value(p.name, p.shape, i)
    return W


def savi_cfg():
    """Restates video_based/configs/savi/savi_movie_params-res128.py (values only)."""
    d = 192
    return dict(resolution=(128, 128), clip_len=3,
                slot_dict=dict(num_slots=15, slot_size=d, slot_mlp_size=2 * d, num_iterations=2),
                enc_dict=dict(resnet='resnet18', use_layer4=False, enc_out_channels=d,
                              replace_stride_with_dilation=[False, False, False]),
                dec_dict=dict(dec_channels=(d, 64, 64, 64, 64), dec_resolution=(8, 8), dec_ks=5,
                              dec_norm=''),
                pred_dict=dict(pred_type='transformer', pred_rnn=False, pred_norm_first=True,
                               pred_num_layers=2, pred_num_heads=4, pred_ffn_dim=4 * d,
                               pred_sg_every=None),
                loss_dict=dict(use_img_recon_loss=True))


def oracle_weights_savi(cfg):
    from slotdiffusion_amd.module import build_grid
    sp = spec.savi_model(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                         cfg['pred_dict'])
    W = {}
    for i, p in enumerate(sp):
        W[p.name] = build_grid(p.shape[1:3]) if p.init == 'buf:grid' else det_value(p.name, p.shape, i)
    return W


def oracle_weights_vqvae(cfg):
    va = cfg['dec_dict']['vae_dict']
    sp = spec.vqvae_model(va['enc_dec_dict'], va['vq_dict'])
    return {p.name: det_value(p.name, p.shape, i) for i, p in enumerate(sp)}


def load_golden(name='sadiff_b2.npz'):
    z = np.load(os.path.join(GOLD, name))
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in 'fiu' else z[k]) for k in z.files}


def load_keys():
    with gzip.open(os.path.join(GOLD, 'state_dict_keys.json.gz'), 'rt') as f:
        return json.load(f)


def oracle_weights(cfg, seed=1234):
    """{key: fp32 CPU tensor} with tests/detfill.py values (buffers from the schedule)."""
    from slotdiffusion_amd.module import build_grid, ddpm_schedule
    sp = spec.sa_diffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'])
    dd = {k: v for k, v in cfg['dec_dict']['diffusion_dict'].items()
          if k in ('timesteps', 'beta_schedule', 'linear_start', 'linear_end')}
    sched = ddpm_schedule(**dd)
    W = {}
    for i, p in enumerate(sp):
        if p.init.startswith('buf:'):
            key = p.init[4:]
            W[p.name] = build_grid(p.shape[1:3]) if key == 'grid' else \
                torch.tensor(sched[key], dtype=torch.float32)
        else:
            W[p.name] = det_value(p.name, p.shape, i, seed)
    return W


def movie_cfg():
    """Restates video_based/configs/savi_ldm/savi_ldm_movie_params-res128.py (values only)."""
    cfg = clevrtex_cfg(num_slots=15)
    cfg['slot_dict']['num_iterations'] = 2
    cfg['clip_len'] = 3
    cfg['pred_dict'] = dict(pred_type='transformer', pred_rnn=False, pred_norm_first=True,
                            pred_num_layers=2, pred_num_heads=4, pred_ffn_dim=768,
                            pred_sg_every=None)
    cfg['dec_dict']['vae_dict']['vqvae_ckp_path'] = './pretrained/vqvae_movie_params-res128.pth'
    return cfg


def movid_cfg(num_slots=11, clip_len=6):
    """video_based/configs/savi_ldm/savi_ldm_movid_params-res128.py with the two values BASELINE.json
    config 2 overrides (11 slots, 6-frame clips; the file ships 15 / 3).  Everything else equals
    the MOVi-E config (only dataset level and checkpoint path differ)."""
    cfg = movie_cfg()
    cfg['slot_dict']['num_slots'] = num_slots
    cfg['clip_len'] = clip_len
    cfg['dec_dict']['vae_dict']['vqvae_ckp_path'] = './pretrained/vqvae_movid_params-res128.pth'
    return cfg


def dino_coco_cfg():
    """img_based/configs/sa_ldm/sa_ldm_dino_coco_params-res224.py (BASELINE config 5) from its dumped
    values (tests/golden/configs): 224 x 224, DINO ViT-S/8 features 28 x 28 x 384, 7 slots of 256,
    latent 56 x 56."""
    with open(os.path.join(GOLD, 'configs', 'img_based__sa_ldm_dino_coco_params-res224.json')) as f:
        d = json.load(f)
    return {k: d[k] for k in ('resolution', 'slot_dict', 'enc_dict', 'dec_dict', 'loss_dict')}


def dino_inputs():
    g = torch.Generator().manual_seed(21)
    img = (torch.randn(1, 3, 224, 224, generator=g) * 0.5).clamp(-1, 1)
    noise = torch.randn(1, 3, 56, 56, generator=g)
    return img, noise


def oracle_weights_generic(sp):
    """{key: fp32 CPU tensor} for a spec list with tests/detfill.py values (buffers: grid / schedule
    are filled by the caller where needed)."""
    return {p.name: det_value(p.name, p.shape, i) for i, p in enumerate(sp) if not p.init.startswith('buf:')}


def oracle_weights_video(cfg, seed=1234):
    from slotdiffusion_amd.module import build_grid, ddpm_schedule
    sp = spec.savi_diffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                             cfg['pred_dict'])
    dd = {k: v for k, v in cfg['dec_dict']['diffusion_dict'].items()
          if k in ('timesteps', 'beta_schedule', 'linear_start', 'linear_end')}
    sched = ddpm_schedule(**dd)
    W = {}
    for i, p in enumerate(sp):
        if p.init.startswith('buf:'):
            key = p.init[4:]
            W[p.name] = build_grid(p.shape[1:3]) if key == 'grid' else \
                torch.tensor(sched[key], dtype=torch.float32)
        else:
            W[p.name] = det_value(p.name, p.shape, i, seed)
    return W


class Params:
    """Config object with the attribute surface of the reference's *_params.py classes."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def get(self, key, default=None):
        return getattr(self, key, default)


def make_params(model='SADiffusion', batch=2):
    if model == 'VQVAE':           # img_based/configs/sa_ldm/vqvae_clevrtex_params-res128.py (values)
        va = clevrtex_cfg()['dec_dict']['vae_dict']
        return Params(model=model, optimizer='Adam', weight_decay=0.0, max_epochs=1, lr=1e-3,
                      clip_grad=-1, warmup_steps_pct=0.05, train_batch_size=batch, dataset='clevrtex',
                      san_check_val_step=0, resolution=(128, 128), enc_dec_dict=va['enc_dec_dict'],
                      vq_dict=va['vq_dict'], recon_loss_w=1., quant_loss_w=1., percept_loss_w=1.)
    cfg = sa_plain_cfg() if model == 'SA' else clevrtex_cfg()
    extra = dict(lr=4e-4, clip_grad=-1, warmup_steps_pct=0.025, img_recon_loss_w=1.) if model == 'SA' \
        else dict(lr=1e-4, dec_lr=2e-4, clip_grad=1.0, warmup_steps_pct=0.05, denoise_loss_w=1.)
    return Params(model=model, optimizer='Adam', weight_decay=0.0, max_epochs=1,
                  train_batch_size=batch, dataset='clevrtex', san_check_val_step=0, **cfg, **extra)
