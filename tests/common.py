"""Shared fixtures: golden loader, oracle weight dicts (config dicts: slotdiffusion_amd/configs.py)."""
import gzip
import json
import os

import numpy as np
import torch

from slotdiffusion_amd import spec
from slotdiffusion_amd.configs import (clevrtex_cfg, dino_coco_cfg, movid_cfg, movie_cfg,  # noqa: F401
                                       sa_plain_cfg, savi_cfg)
from tests.detfill import det_value, is_buffer_name, make_inputs  # noqa: F401

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def oracle_weights_sa(cfg):
    from slotdiffusion_amd.module import build_grid
    sp = spec.sa_model(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'])
    W = {}
    for i, p in enumerate(sp):
        W[p.name] = build_grid(p.shape[1:3]) if p.init == 'buf:grid' else det_value(p.name, p.shape, i)
    return W


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
