"""Materialise a flat parameter spec as an ``nn.Module`` tree.

The tree consists of *bare containers* (no forward code): its only job is to
own the tensors under exactly the dotted names the reference's checkpoints
use, so ``state_dict()``, ``load_state_dict()``, ``named_parameters()`` and
``.to()/.cuda()`` behave like the reference modules'.  Compute lives in
``engine.py`` and reads tensors from the flat dict returned by
:meth:`FlatModule.tensors`.

Convolution weights ``[Cout, Cin, kh, kw]`` are allocated with
``torch.channels_last`` strides: the logical shape (and therefore the
checkpoint format) is the reference's, while the bytes in HBM are
``[Cout][kh][kw][Cin]`` -- the K-contiguous operand layout the implicit-GEMM
kernels consume.  Gradients are produced in the same physical layout, so the
fused Adam kernel walks weights, grads and moments linearly.
"""
import math

import numpy as np
import torch
import torch.nn as nn


class _Node(nn.Module):
    """Container node; children/parameters are attached dynamically."""


def make_beta_schedule_linear(n, start, end):
    # reference ddpm/utils.py:21-27 -- float64 linspace of sqrt, squared
    return np.linspace(start ** 0.5, end ** 0.5, n, dtype=np.float64) ** 2


def ddpm_schedule(timesteps=1000, beta_schedule='linear', linear_start=1e-4, linear_end=2e-2,
                  **_unused):
    """float64 schedule tables, cast to fp32 by the caller (ddpm.py:69-131)."""
    assert beta_schedule == 'linear', 'hot path covers the linear schedule of every LDM config'
    betas = make_beta_schedule_linear(timesteps, linear_start, linear_end)
    alphas = 1. - betas
    ab = np.cumprod(alphas, axis=0)
    ab_prev = np.append(1., ab[:-1])
    pv = betas * (1. - ab_prev) / (1. - ab)
    return {
        'betas': betas,
        'alphas_bar': ab,
        'alphas_bar_prev': ab_prev,
        'sqrt_alphas_bar': np.sqrt(ab),
        'sqrt_one_minus_alphas_bar': np.sqrt(1. - ab),
        'log_one_minus_alphas_bar': np.log(1. - ab),
        'sqrt_recip_alphas_bar': np.sqrt(1. / ab),
        'sqrt_recipm1_alphas_bar': np.sqrt(1. / ab - 1),
        'posterior_variance': pv,
        'posterior_log_variance_clipped': np.log(np.maximum(pv, 1e-20)),
        'posterior_mean_coef1': betas * np.sqrt(ab_prev) / (1. - ab),
        'posterior_mean_coef2': (1. - ab_prev) * np.sqrt(alphas) / (1. - ab),
    }


def build_grid(resolution):
    """[1, H, W, 4] = (y, x, 1-y, 1-x) in [0, 1] (models/utils.py:37-44)."""
    ys = torch.linspace(0., 1., steps=resolution[0])
    xs = torch.linspace(0., 1., steps=resolution[1])
    gy, gx = torch.meshgrid(ys, xs, indexing='ij')
    g = torch.stack([gy, gx], dim=-1).unsqueeze(0)
    return torch.cat([g, 1. - g], dim=-1)


def _init_tensor(p, gen, sched):
    shape = p.shape
    if p.init.startswith('buf:'):
        key = p.init[4:]
        if key == 'grid':
            return build_grid(shape[1:3])
        return torch.tensor(sched[key], dtype=torch.float32)
    t = torch.empty(shape, dtype=torch.float32)
    if p.init == 'lin':
        b = 1. / math.sqrt(p.fan_in) if p.fan_in > 0 else 0.
        t.uniform_(-b, b, generator=gen)
    elif p.init == 'kfo':
        fan_out = shape[0] * shape[2] * shape[3]
        t.normal_(0., math.sqrt(2. / fan_out), generator=gen)
    elif p.init == 'one':
        t.fill_(1.)
    elif p.init in ('zero', 'zlin'):
        t.zero_()
    elif p.init == 'n01':
        t.normal_(0., 1., generator=gen)
    elif p.init == 'vq':
        t.uniform_(-1. / p.fan_in, 1. / p.fan_in, generator=gen)
    elif p.init == 'gru':
        b = 1. / math.sqrt(p.fan_in)
        t.uniform_(-b, b, generator=gen)
    elif p.init == 'xav':
        b = math.sqrt(6. / (shape[0] + shape[1]))
        t.uniform_(-b, b, generator=gen)
    else:
        raise ValueError(p.init)
    return t


class FlatModule(nn.Module):
    """nn.Module whose parameters/buffers are created from a list of spec.P."""

    def __init__(self, spec_list, schedule_kwargs=None, seed=0, node_classes=None):
        super().__init__()
        node_classes = node_classes or {}
        self._spec = list(spec_list)
        gen = torch.Generator().manual_seed(seed)
        sched = ddpm_schedule(**schedule_kwargs) if schedule_kwargs is not None else None
        for p in self._spec:
            t = _init_tensor(p, gen, sched)
            if t.dim() == 4:
                t = t.contiguous(memory_format=torch.channels_last)
            parts = p.name.split('.')
            node = self
            for depth, part in enumerate(parts[:-1]):
                if not hasattr(node, part):
                    cls = node_classes.get('.'.join(parts[:depth + 1]), _Node)
                    node.add_module(part, cls())
                node = getattr(node, part)
            if p.init.startswith('buf:'):
                node.register_buffer(parts[-1], t)
            else:
                node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=p.trainable))

    def tensors(self):
        """Flat {dotted name: tensor} view over parameters and buffers."""
        out = dict(self.named_parameters())
        out.update(dict(self.named_buffers()))
        return out

    def _apply(self, fn, recurse=True):
        # keep conv weights channels_last across .to()/.cuda() (nn.Module preserves strides for
        # dense non-overlapping tensors; assert instead of silently re-laying out)
        r = super()._apply(fn, recurse)
        for n, p in self.named_parameters():
            if p.dim() == 4 and p.shape[1] > 1 and (p.shape[2] > 1 or p.shape[3] > 1):
                assert p.is_contiguous(memory_format=torch.channels_last), n
        return r
