"""Materialise a flat parameter spec as an ``nn.Module`` tree backed by flat HBM arenas.

The tree consists of *bare containers* (no forward code): its only job is to own the tensors
under exactly the dotted names the reference's checkpoints use, so ``state_dict()``,
``load_state_dict()``, ``named_parameters()`` and ``.to()/.cuda()`` behave like the reference
modules'.  Compute lives in ``engine.py`` and reads tensors from :meth:`FlatModule.tensors`.

MI355X-side memory design
  * every parameter is a VIEW into one flat fp32 arena, laid out in spec order:
        [ slot-encoder group | dm_decoder (UNet) group | frozen VQ-VAE ]
    so the two learning-rate groups of the reference's optimiser (vb/method.py:291-341) are two
    contiguous ranges: global-norm clip + Adam are single streaming kernels over the arena, and a
    DDP all-reduce is a handful of large flat buckets;
  * gradients live in a second arena with identical offsets (``p.grad`` are views of it): the
    wgrad kernels write their fp32 results straight into it;
  * the bf16 compute path reads a bf16 shadow arena (same offsets) refreshed by the Adam kernel;
  * conv weights ``[Cout, Cin, kh, kw]`` keep the reference's logical shape but are stored
    ``[Cout][kh][kw][Cin]`` (``torch.channels_last`` strides) -- the K-contiguous operand layout of
    the implicit-GEMM kernels; q|k|v (and cross-attention k|v) projections are adjacent in the
    arena, so their fused [3C, C] operand / gradient is one contiguous slice.
"""
import math

import numpy as np
import torch
import torch.nn as nn

ALIGN = 8     # elements: keeps every tensor 16-byte aligned in the bf16 shadow too


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


def _view(arena, off, shape):
    n = 1
    for s in shape:
        n *= s
    flat = arena[off:off + n]
    if len(shape) == 4:                      # physical [Cout][kh][kw][Cin], logical NCHW shape
        co, ci, kh, kw = shape
        return flat.view(co, kh, kw, ci).permute(0, 3, 1, 2)
    return flat.view(shape)


class FlatModule(nn.Module):
    """nn.Module whose parameters/buffers are created from a list of spec.P (arena-backed)."""

    def __init__(self, spec_list, schedule_kwargs=None, seed=0, node_classes=None,
                 lr_group_of=None):
        super().__init__()
        node_classes = node_classes or {}
        self._spec = list(spec_list)
        self.seed = int(seed)               # run seed (also keys the dropout generator)
        gen = torch.Generator().manual_seed(seed)
        sched = ddpm_schedule(**schedule_kwargs) if schedule_kwargs is not None else None
        # ---- arena layout: trainable params in spec order, then frozen ones
        pspecs = [p for p in self._spec if not p.init.startswith('buf:')]
        order = [p for p in pspecs if p.trainable] + [p for p in pspecs if not p.trainable]
        self._offsets = {}
        off = 0
        for p in order:
            n = int(np.prod(p.shape))
            self._offsets[p.name] = (off, n)
            off += (n + ALIGN - 1) // ALIGN * ALIGN
        self._n_train = 0
        for p in order:
            if p.trainable:
                o, n = self._offsets[p.name]
                self._n_train = max(self._n_train, (o + n + ALIGN - 1) // ALIGN * ALIGN)
        self._n_total = off
        lr_group_of = lr_group_of or (lambda name: 1 if 'dm_decoder' in name else 0)
        # contiguous runs of equal lr group over the trainable range (image model: 2 runs,
        # video model: 3 -- the predictor sits after dm_decoder in checkpoint order)
        runs = []
        for p in order:
            if not p.trainable:
                continue
            g = lr_group_of(p.name)
            o, n = self._offsets[p.name]
            end = (o + n + ALIGN - 1) // ALIGN * ALIGN
            if runs and runs[-1][2] == g:
                runs[-1][1] = end
            else:
                runs.append([o, end, g])
        self._lr_runs = [tuple(r) for r in runs]
        self._group_split = next((r[0] for r in self._lr_runs if r[2] == 1), self._n_train)
        object.__setattr__(self, '_arena', torch.zeros(self._n_total, dtype=torch.float32))
        object.__setattr__(self, '_garena', None)
        object.__setattr__(self, '_shadow', None)
        self._where = {}
        for p in self._spec:
            parts = p.name.split('.')
            node = self
            for depth, part in enumerate(parts[:-1]):
                if not hasattr(node, part):
                    cls = node_classes.get('.'.join(parts[:depth + 1]), _Node)
                    node.add_module(part, cls())
                node = getattr(node, part)
            t = _init_tensor(p, gen, sched)
            if p.init.startswith('buf:'):
                node.register_buffer(parts[-1], t)
            else:
                v = _view(self._arena, self._offsets[p.name][0], p.shape)
                v.copy_(t)
                node.register_parameter(parts[-1], nn.Parameter(v, requires_grad=p.trainable))
                self._where[p.name] = (node, parts[-1])

    # ------------------------------------------------------------------------------------
    def tensors(self):
        """Flat {dotted name: tensor} view over parameters and buffers."""
        out = dict(self.named_parameters())
        out.update(dict(self.named_buffers()))
        return out

    def arena(self):
        return self._arena

    def arena_ranges(self):
        """(first element of lr group 1, n_train, n_total); see lr_runs() for the exact ranges."""
        return self._group_split, self._n_train, self._n_total

    def lr_runs(self):
        """[(lo, hi, lr_group)] contiguous arena ranges of the trainable parameters."""
        return list(self._lr_runs)

    def arena_slice(self, name, arena=None):
        o, n = self._offsets[name]
        return (self._arena if arena is None else arena)[o:o + n]

    def grad_arena(self):
        """fp32 gradient arena over the trainable range; `p.grad` are views into it."""
        if self._garena is None or self._garena.device != self._arena.device:
            g = torch.zeros(self._n_train, dtype=torch.float32, device=self._arena.device)
            object.__setattr__(self, '_garena', g)
            for p in self._spec:
                if p.trainable and not p.init.startswith('buf:'):
                    node, attr = self._where[p.name]
                    getattr(node, attr).grad = _view(g, self._offsets[p.name][0], p.shape)
        return self._garena

    def shadow_arena(self, refresh=False):
        """bf16 copy of the whole parameter arena (same offsets)."""
        if self._shadow is None or self._shadow.device != self._arena.device:
            object.__setattr__(self, '_shadow', torch.empty(self._n_total, dtype=torch.bfloat16,
                                                            device=self._arena.device))
            refresh = True
        if refresh:
            from . import ops
            ops.cast2d(self._arena.view(1, -1), torch.bfloat16, out=self._shadow.view(1, -1))
        return self._shadow

    def _apply(self, fn, recurse=True):
        new = fn(self._arena)
        assert new.dtype == torch.float32, 'master parameters stay fp32 (compute dtype is separate)'
        object.__setattr__(self, '_arena', new)
        object.__setattr__(self, '_garena', None)
        object.__setattr__(self, '_shadow', None)
        for p in self._spec:
            if p.init.startswith('buf:'):
                continue
            node, attr = self._where[p.name]
            param = getattr(node, attr)
            param.data = _view(new, self._offsets[p.name][0], p.shape)
            param.grad = None
        for mod in self.modules():
            for key, buf in mod._buffers.items():
                if buf is not None:
                    mod._buffers[key] = fn(buf)
        return self
