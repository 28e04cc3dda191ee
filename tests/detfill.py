"""Deterministic, RNG-portable weight fill shared by tools/gen_golden.py (reference model,
build container) and the tests (oracle dict / product model, any box).

Every tensor is filled from its own CPU ``torch.Generator`` seeded by its position in
``state_dict()`` order, so the reference model and ours receive identical values as long as
key order/shape match (pinned by tests/golden/state_dict_keys.json).  Zero-initialised layers
get non-zero values on purpose: a fresh UNet otherwise outputs exactly 0 and hides bugs.
"""
import math

import torch

_NORM_HINTS = ('norm', '.bn', 'in_layers.0', 'out_layers.0', 'out.0', 'downsample.1',
               'encoder_out_layer.0', 'project_q.0', 'mlp.0')


def det_value(name, shape, index, seed=1234):
    g = torch.Generator().manual_seed(seed * 100003 + index)
    if len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        if name == 'init_latents':
            return torch.randn(shape, generator=g)
        return torch.randn(shape, generator=g) / math.sqrt(fan_in)
    if name.endswith('.weight') and any(h in name for h in _NORM_HINTS):
        return 1. + 0.1 * torch.randn(shape, generator=g)
    return 0.05 * torch.randn(shape, generator=g)


def det_fill_(named_tensors, skip=lambda n: False, seed=1234):
    """In-place fill of an ordered iterable of (name, tensor); returns {name: cpu fp32 copy}."""
    out = {}
    with torch.no_grad():
        for i, (n, t) in enumerate(named_tensors):
            if skip(n):
                out[n] = t.detach().float().cpu().clone()
                continue
            v = det_value(n, tuple(t.shape), i, seed)
            t.copy_(v.to(t.dtype))
            out[n] = v
    return out


def is_buffer_name(n):
    return n.endswith('.grid') or (n.startswith('dm_decoder.') and n.count('.') == 1)


def make_inputs(B, seed=7):
    g = torch.Generator().manual_seed(seed)
    img = (torch.randn(B, 3, 128, 128, generator=g) * 0.5).clamp(-1, 1)
    # smooth blobs so slot attention has structure to segment
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, 128), torch.linspace(-1, 1, 128), indexing='ij')
    for b in range(B):
        for k in range(4):
            cy, cx = torch.rand(2, generator=g) * 1.6 - 0.8
            col = torch.rand(3, generator=g) * 2 - 1
            blob = (((yy - cy) ** 2 + (xx - cx) ** 2) < 0.08).float()
            img[b] = img[b] * (1 - blob) + (col[:, None, None] + 0.1 * img[b]) * blob
    img = img.clamp(-1, 1)
    t = torch.tensor(([37, 812] + [500, 3, 999, 250, 640, 77])[:B])
    noise = torch.randn(B, 3, 32, 32, generator=g)
    x_T = torch.randn(B, 3, 32, 32, generator=g)
    return img, t, noise, x_T
