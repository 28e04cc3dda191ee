"""Import harness for the (read-only) reference at /root/reference.

TEST INFRASTRUCTURE, BUILD-CONTAINER ONLY.  Nothing under ``slotdiffusion_amd/``,
``bench.py`` or the ``-m gpu`` tests imports this file: /root/reference does not
exist on the GPU box.  It is used only by ``tools/gen_golden.py`` to produce the
golden vectors under ``tests/golden/``.

The reference depends on packages absent from this image (nerv, torchvision,
lpips, cv2, wandb ...).  We register minimal stand-ins in ``sys.modules``
before importing it.  The nerv stand-ins restate nerv v0.4.0's documented
behaviour (docs/install.md:24 pins the tag); only the plain SA CNN decoder
depends on their arithmetic, the LDM hot path does not.
"""
import importlib
import sys
import types
from unittest import mock

import torch
import torch.nn as nn

REF_ROOT = '/root/reference'


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _BaseModel(nn.Module):

    def load_weight(self, path, strict=True):
        ckp = torch.load(path, map_location='cpu')
        self.load_state_dict(ckp.get('state_dict', ckp), strict=strict)


class _BaseParams:

    def get(self, key, default=None):
        return getattr(self, key, default)


def _act(name):
    return {'relu': nn.ReLU(), '': nn.Identity()}[name]


def _conv_norm_act(i, o, kernel_size, stride=1, dilation=1, groups=1,
                   norm='', act='relu', dim='2d', **kw):
    assert norm == ''
    return nn.Sequential(
        nn.Conv2d(i, o, kernel_size, stride, kernel_size // 2, dilation,
                  groups, bias=True), nn.Identity(), _act(act))


def _deconv_norm_act(i, o, kernel_size, stride=1, dilation=1, groups=1,
                     norm='', act='relu', dim='2d', **kw):
    assert norm == ''
    return nn.Sequential(
        nn.ConvTranspose2d(i, o, kernel_size, stride=stride,
                           padding=kernel_size // 2,
                           output_padding=stride - 1, dilation=dilation,
                           groups=groups, bias=True), nn.Identity(), _act(act))


def _deconv_out_shape(in_size, stride, padding, kernel_size, out_padding,
                      dilation=1):
    return tuple((s - 1) * stride - 2 * padding + dilation *
                 (kernel_size - 1) + out_padding + 1 for s in in_size)


_INSTALLED = False


def install():
    global _INSTALLED
    if _INSTALLED:
        return
    _stub('nerv')
    _stub('nerv.training', BaseModel=_BaseModel, BaseParams=_BaseParams,
          BaseMethod=object, BaseDataModule=object,
          CosineAnnealingWarmupRestarts=object)
    _stub('nerv.models', conv_norm_act=_conv_norm_act,
          deconv_norm_act=_deconv_norm_act,
          deconv_out_shape=_deconv_out_shape)
    _stub('nerv.utils', **{k: None for k in [
        'load_obj', 'dump_obj', 'glob_all', 'mkdir_or_exist', 'AverageMeter',
        'read_all_lines', 'VideoReader', 'save_video', 'strip_suffix',
        'check_file_exist', 'convert4save']})
    for n in ['transformers', 'cv2', 'wandb', 'lpips', 'skimage',
              'skimage.metrics', 'pycocotools', 'pycocotools.coco',
              'pycocotools.mask', 'torchvision', 'torchvision.utils',
              'torchvision.ops', 'torchvision.transforms',
              'torchvision.transforms.functional', 'torchvision.datasets']:
        sys.modules[n] = mock.MagicMock()
    sys.path.insert(0, REF_ROOT)
    _INSTALLED = True


def ref_models(task):
    """task in {'img_based', 'video_based'} -> reference `models` module."""
    install()
    return importlib.import_module(f'slotdiffusion.{task}.models')


def ref_params(task, cfg_dir, cfg_name):
    """Load a reference config class instance (a fresh one per model)."""
    install()
    d = f'{REF_ROOT}/slotdiffusion/{task}/configs/{cfg_dir}'
    if d not in sys.path:
        sys.path.insert(0, d)
    mod = importlib.import_module(cfg_name)
    mod = importlib.reload(mod)   # fresh nested dicts (reference pops keys)
    return mod.SlotAttentionParams()
