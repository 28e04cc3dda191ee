"""Every in-scope reference config builds through the registry under the reference's package name.

tests/golden/configs/*.json hold the VALUES of the reference's own config files (dumped by
tools/dump_ref_configs.py, which loads the unmodified files through the BaseParams shim of
slotdiffusion_amd.compat).  `slotdiffusion.<task>.build_model(params)` -- the call scripts/train.py:97-100
makes -- must build each of them, with the checkpoint key set of the reference where a key fixture
exists; configs outside the hot path (the plain nerv CNN encoder of MOVi-Solid / MOVi-Tex) must refuse with
a message that names the scope decision."""
import glob
import importlib
import json
import os

import pytest

from slotdiffusion_amd import compat
from tests import common as C

CFG = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'configs', '*.json')))
OUT_OF_SCOPE = ('movisolid', 'movitex')      # plain nerv CNN encoder


def _load(path):
    d = json.load(open(path))
    return d, compat.Params(**d)


def test_all_reference_configs_are_dumped():
    assert len(CFG) == 24
    models = {json.load(open(f))['model'] for f in CFG}
    assert models == {'SA', 'SADiffusion', 'SAVi', 'SAViDiffusion', 'VQVAE'}


@pytest.mark.parametrize('path', CFG, ids=[os.path.basename(f)[:-5] for f in CFG])
def test_build_model_from_reference_config(path):
    d, P = _load(path)
    task = importlib.import_module('slotdiffusion.' + d['_task'])      # the reference's package name
    name = os.path.basename(path)
    if any(k in name for k in OUT_OF_SCOPE) and d['model'] != 'VQVAE':
        with pytest.raises((AssertionError, NotImplementedError), match='hot path'):
            task.build_model(P)
        return
    m = task.build_model(P)
    assert type(m).__name__ == d['model']
    n = sum(p.numel() for p in m.parameters())
    expect = {'SA': (4.5e6, 5.5e6), 'SAVi': (4.5e6, 5.5e6), 'VQVAE': (13.8e6, 13.9e6),
              'SADiffusion': (151e6, 152e6), 'SAViDiffusion': (152e6, 153e6)}[d['model']]
    if 'dino' in name:        # frozen DINO ViT-S/8 (21.8 M) instead of the ResNet-18 (2.8 M); 256-d slots
        expect = (170e6, 176e6)
        keys_ = set(m.state_dict().keys())
        assert 'encoder.dino.embeddings.cls_token' in keys_
        assert 'encoder.dino.encoder.layer.11.attention.attention.value.bias' in keys_
        assert all(not p.requires_grad for k, p in m.named_parameters() if k.startswith('encoder.dino.'))
        assert tuple(m.visual_resolution) == (28, 28) and tuple(m.latent_res) == (56, 56)
    assert expect[0] < n < expect[1], n
    if d['model'] in ('SADiffusion', 'SAViDiffusion', 'SA', 'SAVi'):
        assert m.num_slots == d['slot_dict']['num_slots']
        assert tuple(m.resolution) == tuple(d['resolution'])
    # checkpoint keys: the fixtures were captured with 7 (img) / 15 (video) slots; key NAMES do not
    # depend on the slot count
    keys = C.load_keys()
    ref = {'SADiffusion': 'img_based/SADiffusion/clevrtex-7slot', 'SA': 'img_based/SA/clevrtex-7slot',
           'SAViDiffusion': 'video_based/SAViDiffusion/movie-15slot'}.get(d['model'])
    if ref is not None and 'dino' not in name:
        want = {e[0] for e in keys[ref]['state']}
        assert set(m.state_dict().keys()) == want


def test_unmodified_reference_style_config_file_loads(tmp_path):
    """A config FILE in the reference's style (`from nerv.training import BaseParams`, class
    SlotAttentionParams) loads through compat.load_params without nerv installed, and the resulting
    object drives build_model / build_dataset / build_method like scripts/train.py does."""
    src = tmp_path / 'my_sa_params-res128.py'
    d = json.load(open([f for f in CFG if 'img_based__sa_clevrtex' in f][0]))
    body = ['from nerv.training import BaseParams', '', '', 'class SlotAttentionParams(BaseParams):']
    for k, v in d.items():
        if not k.startswith('_'):
            body.append(f'    {k} = {v!r}')
    src.write_text('\n'.join(body) + '\n')
    P = compat.load_params(str(src))
    assert P.model == 'SA' and P.get('nonexistent', 5) == 5 and P.slot_dict['num_slots'] == d['slot_dict']['num_slots']
    task = importlib.import_module('slotdiffusion.img_based')
    m = task.build_model(P)
    dm = task.build_dataset(P)
    meth = task.build_method(model=m, datamodule=dm, params=P, ckp_path=None, local_rank=0, use_ddp=False,
                             use_fp16=False)
    opt = meth._configure_optimizers() if False else None          # (needs the device arena: GPU tests)
    assert opt is None and len(dm) > 0
