"""Loading the reference's own `*_params.py` config files without `nerv`.

The reference's configs are classes deriving from `nerv.training.BaseParams` (e.g.
img_based/configs/sa_ldm/sa_ldm_clevrtex_params-res128.py:1-4) and `scripts/train.py:104-108` loads
them with importlib.  nerv itself (trainer, wandb logging, Slurm glue) is out of scope; what a config
file needs from it is only the `BaseParams` base class: attribute access plus `.get(key, default)`.
`install_nerv_shim()` registers that class under `nerv.training` when the real package is absent, so
an unmodified reference config file imports here, and `load_params(path)` is the loader train.py uses.
"""
import importlib
import importlib.util
import os
import sys
import types


class BaseParams:
    """Attribute container of nerv's BaseParams (recalled API: attribute access, `get`, `to_dict`)."""

    def get(self, key, default=None):
        return getattr(self, key, default)

    def to_dict(self):
        out = {}
        for k in dir(self):
            if k.startswith('_'):
                continue
            v = getattr(self, k)
            if not callable(v):
                out[k] = v
        return out

    def __repr__(self):
        return f'{type(self).__name__}({self.to_dict()})'


def install_nerv_shim():
    """Make `from nerv.training import BaseParams` work (no-op when nerv is installed)."""
    try:
        importlib.import_module('nerv.training')
        return False
    except ImportError:
        pass
    nerv = sys.modules.get('nerv') or types.ModuleType('nerv')
    training = types.ModuleType('nerv.training')
    training.BaseParams = BaseParams
    nerv.training = training
    sys.modules['nerv'] = nerv
    sys.modules['nerv.training'] = training
    return True


def load_params(path, cls='SlotAttentionParams'):
    """scripts/train.py:104-108: import the config module at `path` and instantiate its params class."""
    install_nerv_shim()
    if path.endswith('.py'):
        path = path[:-3]
    name = os.path.basename(path)
    spec = importlib.util.spec_from_file_location(name.replace('-', '_'), path + '.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return getattr(mod, cls)()


class Params(BaseParams):
    """Params object from plain values (e.g. the JSON dumps under tests/golden/configs)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)
