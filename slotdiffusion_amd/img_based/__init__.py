"""Registry module with the surface of `slotdiffusion.img_based` (scripts/train.py:97-100):
build_dataset / build_model / build_method."""
from ..method import SyntheticDataModule, build_method  # noqa: F401
from ..models import build_model  # noqa: F401


def build_dataset(params, val_only=False):
    """Datasets are out of scope; every config gets the synthetic module."""
    return SyntheticDataModule(params)
