"""Registry module with the surface of `slotdiffusion.video_based` (scripts/train.py:97-100)."""
from ..method import SyntheticDataModule, build_method  # noqa: F401
from ..models import build_model  # noqa: F401


def build_dataset(params, val_only=False):
    """Datasets are out of scope; clips of `n_sample_frames` synthetic frames."""
    return SyntheticDataModule(params, frames=getattr(params, 'n_sample_frames', 3))
