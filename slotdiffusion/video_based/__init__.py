"""`slotdiffusion.video_based` registry surface (build_dataset / build_model / build_method)."""
from slotdiffusion_amd.video_based import build_dataset, build_method, build_model  # noqa: F401
