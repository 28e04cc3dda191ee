"""`slotdiffusion.img_based` registry surface (build_dataset / build_model / build_method).

`slotdiffusion` is deliberately a NAMESPACE package here (no __init__.py), exactly like the reference's
(`scripts/train.py:97` does `importlib.import_module(f'slotdiffusion.{args.task}')`): this directory
only contributes the task modules, which re-export the MI355X implementation (`slotdiffusion_amd`)."""
from slotdiffusion_amd.img_based import build_dataset, build_method, build_model  # noqa: F401
