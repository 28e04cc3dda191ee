"""MI355X-native SlotDiffusion hot path."""
import os as _os
import warnings as _warnings


def configure_runtime(graph_queues=2, warn=True):
    """HIP runtime setting of the TRAINING / BENCH entry points (bench.py, Method.fit, GraphedTrainStep);
    importing the package changes nothing.  A captured train step with parallel branches (weight
    gradients on side streams) is mapped onto a pool of hardware queues; with 2 queues the
    cross-queue signalling costs least (same-box A/B of the graphed step: 31.5 / 32.1 ms against
    32.7 / 32.8 ms with the default pool, 34.8 with 3).  The runtime reads the variable when it
    initialises, i.e. at the first device call: call this before any HIP work; a value set by the user
    wins.  -> the value in effect, or None when it is too late to apply one."""
    key = 'DEBUG_HIP_FORCE_GRAPH_QUEUES'
    if key in _os.environ:
        return _os.environ[key]
    import torch
    if torch.cuda.is_initialized():
        if warn:
            _warnings.warn(f'slotdiffusion_amd.configure_runtime(): HIP is already initialised, {key} was not '
                           'applied (call it before the first device call, or export the variable)')
        return None
    _os.environ[key] = str(graph_queues)
    return _os.environ[key]
