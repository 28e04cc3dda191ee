"""Kernel-selection policy in ONE place.

Every switch of the product path is a row of `SWITCHES`: name -> (default, what it selects).  The environment can
override a row for an A/B run (`SDMI_<NAME>=0/1`, read once at import); nothing else in the package reads the
environment for kernel selection.  Measured choices that have no switch any more (the losing side was deleted) are
listed in DESIGN.md's negatives ledger with their numbers."""
import os

SWITCHES = {
    # --- inference (sampler) fusions
    'ST_FUSED':      (1, 'fused SpatialTransformer block at inference (sdmi_st_block: 2 launches instead of 10-13)'),
    'ST_FF_SPLIT':   (1, 'feed-forward of that block split over workgroup pairs where its grid leaves half the chip idle'),
    'CROSS_FOLD':    (1, 'slot cross-attention folded into per-image projection weights (<= 16 slots)'),
    'LN_FOLD':       (1, 'LayerNorm folded into the linear layer behind it (bf16 inference)'),
    'RES_MERGE':     (1, 'ResBlock out_layers.3 + 1x1 skip convolution as one implicit GEMM'),
    'FF_MERGE':      (1, 'ff.net.2 + proj_out as one GEMM over [g | tok] with pre-multiplied weights'),
    'UPS_PARITY':    (1, 'upsample convolutions as four 2x2 parity convolutions in one launch'),
    'DEFER_SPLITK':  (1, 'split-K second stage finished by the GroupNorm behind the convolution'),
    'LAZY_CAT':      (1, 'UNet skip concatenations read in place by their two consumers'),
    # --- training
    'ST_TRAIN':      (1, 'fused training forward of the SpatialTransformer block (sdmi_st_train_fwd)'),
    'ST_TRAIN_BWD':  (1, 'fused backward data path of that block (sdmi_st_train_bwd)'),
    'ST_WGRAD_GROUP': (1, "the block's eight weight gradients as two grouped launches"),
    'BWD_PAIR':      (1, 'data + weight gradient of a conv / linear layer in one launch (sdmi_bwd_pair)'),
    'WGRAD_STREAM':  (1, 'stand-alone weight gradients on side HIP streams'),
    'WGRAD_HALO':    (1, 'direct 3x3 weight gradient on channel pairs where the layer leaves the pair launch'),
    'GEGLU_FUSE':    (1, 'GEGLU in the epilogue of the ff.net.0.proj GEMM (per-layer training path)'),
    'DEFER_COLSUM':  (1, 'dgamma / dbeta column sums of the whole step folded by a few grouped launches'),
    # --- debugging
    'DEBUG_DEFER':   (0, 'poison the outputs of deferred split-K launches until their second stage ran (tests)'),
}


def flag(name):
    default = SWITCHES[name][0]
    v = os.environ.get('SDMI_' + name)
    return default if v is None else int(v)
