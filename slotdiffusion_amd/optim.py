"""Fused optimiser over the flat parameter arena (replaces torch.optim.Adam + clip_grad_norm_ as
configured by the reference's Method._configure_optimizers, img_based/method.py:235-285):
Adam (no weight decay), two contiguous lr groups (slot encoder: lr, dm_decoder: dec_lr), global
gradient-norm clip, cosine schedule with linear warm-up stepped per iteration.

One `sdmi_sqsum_partial` launch produces the global squared norm of the whole gradient arena and
two `sdmi_adam_clip` launches (one per lr group) update master weights, moments and -- for the bf16
compute path -- the bf16 shadow arena in the same pass.
"""
import math

import torch

from ._lib import call


def _p(t):
    return 0 if t is None else t.data_ptr()


class FusedAdam:
    def __init__(self, model, lr=1e-4, dec_lr=None, clip_grad=1.0, betas=(0.9, 0.999), eps=1e-8,
                 total_steps=None, warmup_pct=0.05):
        self.model = model
        self.lr, self.dec_lr = lr, (dec_lr if dec_lr is not None else lr)
        self.clip = clip_grad
        self.b1, self.b2, self.eps = betas[0], betas[1], eps
        self.split, self.n_train, _ = model.arena_ranges()
        dev = model.arena().device
        self.m = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.nblk = 1024
        self.partial = torch.empty(self.nblk, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.total_steps, self.warmup = total_steps, (int(warmup_pct * total_steps)
                                                       if total_steps else 0)

    def lr_scale(self, it):
        """CosineAnnealingWarmupRestarts(min_lr=0) factor at iteration `it` (single cycle)."""
        if not self.total_steps:
            return 1.0
        if it < self.warmup:
            return it / max(1, self.warmup)
        prog = (it - self.warmup) / max(1, self.total_steps - self.warmup)
        return 0.5 * (1. + math.cos(math.pi * min(1.0, prog)))

    def zero_grad(self):
        self.model.grad_arena().zero_()

    def grad_norm(self):
        """Global L2 norm of the last step's gradients (syncs; diagnostics only)."""
        return float(self.partial.double().sum().sqrt())

    @torch.no_grad()
    def step(self):
        m = self.model
        g = m.grad_arena()
        st = torch.cuda.current_stream().cuda_stream
        self.step_count += 1
        call('sdmi_sqsum_partial', st, g=_p(g), partial=_p(self.partial), n=self.n_train,
             nblk=self.nblk)
        scale = self.lr_scale(self.step_count)
        arena = m.arena()
        bf16 = m.compute_dtype == torch.bfloat16
        shadow = m.shadow_arena() if bf16 else None
        for lo, hi, grp in m.lr_runs():
            lr = self.dec_lr if grp == 1 else self.lr
            call('sdmi_adam_clip', st, p=_p(arena[lo:]), g=_p(g[lo:]), m=_p(self.m[lo:]),
                 v=_p(self.v[lo:]), shadow_bf16=(_p(shadow[lo:]) if bf16 else 0),
                 sq_partial=_p(self.partial), nblk=self.nblk, n=hi - lo, lr=lr * scale,
                 beta1=self.b1, beta2=self.b2, eps=self.eps, clip=self.clip, step=self.step_count)
        m.weights_updated(shadow_fresh=True)
