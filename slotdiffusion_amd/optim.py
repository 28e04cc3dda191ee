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
                 total_steps=None, warmup_pct=0.05, min_lr_ratio=0.0):
        self.model = model
        self.lr, self.dec_lr = lr, (dec_lr if dec_lr is not None else lr)
        self.clip = clip_grad
        # floor of the schedule as a fraction of each group's max lr: 0 for the diffusion models
        # (img_based/method.py:277-283, video_based/method.py:186-194, 331-339), 1/100 for the base
        # method SA / SAVi / VQ-VAE stage 1 train with (img_based/method.py:69-85, video_based/method.py:86-96)
        self.min_lr_ratio = float(min_lr_ratio)
        self.b1, self.b2, self.eps = betas[0], betas[1], eps
        self.split, self.n_train, _ = model.arena_ranges()
        dev = model.arena().device
        self.m = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.n_train, dtype=torch.float32, device=dev)
        self.nblk = 1024
        self.partial = torch.empty(self.nblk, dtype=torch.float32, device=dev)
        self.step_count = 0
        # device-resident step counter / learning rates so a captured HIP graph stays valid
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_dev = torch.tensor([self.lr, self.dec_lr], dtype=torch.float32, device=dev)
        self._lr_ring = torch.empty(64, 2, dtype=torch.float32)
        self._lr_ev = [None] * 64             # copy-done event per ring slot
        if dev.type == 'cuda':
            self._lr_ring = self._lr_ring.pin_memory()
        # warm-up length stays a float, as the reference passes `warmup_steps_pct * total_steps`
        self.total_steps, self.warmup = total_steps, (float(warmup_pct * total_steps)
                                                       if total_steps else 0.0)

    def lr_scale(self, done):
        """CosineAnnealingWarmupRestarts(first_cycle_steps=total, min_lr = min_lr_ratio * max_lr, single
        cycle) factor for the optimiser step that follows `done` completed steps -- the scheduler is
        stepped after the optimiser, so the first update runs at min_lr and update k at
        min + (max - min) (k-1)/warmup (nerv's scheduler, recalled: not under /root/reference, see
        DESIGN section 2)."""
        if not self.total_steps:
            return 1.0
        if done < self.warmup:
            f = done / self.warmup
        else:
            span = self.total_steps - self.warmup
            prog = (done - self.warmup) / span if span > 0 else 1.0
            f = 0.5 * (1. + math.cos(math.pi * min(1.0, prog)))
        return self.min_lr_ratio + (1.0 - self.min_lr_ratio) * f

    def zero_grad(self):
        g = self.model.grad_arena()
        if g.is_cuda:
            call('sdmi_memset0', torch.cuda.current_stream().cuda_stream, ptr=_p(g), bytes=g.numel() * 4)
        else:
            g.zero_()

    def state_dict(self):
        """Everything a resumed run needs: moments, step (bias correction + schedule position)."""
        return {'m': self.m.detach().cpu(), 'v': self.v.detach().cpu(), 'step_count': self.step_count,
                'total_steps': self.total_steps, 'warmup': self.warmup, 'min_lr_ratio': self.min_lr_ratio}

    def load_state_dict(self, sd):
        """The reference loads the scheduler state and continues (nerv trainer); so does this.  The schedule
        (total_steps, warmup, min_lr_ratio) is ONE unit: a checkpoint that recorded a schedule (total_steps not None)
        replaces this run's whole triple -- a changed max_epochs or dataset length must not make a checkpoint
        unloadable -- and a checkpoint without one leaves this run's triple alone; every difference is reported.  A
        checkpoint from before the floor was recorded keeps this run's floor."""
        import warnings
        mine = (self.total_steps, self.warmup, self.min_lr_ratio)
        if sd.get('total_steps') is not None:
            theirs = (int(sd['total_steps']), float(sd.get('warmup', 0.0) or 0.0),
                      float(sd['min_lr_ratio']) if 'min_lr_ratio' in sd else self.min_lr_ratio)
            if mine[0] is None or any(abs(float(a) - float(b)) > 1e-9 for a, b in zip(mine, theirs)):
                warnings.warn(f'optimizer checkpoint was written with (total_steps, warmup, min_lr_ratio) = {theirs}, '
                              f'this run was configured with {mine}: continuing the checkpoint\'s schedule')
            self.total_steps, self.warmup, self.min_lr_ratio = theirs
        elif mine[0] is not None:
            warnings.warn(f'optimizer checkpoint carries no learning-rate schedule; keeping this run\'s {mine}')
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])
        self.step_count = int(sd['step_count'])
        self.step_dev.fill_(self.step_count)
        self.set_lr_for_next_step()

    def grad_norm(self):
        """Global L2 norm of the last step's gradients (syncs; diagnostics only).  Under data parallel `partial` holds
        the squared norm of the SUM over ranks (1 / world is applied inside the kernels): scaled back here, so the value
        is the averaged gradient's norm on every world size."""
        return float(self.partial.double().sum().sqrt()) * getattr(self, '_last_gscale', 1.0)

    def set_lr_for_next_step(self):
        """Host side of the schedule: refresh the device lr scalars (outside any graph)."""
        scale = self.lr_scale(self.step_count)
        i = self.step_count % 64
        if self._lr_ev[i] is not None:                  # the slot's previous async copy has landed
            self._lr_ev[i].synchronize()
        slot = self._lr_ring[i]
        slot[0], slot[1] = self.lr * scale, self.dec_lr * scale
        self.lr_dev.copy_(slot, non_blocking=True)
        if self.lr_dev.is_cuda:
            self._lr_ev[i] = torch.cuda.Event()
            self._lr_ev[i].record()

    @torch.no_grad()
    def step(self, capturable=False, grad_src=None, grad_scale=1.0):
        """clip + Adam over the arena.  capturable=True reads step / lr from device memory (the
        caller has run set_lr_for_next_step()), so the launch sequence can live in a HIP graph.
        grad_src / grad_scale (data parallel, parallel.GradReducer): the gradients are read from `grad_src` (the
        fp32 arena, or the bf16 wire buffer with the same offsets) as the all-reduce left them -- the SUM over ranks --
        and multiplied by `grad_scale` = 1 / world inside the kernels; the clip sees the norm of the scaled gradients."""
        self._last_gscale = float(grad_scale)
        m = self.model
        g = m.grad_arena() if grad_src is None else grad_src
        assert g.numel() >= self.n_train and g.dtype in (torch.float32, torch.bfloat16)
        g_dtype = 1 if g.dtype == torch.bfloat16 else 0
        st = torch.cuda.current_stream().cuda_stream
        if capturable:          # host bookkeeping (step_count, lr) is the replaying caller's job
            call('sdmi_counters_inc', st, step=_p(self.step_dev))
        else:
            self.set_lr_for_next_step()
            self.step_count += 1
            self.step_dev.fill_(self.step_count)
        call('sdmi_sqsum_partial', st, g=_p(g), partial=_p(self.partial), n=self.n_train,
             nblk=self.nblk, g_dtype=g_dtype)
        arena = m.arena()
        bf16 = m.compute_dtype == torch.bfloat16
        shadow = m.shadow_arena() if bf16 else None
        for lo, hi, grp in m.lr_runs():
            call('sdmi_adam_clip', st, p=_p(arena[lo:]), g=_p(g[lo:]), m=_p(self.m[lo:]),
                 v=_p(self.v[lo:]), shadow_bf16=(_p(shadow[lo:]) if bf16 else 0),
                 sq_partial=_p(self.partial), nblk=self.nblk, n=hi - lo, lr=0.0,
                 beta1=self.b1, beta2=self.b2, eps=self.eps, clip=self.clip,
                 step=(0 if capturable else self.step_count), lr_dev=_p(self.lr_dev[grp:]),
                 step_dev=(_p(self.step_dev) if capturable else 0), gscale=float(grad_scale), g_dtype=g_dtype)
        m.weights_updated(shadow_fresh=True)


# Captures use the thread-local error mode: with a process group alive, RCCL's watchdog thread
# polls events (hipEventQuery) at any time, which a capture in the default global mode treats as an
# illegal call and is invalidated by.
CAPTURE_MODE = 'thread_local'


def split_runs(model):
    """-> (denoiser runs, encoder runs) of the flat gradient arena: lr group 1 = dm_decoder (97 % of the bytes, reduced
    while the slot encoder's backward runs), everything else follows."""
    runs = model.lr_runs()
    return [(lo, hi) for lo, hi, grp in runs if grp == 1], [(lo, hi) for lo, hi, grp in runs if grp != 1]


def start_reduce_runs(reducer, runs):
    """Start the bucketed all-reduce of the arena ranges `runs` on a parallel.GradReducer -> work handles."""
    works = []
    for lo, hi in runs:
        # >= 4 buckets over the denoiser's range (135 M floats: ~70 MB each on a bf16 wire), one for small runs:
        # the ring of the first bucket is busy while the next one is still being converted
        works += reducer.start(lo, hi, n_buckets=(4 if hi - lo > (16 << 20) else 1))
    return works


class GraphedTrainStep:
    """zero-grad -> forward -> loss -> backward -> clip+Adam captured once into HIP graphs and
    replayed per step (about 1.5k kernel launches per replay instead of as many host launches).
    The draws of a step -- t, noise (`sdmi_draw_tn`) and the dropout masks -- come from the library's
    counter-based generator keyed on the device word `model.step_seed`, which the captured forward
    advances itself; Adam's step / lr are device scalars -- nothing host-side is baked in.

    world == 1: one graph for forward + backward, one for the update.
    world  > 1 (`allreduce` given): the backward is split at the slots -- graph A = forward + loss +
    the denoiser's backward (97 % of the gradient bytes), then the all-reduce of that gradient
    range starts on the collective stream while graph B (the slot encoder's backward: the
    64-channel convolutions at full resolution) runs; the small encoder range follows, then the
    update graph.  `allreduce` may be the legacy callable (whole arena, no overlap) or True."""

    def __init__(self, model, opt, example_batch, allreduce=None, loss_key='denoise_loss',
                 loss_weight=1.0, world=None, wire=None):
        self.model, self.opt, self.allreduce = model, opt, allreduce
        self.loss_key, self.loss_weight = loss_key, loss_weight
        from . import configure_runtime
        configure_runtime(warn=False)      # (normally too late here: the entry points call it first)
        self.static = {k: v.clone() for k, v in example_batch.items()}
        dev = model.arena().device
        if getattr(model, 'step_seed', None) is None or model.step_seed.device != dev:
            model.step_seed = torch.zeros(1, dtype=torch.int64, device=dev)   # dropout seed word
        self.world = world
        if allreduce is not None and world is None:
            import torch.distributed as dist
            self.world = dist.get_world_size()
        # `allreduce=True`: the exchange is a parallel.GradReducer (wire = 'fp32' unless the caller opts into 'bf16');
        # the update graph reads the summed gradients where the collective left them, scaled by 1 / world in-kernel
        self.reducer = None
        if allreduce is True:
            from . import parallel
            self.reducer = parallel.GradReducer(model.grad_arena(), self.world, wire)
        # overlap mode needs the denoiser / encoder split of the arena (lr group 1 = dm_decoder)
        self.dec_runs, self.enc_runs = split_runs(model)
        self.overlap = allreduce is True and len(self.dec_runs) > 0 and len(self.enc_runs) > 0
        if allreduce is True and not self.overlap:        # no denoiser range to split at: one exchange of the arena
            self.allreduce = allreduce = lambda g: self.reducer.reduce_all(4)
        self.loss = None
        self._slots = self._dslots = None
        # The warm-up passes below (lazy operands, func attributes, allocator pool) are real steps on
        # the example batch: weights, moments, step counters and the dropout seed word are
        # snapshotted and rolled back, so a captured run walks the same trajectory as an eager one.
        snap = (model.arena().detach().clone(), opt.m.clone(), opt.v.clone(), opt.step_count,
                opt.step_dev.clone(), model.step_seed.clone())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                      # warm-up: lazy operands, func attributes
                opt.set_lr_for_next_step()
                opt.step_count += 1
                if self.overlap:
                    self._fwd_bwd_denoiser()
                    works = self._start_reduce(self.dec_runs)
                    self._bwd_encoder()
                    works += self._start_reduce(self.enc_runs)
                    self._finish_reduce(works)
                else:
                    self._fwd_bwd()
                    if allreduce is not None:
                        allreduce(model.grad_arena())
                self._update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            model.arena().copy_(snap[0])
            opt.m.copy_(snap[1])
            opt.v.copy_(snap[2])
            opt.step_count = snap[3]
            opt.step_dev.copy_(snap[4])
            model.step_seed.copy_(snap[5])
        model.weights_updated()
        del snap
        torch.cuda.synchronize()
        self.g_fb = torch.cuda.CUDAGraph()
        self.g_enc = None
        if self.overlap:
            with torch.cuda.graph(self.g_fb, capture_error_mode=CAPTURE_MODE):
                self.loss = self._fwd_bwd_denoiser()
            self.g_enc = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_enc, pool=self.g_fb.pool(), capture_error_mode=CAPTURE_MODE):
                self._bwd_encoder()
        else:
            with torch.cuda.graph(self.g_fb, capture_error_mode=CAPTURE_MODE):
                self.loss = self._fwd_bwd()
        self.g_up = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_up, pool=self.g_fb.pool(), capture_error_mode=CAPTURE_MODE):
            self._update()

    # -- pieces ---------------------------------------------------------------------------
    def _forward_loss(self):
        m = self.model
        self.opt.zero_grad()             # (the model's forward starts a new dropout step itself)
        out = m(self.static)
        loss = m.calc_train_loss(self.static, out)[self.loss_key]
        if self.loss_weight != 1.0:
            loss = loss * self.loss_weight
        return out, loss

    def _fwd_bwd(self):
        _, loss = self._forward_loss()
        loss.backward()
        return loss.detach()

    def _fwd_bwd_denoiser(self):
        out, loss = self._forward_loss()
        slots = out['slots']
        dslots, _ = torch.autograd.grad(loss, [slots, self.model.bank().anchor_dec], allow_unused=True)
        self._slots, self._dslots = slots, dslots
        return loss.detach()

    def _bwd_encoder(self):
        torch.autograd.backward([self._slots], [self._dslots])
        self._slots = self._dslots = None

    def _start_reduce(self, runs):
        return start_reduce_runs(self.reducer, runs)

    def _finish_reduce(self, works):
        self.reducer.finish(works)          # (no pass over the arena: 1 / world and the wire dtype ride in _update)

    def _update(self):
        if self.reducer is not None:
            self.opt.step(capturable=True, grad_src=self.reducer.grad_src, grad_scale=self.reducer.grad_scale)
        else:
            self.opt.step(capturable=True)

    def __call__(self, batch):
        for k, v in batch.items():
            self.static[k].copy_(v, non_blocking=True)
        self.opt.set_lr_for_next_step()
        self.opt.step_count += 1
        self.g_fb.replay()
        if self.overlap:
            works = self._start_reduce(self.dec_runs)
            self.g_enc.replay()
            works += self._start_reduce(self.enc_runs)
            self._finish_reduce(works)
        elif self.allreduce is not None:
            self.allreduce(self.model.grad_arena())
        self.g_up.replay()
        return self.loss
