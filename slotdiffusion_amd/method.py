"""Training-method stand-in behind the reference's plugin API
(`task.build_method(model=, datamodule=, params=, ckp_path=, local_rank=, use_ddp=, use_fp16=)`,
scripts/train.py:65-76; optimiser groups and schedule of img_based/method.py:235-285 /
video_based/method.py:291-341; loss weighting `params.<loss>_w`, clipping, AMP switch and the
per-step hook as nerv's BaseMethod does them).

The reference delegates the loop to nerv's trainer (logging, wandb, Slurm requeue ...), which is out
of scope; this class keeps the part of it that IS the hot path: one process per
GPU, fused clip+Adam over the flat arena, gradient all-reduce of the arena over RCCL, optional HIP
graph replay of the whole step.
"""
import os

import torch

from . import parallel
from .optim import FusedAdam, GraphedTrainStep


class SyntheticDataModule:
    """`build_dataset` stand-in: seeded random images in [-1, 1] of the configured shape
    (datasets themselves are out of scope; SURVEY section 8(d) defines this synthetic input)."""

    def __init__(self, params, steps_per_epoch=8, frames=None, device='cuda', seed=1234):
        self.params = params
        self.steps_per_epoch = steps_per_epoch
        self.frames = frames
        self.device = device
        self.seed = seed
        self.rank = int(os.environ.get('RANK', 0))

    def __len__(self):
        return self.steps_per_epoch

    def train_loader(self, epoch=0):
        B = self.params.train_batch_size
        res = self.params.resolution
        H, W = (res, res) if isinstance(res, int) else res
        g = torch.Generator().manual_seed(self.seed + 1000 * epoch + self.rank)
        for _ in range(self.steps_per_epoch):
            shape = (B, 3, H, W) if self.frames is None else (B, self.frames, 3, H, W)
            yield {'img': (torch.randn(shape, generator=g) * 0.5).clamp_(-1, 1).to(self.device)}


class Method:
    """fit() = epochs x steps of {forward, weighted loss, backward, (all-reduce), clip+Adam, hook}."""

    def __init__(self, model, datamodule, params, ckp_path=None, local_rank=0, use_ddp=False,
                 use_fp16=False):
        self.model, self.datamodule, self.params = model, datamodule, params
        self.ckp_path, self.local_rank, self.use_ddp = ckp_path, local_rank, use_ddp
        from . import configure_runtime
        # effective when the method is built before the first device call; INTEGRATION.md: call
        # slotdiffusion_amd.configure_runtime() before build_model(...).cuda() -- one warning when it is too late
        configure_runtime(warn=not getattr(Method, '_warned_runtime', False))
        Method._warned_runtime = True
        if use_fp16:                      # the reference's --fp16 switch = our bf16 compute path
            model.set_compute_dtype('bf16')
        self.world = torch.distributed.get_world_size() if use_ddp else 1
        self.it = 0
        self.use_graph = os.environ.get('SDMI_GRAPH', '1') != '0'
        self.optimizer = None
        self.history = []

    def _get(self, key, default=None):
        p = self.params
        return p.get(key, default) if hasattr(p, 'get') else getattr(p, key, default)

    # img_based/method.py:235-285 (SA / SADiffusion), video_based/method.py:291-341
    def _configure_optimizers(self):
        p = self.params
        assert p.optimizer.lower() in ('adam', 'adamw') and p.weight_decay == 0., \
            'the path covers Adam without weight decay (every shipped config)'
        total = p.max_epochs * len(self.datamodule)
        clip = self._get('clip_grad', 0) or 0
        clip = clip if clip > 0 else 0.0
        # schedule floor: the diffusion methods anneal to 0 (img_based/method.py:277-283,
        # video_based/method.py:186-194 / 331-339); SA / SAVi / VQ-VAE stage 1 run on the base
        # method's schedule, min_lr = lr / 100 (img_based/method.py:69-85, video_based/method.py:86-96)
        from .models import SADiffusion, SAViDiffusion
        diffusion = isinstance(self.model, (SADiffusion, SAViDiffusion))
        return FusedAdam(self.model, lr=p.lr, dec_lr=self._get('dec_lr', p.lr), clip_grad=clip,
                         total_steps=total, warmup_pct=p.warmup_steps_pct,
                         min_lr_ratio=(0.0 if diffusion else 0.01))

    def _loss(self, batch):
        out = self.model(batch)
        losses = self.model.calc_train_loss(batch, out)
        total = None
        for k, v in losses.items():
            w = self._get(f'{k}_w', 1.0)
            total = v * w if total is None else total + v * w
        return total, losses

    def _eager_step(self, batch):
        self.optimizer.zero_grad()
        total, losses = self._loss(batch)
        total.backward()
        if self.world > 1:
            if getattr(self, '_reducer', None) is None:
                self._reducer = parallel.GradReducer(self.model.grad_arena(), self.world, self._get('ddp_grad_dtype'))
            r = self._reducer.reduce_all(4)
            self.optimizer.step(grad_src=r.grad_src, grad_scale=r.grad_scale)
        else:
            self.optimizer.step()
        return total.detach()

    def fit(self, resume_from='', san_check_val_step=0, max_steps=None):
        ckp = None
        if resume_from:
            ckp = torch.load(resume_from, map_location='cpu')
            self.model.load_state_dict(ckp.get('state_dict', ckp))
        self.model.train()
        if self.use_ddp:
            parallel.broadcast_parameters(self.model.arena())
        self.optimizer = self._configure_optimizers()
        if ckp is not None:
            self._restore_training_state(ckp)
        graphed = None
        steps_per_epoch = len(self.datamodule)
        first_epoch = self.it // max(1, steps_per_epoch)
        # a mid-epoch checkpoint resumes BEHIND the batches it already consumed: the run then sees the
        # same batch sequence, and executes exactly max_epochs * steps_per_epoch steps in total
        skip = self.it - first_epoch * steps_per_epoch
        for epoch in range(first_epoch, self.params.max_epochs):
            for bi, batch in enumerate(self.datamodule.train_loader(epoch)):
                if epoch == first_epoch and bi < skip:
                    continue
                if max_steps is not None and self.it >= max_steps:
                    return self
                if self.use_graph and graphed is None and len(self._loss_names()) == 1:
                    ar = True if self.world > 1 else None      # overlapped gradient all-reduce
                    key = self._loss_names()[0]
                    # (the capture's warm-up passes are rolled back: optim.GraphedTrainStep)
                    graphed = GraphedTrainStep(self.model, self.optimizer, batch, allreduce=ar,
                                               loss_key=key, loss_weight=self._get(f'{key}_w', 1.0),
                                               world=self.world, wire=self._get('ddp_grad_dtype'))
                loss = graphed(batch) if graphed is not None else self._eager_step(batch)
                self.it += 1
                self.history.append(loss)
                self.model._training_step_end(self)
                if max_steps is not None and self.it >= max_steps:
                    return self
        return self

    def _loss_names(self):
        from .models import SA, VQVAE
        if isinstance(self.model, VQVAE):        # three weighted terms: eager steps (vqvae/loss.py)
            return ['quant_loss', 'recon_loss', 'percept_loss']
        return ['img_recon_loss'] if isinstance(self.model, SA) else ['denoise_loss']

    def save(self, path):
        """Checkpoint = weights + everything a resumed run needs to continue the same trajectory:
        Adam moments, step (bias correction and schedule position), iteration, EMA shadow."""
        ckp = {'state_dict': self.model.state_dict(), 'it': self.it}
        if self.optimizer is not None:
            ckp['optimizer'] = self.optimizer.state_dict()
        dm = getattr(self.model, 'dm_decoder', None)
        if dm is not None and getattr(dm, 'use_ema', False):
            ckp['ema'] = {'shadow': dm._ema_shadow.detach().cpu(), 'num_updates': dm.ema_num_updates,
                          'decay': dm.ema_decay}
        seed = getattr(self.model, 'step_seed', None)
        if seed is not None:
            ckp['step_seed'] = int(seed)
        eseed = getattr(self.model, 'eval_seed', None)       # validation t / noise draws continue too
        if eseed is not None:
            ckp['eval_seed'] = int(eseed)
        if self.model.arena().is_cuda:           # t / noise draws continue the same stream
            ckp['cuda_rng_state'] = torch.cuda.get_rng_state(self.model.arena().device)
        torch.save(ckp, path)

    def _restore_training_state(self, ckp):
        self.it = int(ckp.get('it', 0))
        if 'optimizer' in ckp:
            self.optimizer.load_state_dict(ckp['optimizer'])
        dm = getattr(self.model, 'dm_decoder', None)
        if dm is not None and 'ema' in ckp:
            e = ckp['ema']
            dm.enable_ema(e['decay'], use_num_updates=e['num_updates'] >= 0)
            dm._ema_shadow.copy_(e['shadow'])
            dm.ema_num_updates = e['num_updates']
        if 'cuda_rng_state' in ckp and self.model.arena().is_cuda:
            torch.cuda.set_rng_state(ckp['cuda_rng_state'], self.model.arena().device)
        if 'step_seed' in ckp:
            dev = self.model.arena().device
            self.model.step_seed = torch.full((1,), ckp['step_seed'], dtype=torch.int64, device=dev)
        if 'eval_seed' in ckp:
            dev = self.model.arena().device
            self.model.eval_seed = torch.full((1,), ckp['eval_seed'], dtype=torch.int64, device=dev)


def build_method(**kwargs):
    """Same call signature as the reference's task.build_method."""
    return Method(**kwargs)
