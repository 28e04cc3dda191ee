"""Model classes with the reference's registry surface, executed by the HIP engine.

Class / method / key names follow the reference so a `scripts/train.py`-style driver and the
evaluation scripts can switch packages:
  SADiffusion ............ slotdiffusion/img_based/models/sa_diffusion.py:73-246
  LDM (model.dm_decoder) . slotdiffusion/img_based/models/ddpm/ldm.py:18-129, cond_ddpm.py:134-212
  VQVAEWrapper (.vae) .... slotdiffusion/video_based/models/vqvae/VQVAE.py:152-194
Tensors cross this boundary in the reference's layout (NCHW fp32 images / latents, [B,N,D] slots);
inside, everything is NHWC in the compute dtype (fp32 for parity runs, bf16 for throughput).
"""
import contextlib
import copy
import os

import torch

from . import dpm, engine, kern, ops, spec
from .module import FlatModule, _Node

_DTYPES = {'fp32': torch.float32, 'float32': torch.float32, 'bf16': torch.bfloat16,
           'bfloat16': torch.bfloat16}


def default_compute_dtype():
    return _DTYPES[os.environ.get('SDMI_DTYPE', 'fp32').lower()]


class _Owned(_Node):
    """Container node that can reach the root model (kept out of the module tree)."""

    def _bind(self, root):
        object.__setattr__(self, '_root_ref', root)

    @property
    def root(self):
        return self._root_ref


class VQVAEWrapper(_Owned):
    """`model.dm_decoder.vae`: encode / decode / quantize on NCHW fp32 tensors."""

    @torch.no_grad()
    def encode(self, x):
        r = self.root
        z = engine.vae_encode(r.K(), r._to_nhwc(x), r.ed, scale_factor=r.z_scale)
        return ops.nhwc_to_nchw(z, 3)

    @torch.no_grad()
    def decode(self, h, quantize=True):
        r = self.root
        img = engine.vae_decode(r.K(), r._latent_nhwc(h), r.ed, scale_factor=r.z_scale,
                                quantize=quantize)
        return ops.nhwc_to_nchw(img, 3)

    @torch.no_grad()
    def quantize(self, h):
        r = self.root
        _, zq = ops.vq_nearest(r._latent_nhwc(h), r.bank().f(r.vq_key), scale=r.z_scale,
                               want_idx=False)
        return ops.nhwc_to_nchw(zq, 3)

    @torch.no_grad()
    def quantize_indices(self, h):
        r = self.root
        idx, _ = ops.vq_nearest(r._latent_nhwc(h), r.bank().f(r.vq_key), scale=r.z_scale,
                                want_zq=False)
        return idx


class LDM(_Owned):
    """`model.dm_decoder`: denoising loss + DPM-Solver++ sampling of the slot-conditioned LDM."""
    use_ema = False
    cond_stage_key = 'slots'
    clip_denoised = False
    vq_denoised = True
    pred_target = 'eps'

    @property
    def num_timesteps(self):
        return self.betas.shape[0]

    # -- a18: EMA of the denoiser (ddpm/ema.py:29-85, ddpm.py:134-147, 275-278) ---------------
    # The shipped configs have use_ema=False; `enable_ema()` switches the hook on.  The shadow is
    # one flat fp32 copy of the UNet's arena range, updated by a single streaming kernel.
    def _ema_range(self):
        r = self.root
        offs = [r._offsets[n] for n, _ in r.named_parameters() if n.startswith('dm_decoder.model.')]
        lo = min(o for o, _ in offs)
        hi = max(o + c for o, c in offs)
        return lo, hi

    def enable_ema(self, decay=0.9999, use_num_updates=True):
        lo, hi = self._ema_range()
        self.use_ema = True
        self.ema_decay = float(decay)
        self.ema_num_updates = 0 if use_num_updates else -1
        self._ema_shadow = self.root.arena()[lo:hi].detach().clone()
        self._ema_stored = None
        return self

    @contextlib.contextmanager
    def ema_scope(self, context=None):
        """Temporarily run with the EMA weights (store -> copy_to ... restore)."""
        if not self.use_ema:
            yield None
            return
        lo, hi = self._ema_range()
        arena = self.root.arena()
        self._ema_stored = arena[lo:hi].clone()
        with torch.no_grad():
            arena[lo:hi].copy_(self._ema_shadow)
        self.root.weights_updated()
        try:
            yield None
        finally:
            with torch.no_grad():
                arena[lo:hi].copy_(self._ema_stored)
            self._ema_stored = None
            self.root.weights_updated()

    def _training_step_end(self, *a, **k):
        if not self.use_ema:
            return
        # float32 arithmetic like the reference's buffers (ema.py:29-38: `decay` is a float32 tensor,
        # `num_updates` an int tensor), so 1 - decay is the same float32 value
        import numpy as np
        decay = np.float32(self.ema_decay)
        if self.ema_num_updates >= 0:
            self.ema_num_updates += 1
            decay = min(decay, np.float32(1 + self.ema_num_updates) / np.float32(10 + self.ema_num_updates))
        lo, hi = self._ema_range()
        kern.call('sdmi_ema_update', torch.cuda.current_stream().cuda_stream,
                  shadow=self._ema_shadow.data_ptr(), p=self.root.arena()[lo:hi].data_ptr(),
                  n=hi - lo, one_minus_decay=float(np.float32(1.0) - decay))

    # -- a7/a8 ---------------------------------------------------------------------------

    def encode_x0(self, img):
        """Frozen VQ-VAE encode of a batch of images -> x0 [B,h,w,4] fp32 (ldm.py:61-63, under no_grad)."""
        r = self.root
        with torch.no_grad():
            return engine.vae_encode(r.K(), r._to_nhwc(img), r.ed, scale_factor=r.z_scale)

    def loss_function(self, data_dict, t=None, noise=None):
        """ldm.py:59-83; t / noise may be supplied (fixtures) or are drawn like the reference.
        Records the autograd graph (HIP backward kernels) when `slots` requires grad."""
        r = self.root
        img, slots = data_dict['img'], data_dict[self.cond_stage_key]
        B = img.shape[0]
        train_draw = bool(r.training and torch.is_grad_enabled())    # (read before the no_grad block below)
        with torch.no_grad():
            x0 = self.encode_x0(img)
            tf = None
            if t is None and noise is None:
                # the reference's two draws (ldm.py:65-69) + the schedule gathers in ONE launch of
                # the counter-based generator (keyed on the run seed and the per-step seed word, so a
                # graph replay draws new values); explicit t / noise (fixtures) take the path below
                t, tf, ca, cb, nz = self._draw_tn(B, x0.shape[1], x0.shape[2], img.device, train=train_draw)
            else:
                if t is None:
                    t = torch.randint(0, self.num_timesteps, (B,), device=img.device).long()
                if noise is None:
                    noise = torch.randn(B, 3, x0.shape[1], x0.shape[2], device=img.device)
                nz = ops.nchw_to_nhwc(noise, torch.float32, 4)
                ca = self.sqrt_alphas_bar[t].contiguous()
                cb = self.sqrt_one_minus_alphas_bar[t].contiguous()
            xt = ops.row_lincomb(x0, nz, ca, cb)
        grad = torch.is_grad_enabled() and (slots.requires_grad or r.training)
        Kp = r.KG() if grad else r.K()
        with torch.set_grad_enabled(grad):
            pred = r._unet_eps(xt, tf if tf is not None else t.float(), slots, Kp)
            # the 4th (zero pad) channel adds no error but is counted in n: rescale 4/3
            if self.pred_target == 'eps':                      # ldm.py:71-79
                gt = nz
            elif self.pred_target == 'v':                      # v = alpha_t * noise - sigma_t * x0
                gt = ops.row_lincomb(nz, x0, ca, (-cb).contiguous())
            else:
                gt = x0
            if grad:
                loss = kern.MseFn.apply(pred, gt, 4.0 / 3.0)
            else:
                loss = (ops.mse(pred, gt) * (4.0 / 3.0)).reshape(())
        return {'denoise_loss': loss}

    def _draw_tn(self, B, h, w, device, train=None):
        r = self.root
        hw = h * w
        t = torch.empty((B,), dtype=torch.int64, device=device)
        tf, ca, cb = (torch.empty((B,), dtype=torch.float32, device=device) for _ in range(3))
        nz = torch.empty((B, hw, 4), dtype=torch.float32, device=device)
        st = torch.cuda.current_stream().cuda_stream
        salt = 17
        if train is None:
            train = bool(r.training and torch.is_grad_enabled())
        if train:
            # training forward: `_begin_train_forward` advanced the step's seed word already
            seed_dev = getattr(r, 'step_seed', None)
        else:
            # validation / no_grad calls (calc_eval_loss): the reference draws fresh randint / randn for
            # every batch (ldm.py:65-69), so these calls advance a seed word of their own -- every call
            # sees new timesteps and noise, and the training stream is left untouched
            seed_dev = getattr(r, 'eval_seed', None)
            if seed_dev is None or seed_dev.device != device:
                seed_dev = r.eval_seed = torch.zeros(1, dtype=torch.int64, device=device)
            kern.call('sdmi_counters_inc', st, seed=seed_dev.data_ptr())
            salt = 6151
        kern.call('sdmi_draw_tn', st, t=t.data_ptr(),
                  tf=tf.data_ptr(), ca=ca.data_ptr(), cb=cb.data_ptr(), noise=nz.data_ptr(),
                  tab_a=self.sqrt_alphas_bar.data_ptr(), tab_b=self.sqrt_one_minus_alphas_bar.data_ptr(),
                  B=B, T=self.num_timesteps, per=hw * 4,
                  seed=(int(getattr(r, 'seed', 0)) * 7919 + int(os.environ.get('RANK', 0)) * 104729 + salt),
                  seed_dev=(seed_dev.data_ptr() if seed_dev is not None else 0))
        return t, tf, ca, cb, nz.view(B, h, w, 4)

    # -- a12/a13 -------------------------------------------------------------------------
    @torch.no_grad()
    def generate_imgs(self, cond, batch_size=16, ret_intermed=False, verbose=False,
                      use_ddim=False, use_dpm=True, x_T=None, same_noise=False, **kwargs):
        """cond_ddpm.py:134-212: DPM-Solver++ (use_dpm, takes precedence), DDIM (use_ddim) or the
        T-step ancestral sampler.  Returns latents [B,3,h,w] (NCHW fp32)."""
        r = self.root
        if cond.dim() == 2:
            cond = cond.unsqueeze(0).expand(batch_size, -1, -1)
        cond = cond.contiguous()
        h, w = r.latent_res
        if x_T is None:
            if same_noise:
                x_T = torch.randn(1, 3, h, w, device=cond.device).repeat(batch_size, 1, 1, 1)
            else:
                x_T = torch.randn(batch_size, 3, h, w, device=cond.device)
        x = ops.nchw_to_nhwc(x_T, torch.float32, 4)
        if use_dpm:            # cond_ddpm.py:155-178 (takes precedence, as in the reference)
            x, inter = r._dpm_sample(x, cond, ret_intermed)
        elif use_ddim:         # cond_ddpm.py:180-190: DDIM, max(200, T // 5) steps, eta = 0
            steps = kwargs.get('ddim_steps') or max(200, self.num_timesteps // 5)
            x, inter = r._ddim_sample(x, cond, steps, kwargs.get('eta', 0.), ret_intermed,
                                      kwargs.get('log_every_t', 100))
        else:                  # cond_ddpm.py:191-196: ancestral sampling over all T timesteps
            x, inter = r._ancestral_sample(x, cond, ret_intermed, kwargs.get('log_every_t', 100))
        out = ops.nhwc_to_nchw(x, 3)
        if ret_intermed:
            return out, torch.stack([ops.nhwc_to_nchw(i, 3) for i in inter], 0)
        return out

    @torch.no_grad()
    def log_images(self, batch, ret_intermed=False, **kwargs):
        """ldm.py:86-131 with ret_intermed=False (grids need torchvision, out of scope)."""
        cond = batch[self.cond_stage_key]
        B = cond.shape[0]
        ret = self.generate_imgs(cond=cond, batch_size=B, ret_intermed=False, **kwargs)
        return {'samples': self.vae.decode(ret)}


class SlotModelBase(FlatModule):
    """Plumbing shared by the slot models: compute dtype, weight bank / kernel providers,
    checkpoint loading (nerv BaseModel surface: load_weight, device, dtype)."""
    _bank = None
    _Kinf = _Kgrad = None
    compute_dtype = None

    # -- plumbing ------------------------------------------------------------------------
    @property
    def device(self):
        return self.init_latents.device

    @property
    def dtype(self):
        return self.init_latents.dtype

    # 'fp8': bf16 storage with e4m3fn operands on the denoiser's 3x3 convolutions (BASELINE config 5, "fp8 MFMA
    # UNet"): at inference kern.Kern.conv (weight scales taken once, on the host), in training the FORWARD GEMM of
    # those layers (kern.GemmFn.forward: weights re-quantised on the device every step, WeightBank.w8_dev); the
    # backward pass keeps bf16 operands
    fp8_unet = False
    fp8_prefix = 'dm_decoder.model.'

    def set_compute_dtype(self, dt):
        self.fp8_unet = isinstance(dt, str) and dt.lower() == 'fp8'
        if self.fp8_unet:
            dt = 'bf16'
        self.compute_dtype = _DTYPES[dt] if isinstance(dt, str) else dt
        self.invalidate_weights()
        return self

    def invalidate_weights(self):
        self._bank = None
        self._graph_cache = {}

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate_weights()
        return r

    def load_weight(self, path, strict=True):
        ckp = torch.load(path, map_location='cpu')
        self.load_state_dict(ckp.get('state_dict', ckp), strict=strict)

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        self.invalidate_weights()
        return r

    def bank(self):
        if self._bank is None:
            self._bank = kern.WeightBank(self, self.compute_dtype)
            self._Kinf = kern.Kern(self._bank)
            self._Kgrad = kern.KernGrad(self._bank)
            if self.compute_dtype != torch.float32:
                self.shadow_arena(refresh=True)
        return self._bank

    def K(self):
        self.bank()
        return self._Kinf

    def KG(self):
        self.bank()
        return self._Kgrad

    def weights_updated(self, shadow_fresh=False):
        """Call after the master arena changed (optimizer step / checkpoint load)."""
        if self._bank is not None:
            self._bank.invalidate()
            if self.compute_dtype != torch.float32 and not shadow_fresh:
                self.shadow_arena(refresh=True)
        self._graph_cache = {}

    def _to_nhwc(self, img):
        return ops.nchw_to_nhwc(img.float(), self.compute_dtype, ops.vec_of(self.compute_dtype))

    def _begin_train_forward(self):
        """Every training forward (eager, captured or replayed) starts a new dropout step: the
        device seed word advances, so consecutive steps -- and data-parallel ranks -- draw
        different masks."""
        if self.training and torch.is_grad_enabled():
            self.KG().begin_step()

    def _training_step_end(self, method=None):
        pass


class SADiffusion(SlotModelBase):
    """SlotDiffusion on images (registry name 'SADiffusion')."""

    def __init__(self, resolution, slot_dict, enc_dict, dec_dict, loss_dict=None, eps=1e-6,
                 compute_dtype=None, seed=0):
        dec_dict = copy.deepcopy(dec_dict)
        dd = dec_dict['diffusion_dict']
        sp = self._make_spec(resolution, slot_dict, enc_dict, dec_dict)
        sched = {k: dd[k] for k in ('timesteps', 'beta_schedule', 'linear_start', 'linear_end')
                 if k in dd}
        super().__init__(sp, schedule_kwargs=sched, seed=seed,
                         node_classes={'dm_decoder': LDM, 'dm_decoder.vae': VQVAEWrapper})
        assert dd.get('pred_target', 'eps') in ('eps', 'x0', 'v')     # video_based ddpm.py:79
        self.dm_decoder.pred_target = dd.get('pred_target', 'eps')
        self.resolution = tuple(resolution)
        self.eps = eps
        self.slot_dict, self.enc_dict, self.dec_dict = dict(slot_dict), dict(enc_dict), dec_dict
        self.loss_dict = dict(loss_dict or {'use_denoise_loss': True})
        self.num_slots = slot_dict['num_slots']
        self.slot_size = slot_dict['slot_size']
        self.num_iterations = slot_dict['num_iterations']
        self.rplan, self.visual_resolution = spec.encoder_plan(self.resolution, enc_dict)
        self.latent_res = tuple(dec_dict['resolution'])
        self.ed = dec_dict['vae_dict']['enc_dec_dict']
        self.z_scale = float(dd.get('z_scale_factor', 1.))
        self.vq_key = 'dm_decoder.vae.vqvae.quantize.embedding.weight'
        self.unet_cfg = dec_dict['unet_dict']
        self.testing = False
        # ResBlock dropout (unet_dict['dropout']); parity tests switch it off (RNG streams differ)
        self.train_dropout = float(dec_dict['unet_dict'].get('dropout', 0.0))
        self.compute_dtype = compute_dtype or default_compute_dtype()
        self.dm_decoder._bind(self)
        self.dm_decoder.vae._bind(self)
        self._bank = None
        self._unet = None
        self._plan = None
        self._Kinf = self._Kgrad = None
        self.step_seed = None      # device word mixed into dropout seeds (see optim.GraphedTrainStep)
        self.eval_seed = None      # its no_grad / validation counterpart (LDM._draw_tn)
        self.use_graph = os.environ.get('SDMI_GRAPH', '1') != '0'
        self._graph_cache = {}

    def _make_spec(self, resolution, slot_dict, enc_dict, dec_dict):
        return spec.sa_diffusion(resolution, slot_dict, enc_dict, dec_dict)

    def unet(self):
        if self._unet is None:
            self._unet = engine.UNetRunner(self.unet_cfg)
        return self._unet

    def _latent_nhwc(self, z):
        return ops.nchw_to_nhwc(z.float(), torch.float32, 4)

    def _ctx(self, slots, Kp=None):
        return (Kp or self.K()).cast(slots.contiguous().float(), self.compute_dtype)

    def _unet_in(self, x):
        """fp32 latent state [B,h,w,4] -> UNet input in compute dtype with vector-padded channels."""
        if self.compute_dtype == torch.float32:
            return x
        return ops.cast2d(x, self.compute_dtype, cols=3, ldd=ops.vec_of(self.compute_dtype))

    def _unet_eps(self, xt, t, slots, Kp=None):
        Kp = Kp or self.K()
        u = self.unet()
        return u.forward(Kp, self._unet_in(xt), u.time_rowvecs(Kp, t),
                         u.context_kv(Kp, self._ctx(slots, Kp)))

    # -- sampler -------------------------------------------------------------------------
    def _dpm_sample(self, x, cond, ret_intermed=False, steps=None):
        """x [B,h,w,4] fp32 noise -> x_0.  With use_graph the whole 20-NFE loop (~9k kernel
        launches) is captured once per batch size into a HIP graph and replayed."""
        steps = steps or max(20, self.dm_decoder.num_timesteps // 50)
        if ret_intermed or not self.use_graph:
            return self._dpm_loop(x, cond, self._dpm_prepare(steps, x.device), ret_intermed)
        key = (x.shape[0], steps, tuple(cond.shape))
        g = self._graph_cache.get(key)
        if g is None:
            prep = self._dpm_prepare(steps, x.device)
            sx, sc = torch.empty_like(x), torch.empty_like(cond)
            sx.copy_(x)
            sc.copy_(cond)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):            # warm-up: lazy weight prep, func attributes
                self._dpm_loop(sx, sc, prep, False)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                out, _ = self._dpm_loop(sx, sc, prep, False)
            g = self._graph_cache[key] = (graph, sx, sc, out)
        graph, sx, sc, out = g
        sx.copy_(x)
        sc.copy_(cond)
        graph.replay()
        return out, []

    def _ddim_sample(self, x, cond, steps, eta=0., ret_intermed=False, log_every_t=100):
        """DDIMSampler._sample_x0_from_noise (ddim.py:128-218) for eps-prediction with the VQ
        denoiser: per step  eps -> x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t) -> VQ ->
        x = sqrt(a_prev) x0 + sqrt(1 - a_prev - sigma^2) eps (+ sigma * noise)."""
        plan = dpm.ddim_plan(self.dm_decoder.alphas_bar.detach().float().cpu(), steps, eta)
        inter = [x]
        for x, st in self._ddim_steps(x, cond, plan):
            if st['index'] % log_every_t == 0 or st['index'] == len(plan) - 1:
                inter.append(x)
        return x, (inter if ret_intermed else [])

    def _ancestral_sample(self, x, cond, ret_intermed=False, log_every_t=100):
        """_sample_x0_from_noise (cond_ddpm.py:88-120): t = T-1 .. 0."""
        T = self.dm_decoder.num_timesteps
        inter = [x]
        for x, i in self._ancestral_steps(x, cond, list(reversed(range(T)))):
            if i % log_every_t == 0 or i == T - 1:
                inter.append(x)
        return x, (inter if ret_intermed else [])

    def _ancestral_steps(self, x, cond, ts, noises=None):
        """Generator over ancestral updates x_t -> x_{t-1} for the integer timesteps `ts` (any
        list, in sampling order): _p_mean_variance + _p_sample (cond_ddpm.py:55-86, ddpm.py:167-180)
        with the VQ denoiser.  `noises` (list of [B,3,h,w] or None) replaces the random draws."""
        dm = self.dm_decoder
        u = self.unet()
        Kp = self.K()
        tab = {k: getattr(dm, k).detach().double().cpu() for k in
               ('sqrt_recip_alphas_bar', 'sqrt_recipm1_alphas_bar', 'posterior_mean_coef1',
                'posterior_mean_coef2', 'posterior_log_variance_clipped', 'sqrt_alphas_bar',
                'sqrt_one_minus_alphas_bar')}
        ctx_kv = u.context_kv(Kp, self._ctx(cond))
        B = x.shape[0]
        code = self.bank().f(self.vq_key)
        CH = 50                                   # time-embedding rows are prepared 50 steps at a time
        for c0 in range(0, len(ts), CH):
            chunk = ts[c0:c0 + CH]
            tin = torch.tensor([float(t) for t in chunk], dtype=torch.float32, device=x.device)
            rv_all = u.time_rowvecs(Kp, tin)
            for j, t in enumerate(chunk):
                out = u.forward(Kp, self._unet_in(x), rv_all[j:j + 1].expand(B, -1), ctx_kv)
                if dm.pred_target == 'eps':
                    x0 = ops.lincomb(float(tab['sqrt_recip_alphas_bar'][t]), x,
                                     -float(tab['sqrt_recipm1_alphas_bar'][t]), out)
                elif dm.pred_target == 'v':          # cond_ddpm.py:63-67: x0 = alpha_t x - sigma_t v
                    x0 = ops.lincomb(float(tab['sqrt_alphas_bar'][t]), x,
                                     -float(tab['sqrt_one_minus_alphas_bar'][t]), out)
                else:
                    x0 = out
                x0 = ops.vq_nearest(x0, code, scale=self.z_scale, want_idx=False)[1]
                x = ops.lincomb(float(tab['posterior_mean_coef1'][t]), x0,
                                float(tab['posterior_mean_coef2'][t]), x)
                if t != 0:
                    nz = noises[c0 + j] if noises is not None else torch.randn(
                        B, 3, x.shape[1], x.shape[2], device=x.device)
                    nz = ops.nchw_to_nhwc(nz, torch.float32, 4)
                    sd = float(torch.exp(0.5 * tab['posterior_log_variance_clipped'][t].float()))
                    x = ops.lincomb(1.0, x, sd, nz)
                yield x, t

    def _ddim_steps(self, x, cond, plan):
        """Generator over the DDIM updates of `plan` (a list of dpm.ddim_plan entries, any subset in
        sampling order): yields (x after the step, step)."""
        assert self.dm_decoder.pred_target == 'eps', 'the DDIM sampler is defined for eps models'
        u = self.unet()
        Kp = self.K()
        tin = torch.tensor([float(st['t']) for st in plan], dtype=torch.float32, device=x.device)
        ctx_kv = u.context_kv(Kp, self._ctx(cond))
        rv_all = u.time_rowvecs(Kp, tin)
        B = x.shape[0]
        code = self.bank().f(self.vq_key)
        for i, st in enumerate(plan):
            rv = rv_all[i:i + 1].expand(B, -1)
            eps = u.forward(Kp, self._unet_in(x), rv, ctx_kv)
            x0 = ops.lincomb(1.0, x, -st['som'], eps, div=st['sqrt_a'])
            x0 = ops.vq_nearest(x0, code, scale=self.z_scale, want_idx=False)[1]
            x = ops.lincomb(st['sqrt_a_prev'], x0, st['dir'], eps)
            if st['sigma'] != 0.0:
                nz = ops.nchw_to_nhwc(torch.randn(B, 3, x.shape[1], x.shape[2], device=x.device),
                                      torch.float32, 4)
                x = ops.lincomb(1.0, x, st['sigma'], nz)
            yield x, st

    def _dpm_prepare(self, steps, device):
        if self._plan is None or self._plan[0] != steps:
            betas = self.dm_decoder.betas.detach().float().cpu()
            plan = dpm.build_plan(betas, steps=steps, order=3)
            tin = torch.tensor(dpm.plan_t_inputs(plan), dtype=torch.float32, device=device)
            self._plan = (steps, plan, tin)
        return self._plan[1], self._plan[2]

    def _dpm_loop(self, x, cond, prep, ret_intermed):
        plan, tin = prep
        u = self.unet()
        Kp = self.K()
        ctx_kv = u.context_kv(Kp, self._ctx(cond))
        rv_all = u.time_rowvecs(Kp, tin)                     # [NFE, sum Cout] fp32
        B = x.shape[0]
        code = self.bank().f(self.vq_key)
        nfe = [0]
        x_start = self.dm_decoder.pred_target == 'x0'
        v_pred = self.dm_decoder.pred_target == 'v'

        def data_pred(xc, e):
            rv = rv_all[nfe[0]:nfe[0] + 1].expand(B, -1)      # pitch-0 view: same row for all b
            nfe[0] += 1
            if not (x_start or v_pred):           # x0 formed inside the VQ search (SdmiVqArgs.z2)
                # x0 = (x_t - sigma_t eps) / alpha_t is formed inside the VQ search (sdmi.h: SdmiVqArgs.z2): no launch
                # of its own, and the pad channel of eps is never read (no zero fill either)
                eps = u.forward(Kp, self._unet_in(xc), rv, ctx_kv, zero_pad=False)
                return ops.vq_nearest(xc, code, scale=self.z_scale, want_idx=False,
                                      comb=(1.0, -e['sigma'], eps, e['alpha']))[1]
            eps = u.forward(Kp, self._unet_in(xc), rv, ctx_kv)
            if x_start:        # model_wrapper 'x_start' (dpm_solver.py:358-361): output -> noise
                eps = ops.lincomb(1.0, xc, -e['alpha'], eps, div=e['sigma'])
            elif v_pred:       # model_wrapper 'v' (dpm_solver.py:362-365): alpha_t * v + sigma_t * x
                eps = ops.lincomb(e['alpha'], eps, e['sigma'], xc)
            x0 = ops.lincomb(1.0, xc, -e['sigma'], eps, div=e['alpha'])
            return ops.vq_nearest(x0, code, scale=self.z_scale, want_idx=False)[1]

        inter = []
        for st in plan['steps']:
            ev = st['evals']
            m_s = data_pred(x, ev[0])
            if st['order'] >= 2:
                c = st['to_s1']
                x_s1 = ops.lincomb(c['c0'], x, c['c1'], m_s)
                m_s1 = data_pred(x_s1, ev[1])
            if st['order'] == 3:
                c = st['to_s2']
                x_s2 = ops.lincomb(c['c0'], x, c['c1'], m_s, c['c2'], m_s1, m_s)
                m_s2 = data_pred(x_s2, ev[2])
            f = st['final']
            if st['order'] == 1:
                x = ops.lincomb(f['c0'], x, f['c1'], m_s)
            else:
                m_last = m_s1 if f['which'] == 1 else m_s2
                x = ops.lincomb(f['c0'], x, f['c1'], m_s, f['c2'], m_last, m_s)
            if ret_intermed:
                inter.append(x)
        return x, inter

    # -- a1-a4 ---------------------------------------------------------------------------
    def encode(self, img, init_slots=None):
        """sa_diffusion.py:155-183 -> slots [B,N,D] fp32, masks [B,N,h,w] (train) / [B,N,H,W].
        In training mode with grad enabled the autograd graph (HIP backward) is recorded."""
        B, _, H, W = img.shape
        grad = self.training and torch.is_grad_enabled()
        Kp = self.KG() if grad else self.K()
        with torch.set_grad_enabled(grad):
            tok = engine.encoder_out(Kp, self._to_nhwc(img), self.rplan)
            init = self.init_latents[0] if init_slots is None else init_slots.contiguous().float()
            slots, seg = engine.slot_attention(Kp, tok, init, self.num_iterations, self.eps)
        h, w = self.visual_resolution
        with torch.no_grad():
            if not self.training and (h, w) != (H, W):
                masks, _ = ops.mask_upsample_argmax(seg, h, w, H, W)
            else:
                masks = seg.detach().permute(0, 2, 1).reshape(B, self.num_slots, h, w)
        return slots, masks

    def forward(self, data_dict, **kwargs):
        if kwargs.pop('log_images', False):
            return self.log_images(data_dict, **kwargs)
        assert kwargs == {}
        self._begin_train_forward()
        slots, masks = self.encode(data_dict['img'])
        return {'masks': masks, 'slots': slots}

    def calc_train_loss(self, data_dict, out_dict):
        """sa_diffusion.py:206-213."""
        ddpm_dict = {'img': data_dict['img'], 'slots': out_dict['slots']}
        for k in ('t', 'noise'):
            if k in data_dict:
                ddpm_dict[k] = data_dict[k]
        return self.dm_decoder.loss_function(ddpm_dict, t=ddpm_dict.get('t'),
                                             noise=ddpm_dict.get('noise'))

    @torch.no_grad()
    def calc_eval_loss(self, data_dict, out_dict):
        return self.calc_train_loss(data_dict, out_dict)

    @torch.no_grad()
    def log_images(self, data_dict, **kwargs):
        """sa_diffusion.py:228-241: slots -> DPM-Solver++ -> VQ-VAE decode."""
        out_dict = self.forward(data_dict)
        log = self.dm_decoder.log_images({'img': data_dict['img'], 'slots': out_dict['slots']},
                                         **kwargs)
        log['masks'] = out_dict['masks']
        return log

    def _training_step_end(self, method=None):
        self.dm_decoder._training_step_end()


def _encode_clip(self, img, prev_slots=None):
    """Per-frame Slot Attention recurrence shared by the video models (savi.py:366-397,
    savi_diffusion.py:169-216): the initial slots of frame t are the transformer predictor's output
    on frame t-1's slots (the learnt init_latents on the first frame).  -> slots [B,T,N,D],
    masks [B,T,N,h,w] (train) / [B,T,N,H,W] (eval)."""
    B, T, _, H, W = img.shape
    grad = self.training and torch.is_grad_enabled()
    Kp = self.KG() if grad else self.K()
    h, w = self.visual_resolution
    with torch.set_grad_enabled(grad):
        tok = engine.encoder_out(Kp, self._to_nhwc(img.flatten(0, 1)), self.rplan)
        tok = tok.view(B, T, tok.shape[1], tok.shape[2])
        all_s, all_seg = [], []
        for t in range(T):
            if prev_slots is None:
                lat = self.init_latents[0]
            else:
                lat = engine.transformer_predictor(Kp, prev_slots.contiguous(),
                                                   self.pred_dict['pred_num_layers'],
                                                   self.pred_dict['pred_num_heads'])
            tk = tok[:, t].contiguous() if T > 1 else tok[:, 0]
            s, seg = engine.slot_attention(Kp, tk, lat, self.num_iterations, self.eps)
            all_s.append(s)
            all_seg.append(seg)
            prev_slots = s
        slots = kern.StackTimeFn.apply(*all_s)
    with torch.no_grad():
        seg = kern.StackTimeFn.apply(*[x.detach() for x in all_seg])    # [B,T,M,N]
        if not self.training and (h, w) != (H, W):
            masks, _ = ops.mask_upsample_argmax(seg.flatten(0, 1).contiguous(), h, w, H, W)
            masks = masks.view(B, T, self.num_slots, H, W)
        else:
            masks = seg.permute(0, 1, 3, 2).reshape(B, T, self.num_slots, h, w)
    return slots, masks


class SAViDiffusion(SADiffusion):
    """SlotDiffusion on videos (registry name 'SAViDiffusion', video_based/models/
    savi_diffusion.py:74-302): per-frame Slot Attention whose initial slots are the transformer
    predictor's output on the previous frame's slots; the LDM runs on the flattened B*T frames."""

    def __init__(self, resolution, clip_len, slot_dict, enc_dict, dec_dict, pred_dict,
                 loss_dict=None, eps=1e-6, compute_dtype=None, seed=0):
        assert pred_dict.get('pred_type', 'transformer') == 'transformer' and \
            not pred_dict.get('pred_rnn', False), 'hot path covers the transformer predictor'
        self._pred_dict = dict(pred_dict)
        self.clip_len = clip_len
        super().__init__(resolution, slot_dict, enc_dict, dec_dict, loss_dict, eps, compute_dtype,
                         seed)
        self.pred_dict = dict(pred_dict)
        self.pred_dropout = 0.1            # nn.TransformerEncoderLayer default (predictor.py:33-38)

    def _make_spec(self, resolution, slot_dict, enc_dict, dec_dict):
        return spec.savi_diffusion(resolution, slot_dict, enc_dict, dec_dict, self._pred_dict)

    def encode(self, img, prev_slots=None):
        """savi_diffusion.py:169-216: img [B,T,3,H,W] -> slots [B,T,N,D], masks [B,T,N,*,*]."""
        return _encode_clip(self, img, prev_slots)

    def calc_train_loss(self, data_dict, out_dict):
        """savi_diffusion.py:252-264: the LDM sees the B*T frames as independent images."""
        d = {'img': data_dict['img'].flatten(0, 1), 'slots': out_dict['slots'].flatten(0, 1)}
        return self.dm_decoder.loss_function(d, t=data_dict.get('t'), noise=data_dict.get('noise'))

    @torch.no_grad()
    def log_images(self, data_dict, **kwargs):
        out_dict = self.forward(data_dict)
        B, T = data_dict['img'].shape[:2]
        log = self.dm_decoder.log_images({'img': data_dict['img'].flatten(0, 1),
                                          'slots': out_dict['slots'].flatten(0, 1)}, **kwargs)
        log = {k: v.unflatten(0, (B, T)) for k, v in log.items()}
        log['masks'] = out_dict['masks']
        return log


class SA(SlotModelBase):
    """Plain Slot Attention auto-encoder (registry name 'SA', BASELINE config 0):
    img_based/models/slot_attention.py:126-420 -- ResNet-18 encoder + Slot Attention + the
    spatial-broadcast transposed-conv decoder, trained with an image reconstruction loss."""

    def __init__(self, resolution, slot_dict, enc_dict, dec_dict, loss_dict=None, eps=1e-6,
                 compute_dtype=None, seed=0):
        sp = self._make_spec(resolution, slot_dict, enc_dict, dec_dict)
        super().__init__(sp, seed=seed)
        self.resolution = tuple(resolution)
        self.eps = eps
        self.slot_dict, self.enc_dict, self.dec_dict = dict(slot_dict), dict(enc_dict), dict(dec_dict)
        self.loss_dict = dict(loss_dict or {'use_img_recon_loss': True})
        assert self.loss_dict.get('use_img_recon_loss', True)
        self.num_slots = slot_dict['num_slots']
        self.slot_size = slot_dict['slot_size']
        self.num_iterations = slot_dict['num_iterations']
        self.rplan, self.visual_resolution = spec.encoder_plan(self.resolution, enc_dict)
        self.dec_resolution = tuple(dec_dict['dec_resolution'])
        self.dplan = spec.sa_decoder_plan(self.resolution, dec_dict)
        self.testing = False
        self.compute_dtype = compute_dtype or default_compute_dtype()
        self.step_seed = None
        self.eval_seed = None
        self._graph_cache = {}

    def _make_spec(self, resolution, slot_dict, enc_dict, dec_dict):
        return spec.sa_model(resolution, slot_dict, enc_dict, dec_dict)

    def encode(self, img, init_slots=None):
        """slot_attention.py:318-334 -> slots [B,N,D] fp32."""
        grad = self.training and torch.is_grad_enabled()
        Kp = self.KG() if grad else self.K()
        with torch.set_grad_enabled(grad):
            tok = engine.encoder_out(Kp, self._to_nhwc(img), self.rplan)
            init = self.init_latents[0] if init_slots is None else init_slots.contiguous().float()
            slots, _ = engine.slot_attention(Kp, tok, init, self.num_iterations, self.eps)
        return slots

    def decode(self, slots):
        """slot_attention.py:343-364 -> recon [B,3,H,W], recons [B,N,3,H,W], masks [B,N,1,H,W],
        slots."""
        B, N, _ = slots.shape
        H, W = self.resolution
        grad = self.training and torch.is_grad_enabled() and slots.requires_grad
        Kp = self.KG() if grad else self.K()
        with torch.set_grad_enabled(grad):
            recon, masks, o = engine.sa_decode(Kp, slots.float(), self.dplan, self.dec_resolution,
                                               self.compute_dtype)
            recon_img = kern.NhwcToNchwFn.apply(recon, 3) if grad else ops.nhwc_to_nchw(recon, 3)
        with torch.no_grad():
            recons = ops.nhwc_to_nchw(o.detach(), 3).view(B, N, 3, H, W)
        self._last_recon_nhwc = recon
        return recon_img, recons, masks.detach().view(B, N, 1, H, W), slots

    def forward(self, data_dict):
        self._begin_train_forward()
        slots = self.encode(data_dict['img'])
        if self.testing:
            return {'slots': slots}
        recon_img, recons, masks, _ = self.decode(slots)
        return {'recon_img': recon_img, 'recons': recons, 'masks': masks, 'slots': slots}

    def calc_train_loss(self, data_dict, out_dict):
        """slot_attention.py:366-375: {'img_recon_loss': mse(recon_img, img)}."""
        recon, img = out_dict['recon_img'], data_dict['img']
        if recon.requires_grad:
            # NHWC pair (4th channel zero in both): mean over the 3 real channels -> scale 4/3
            tgt = ops.nchw_to_nhwc(img.float(), torch.float32, 4)
            pred, self._last_recon_nhwc = self._last_recon_nhwc, None    # do not pin the graph
            return {'img_recon_loss': kern.MseFn.apply(pred, tgt, 4.0 / 3.0)}
        tgt = ops.nchw_to_nhwc(img.float(), torch.float32, 4)
        val = ops.mse(ops.nchw_to_nhwc(recon, torch.float32, 4), tgt) * (4.0 / 3.0)
        return {'img_recon_loss': val.reshape(())}

    @torch.no_grad()
    def calc_eval_loss(self, data_dict, out_dict):
        """slot_attention.py:377-420 / savi.py:510-557: the reconstruction loss plus, when GT masks
        are given, ARI / FG-ARI / mIoU / FG-mIoU / mBO of the argmax of the decoder's alpha masks
        (a clip's T frames are folded into the spatial dims: temporal consistency counts)."""
        loss_dict = self.calc_train_loss(data_dict, {k: (v.detach() if torch.is_tensor(v) else v)
                                                     for k, v in out_dict.items()})
        if 'masks' in data_dict:
            from . import metrics
            pm = out_dict['masks']
            if pm.dim() in (5, 6) and pm.shape[-3] == 1:
                pm = pm.squeeze(-3)
            pred = pm.argmax(dim=-3)
            gt = data_dict['masks'].to(pred.device)
            if pred.dim() == 4:
                pred, gt = pred.flatten(1, 2), gt.flatten(1, 2)
            for k, fn in (('ari', metrics.ARI_metric), ('fari', metrics.fARI_metric),
                          ('miou', metrics.miou_metric), ('fmiou', metrics.fmiou_metric),
                          ('mbo', metrics.mbo_metric)):
                loss_dict[k] = torch.tensor(fn(gt, pred), dtype=torch.float32, device=pred.device)
        return loss_dict


class SAVi(SA):
    """Video Slot Attention baseline (registry name 'SAVi', video_based/models/savi.py:117-566):
    the per-frame recurrence of the video models with the plain-SA transposed-conv decoder applied
    to every frame, trained with the image reconstruction loss."""

    def __init__(self, resolution, clip_len, slot_dict, enc_dict, dec_dict, pred_dict,
                 loss_dict=None, eps=1e-6, compute_dtype=None, seed=0):
        assert pred_dict.get('pred_type', 'transformer') == 'transformer' and \
            not pred_dict.get('pred_rnn', False), 'hot path covers the transformer predictor'
        self._pred_dict = dict(pred_dict)
        super().__init__(resolution, slot_dict, enc_dict, dec_dict, loss_dict, eps, compute_dtype,
                         seed)
        self.clip_len = clip_len
        self.pred_dict = dict(pred_dict)
        self.pred_dropout = 0.1            # nn.TransformerEncoderLayer default (predictor.py:33-38)

    def _make_spec(self, resolution, slot_dict, enc_dict, dec_dict):
        return spec.savi_model(resolution, slot_dict, enc_dict, dec_dict, self._pred_dict)

    def encode(self, img, prev_slots=None):
        """savi.py:366-397: img [B,T,3,H,W] -> slots [B,T,N,D]."""
        return _encode_clip(self, img, prev_slots)[0]

    def forward(self, data_dict):
        """savi.py:445-475 (clips longer than clip_len are not split: 288 GB of HBM)."""
        img = data_dict['img']
        B, T = img.shape[:2]
        self._begin_train_forward()
        slots = self.encode(img)
        if self.testing:
            return {'slots': slots}
        recon_img, recons, masks, _ = self.decode(slots.flatten(0, 1))
        out = {'recon_img': recon_img, 'recons': recons, 'masks': masks}
        out = {k: v.unflatten(0, (B, T)) for k, v in out.items()}
        out['slots'] = slots
        return out

    def calc_train_loss(self, data_dict, out_dict):
        """savi.py:500-508: mse over all frames."""
        return super().calc_train_loss({'img': data_dict['img'].flatten(0, 1)},
                                       {'recon_img': out_dict['recon_img'].flatten(0, 1)})


class VQVAE(SlotModelBase):
    """Stand-alone VQ-VAE (registry name 'VQVAE', video_based/models/vqvae/VQVAE.py:40-172):
    the LDM's first stage -- encode, quantize, decode, calc_eval_loss, and stage-1 training
    (SURVEY 8(f) row 1): backward through the decoder, the straight-through quantizer with its
    commitment loss, and the encoder, incl. the single-head AttnBlocks.  The LPIPS term needs the
    `lpips` VGG network (absent): percept_loss is 0 and contributes no gradient."""

    def __init__(self, enc_dec_dict, vq_dict, use_loss=True, compute_dtype=None, seed=0):
        super().__init__(spec.vqvae_model(enc_dec_dict, vq_dict), seed=seed)
        self.ed, self.vq_dict = dict(enc_dec_dict), dict(vq_dict)
        self.resolution = enc_dec_dict['resolution']
        self.embed_dim, self.n_embed = vq_dict['embed_dim'], vq_dict['n_embed']
        self.percept_loss_w = float(vq_dict.get('percept_loss_w', 0.))
        self.beta = 0.25
        self.vq_key = 'quantize.embedding.weight'
        self.compute_dtype = compute_dtype or default_compute_dtype()
        self._graph_cache = {}

    @property
    def device(self):
        return self.quant_conv.weight.device

    @property
    def dtype(self):
        return self.quant_conv.weight.dtype

    def _flat(self, x):
        return (x.flatten(0, 1), x.shape[:2]) if x.dim() == 5 else (x, None)      # temporal_wrapper

    @torch.no_grad()
    def encode(self, x):
        """VQVAE.py:94-99: pre-VQ features [B,embed_dim,h,w] (the x0 of the LDM)."""
        xf, bt = self._flat(x)
        z = ops.nhwc_to_nchw(engine.vae_encode(self.K(), self._to_nhwc(xf), self.ed, prefix=''), 3)
        return z.unflatten(0, tuple(bt)) if bt is not None else z

    def _quantize_nhwc(self, z):
        idx, zq = ops.vq_nearest(z, self.bank().f(self.vq_key))
        return zq, idx

    @torch.no_grad()
    def encode_quantize(self, x):
        """VQVAE.py:86-92 -> quant [B,3,h,w], quant_loss (scalar), token ids [B,h,w]."""
        xf, bt = self._flat(x)
        z = engine.vae_encode(self.K(), self._to_nhwc(xf), self.ed, prefix='')
        zq, idx = self._quantize_nhwc(z)
        # legacy loss mean((zq.detach()-z)^2) + beta*mean((zq-z.detach())^2) = (1+beta)*mse(zq, z)
        # over the embed_dim real channels of the 4-channel NHWC pair
        ql = (ops.mse(zq, z) * (4.0 / 3.0) * (1.0 + self.beta)).reshape(())
        q = ops.nhwc_to_nchw(zq, 3)
        if bt is not None:
            q, idx = q.unflatten(0, tuple(bt)), idx.unflatten(0, tuple(bt))
        return q, ql, idx

    @torch.no_grad()
    def decode(self, quant):
        """VQVAE.py:110-114: already quantized features -> image."""
        qf, bt = self._flat(quant)
        img = engine.vae_decode(self.K(), ops.nchw_to_nhwc(qf.float(), torch.float32, 4), self.ed,
                                prefix='', quantize=False)
        out = ops.nhwc_to_nchw(img, 3)
        return out.unflatten(0, tuple(bt)) if bt is not None else out

    @torch.no_grad()
    def quantize_decode(self, h):
        """VQVAE.py:102-107."""
        hf, bt = self._flat(h)
        img = engine.vae_decode(self.K(), ops.nchw_to_nhwc(hf.float(), torch.float32, 4), self.ed,
                                prefix='', quantize=True)
        out = ops.nhwc_to_nchw(img, 3)
        return out.unflatten(0, tuple(bt)) if bt is not None else out

    def forward(self, data_dict):
        """VQVAE.py:116-126.  In training mode with autograd on, the HIP backward is recorded
        (stage-1 training): encoder -> straight-through quantizer -> decoder."""
        img = data_dict['img']
        if not (torch.is_grad_enabled() and self.training):
            quant, quant_loss, token_id = self.encode_quantize(img)
            return {'recon': self.decode(quant), 'token_id': token_id, 'quant_loss': quant_loss}
        xf, bt = self._flat(img)
        self._begin_train_forward()
        Kp = self.KG()
        z = engine.vae_encode(Kp, self._to_nhwc(xf), self.ed, prefix='')
        off, cnt = self._offsets[self.vq_key]
        dcode = self.grad_arena()[off:off + cnt].view(self.n_embed, self.embed_dim)
        zq, quant_loss, idx = kern.VqFn.apply(z, self.bank().anchor, self.bank().f(self.vq_key), dcode,
                                              self.beta)
        recon = engine.vae_decode(Kp, zq, self.ed, prefix='', quantize=False)
        self._last_recon_nhwc = recon
        rec = kern.NhwcToNchwFn.apply(recon, 3)
        if bt is not None:
            rec, idx = rec.unflatten(0, tuple(bt)), idx.unflatten(0, tuple(bt))
        return {'recon': rec, 'token_id': idx, 'quant_loss': quant_loss}

    def calc_train_loss(self, data_dict, out_dict):
        """VQVAE.calc_train_loss + VQLPIPSLoss.forward (VQVAE.py:128-137, loss.py:19-46): L1 when
        the perceptual term is configured, else MSE; percept_loss = 0 (no LPIPS network here)."""
        l1 = self.percept_loss_w > 0
        if out_dict['recon'].requires_grad:
            img = self._flat(data_dict['img'])[0]
            tgt = ops.nchw_to_nhwc(img.float(), torch.float32, 4)
            pred, self._last_recon_nhwc = self._last_recon_nhwc, None
            rl = kern.MseFn.apply(pred, tgt, 4.0 / 3.0, l1)
            return {'quant_loss': out_dict['quant_loss'], 'recon_loss': rl,
                    'percept_loss': torch.zeros((), dtype=torch.float32, device=pred.device)}
        return self._eval_losses(data_dict, out_dict)

    @torch.no_grad()
    def _eval_losses(self, data_dict, out_dict):
        from . import metrics
        img, recon = data_dict['img'].float(), out_dict['recon'].float()
        x = img.reshape(1, -1).to(recon.device)
        y = recon.reshape(1, -1)
        se, n = metrics._sqerr(x, y, mode=1 if self.percept_loss_w > 0 else 0)
        dev = recon.device
        return {'quant_loss': out_dict['quant_loss'],
                'recon_loss': torch.tensor(float(se[0]) / n, dtype=torch.float32, device=dev),
                'percept_loss': torch.zeros((), dtype=torch.float32, device=dev)}

    @torch.no_grad()
    def calc_eval_loss(self, data_dict, out_dict):
        """VQVAE.py:139-146."""
        from . import metrics
        loss_dict = self.calc_train_loss(data_dict, out_dict)
        img, recon = data_dict['img'].float(), out_dict['recon'].float()
        se, n = metrics._sqerr(img.reshape(1, -1).to(recon.device), recon.reshape(1, -1), mode=0)
        loss_dict['recon_mse'] = torch.tensor(float(se[0]) / n, dtype=torch.float32, device=recon.device)
        return loss_dict


def build_model(params):
    """Registry (img_based/models/__init__.py:12-39, video_based/models/__init__.py:12-33) for the
    hot-path models."""
    if params.model == 'SA':
        return SA(resolution=params.resolution, slot_dict=params.slot_dict, enc_dict=params.enc_dict,
                  dec_dict=params.dec_dict, loss_dict=params.loss_dict)
    if params.model == 'VQVAE':
        return VQVAE(enc_dec_dict=params.enc_dec_dict, vq_dict=params.vq_dict)
    if params.model == 'SAVi':
        return SAVi(resolution=params.resolution, clip_len=params.input_frames,
                    slot_dict=params.slot_dict, enc_dict=params.enc_dict, dec_dict=params.dec_dict,
                    pred_dict=params.pred_dict, loss_dict=params.loss_dict)
    if params.model == 'SAViDiffusion':
        return SAViDiffusion(resolution=params.resolution, clip_len=params.input_frames,
                             slot_dict=params.slot_dict, enc_dict=params.enc_dict,
                             dec_dict=params.dec_dict, pred_dict=params.pred_dict,
                             loss_dict=params.loss_dict)
    if params.model == 'SADiffusion':
        return SADiffusion(resolution=params.resolution, slot_dict=params.slot_dict,
                           enc_dict=params.enc_dict, dec_dict=params.dec_dict,
                           loss_dict=params.loss_dict)
    raise NotImplementedError(f'{params.model} is not on the MI355X hot path yet')
