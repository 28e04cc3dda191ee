"""Generate tests/golden/*.npz from the REAL reference (build container only).

    python tools/gen_golden.py            # writes tests/golden/sadiff_b2.npz ...

The reference (img_based SADiffusion, CLEVRTex config, num_slots overridden to 7 as
BASELINE.json asks) is imported through tools/ref_harness.py, its weights are overwritten by
tests/detfill.py's deterministic recipe, and inputs/outputs of every hot-path row of
SURVEY.md section 8(a) are captured in fp32 on CPU.  Only data is stored (no reference code).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..'))
import ref_harness as rh                      # noqa: E402
from tests.detfill import det_fill_, is_buffer_name, make_inputs   # noqa: E402

OUT = os.path.join(HERE, '..', 'tests', 'golden')


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    im = rh.ref_models('img_based')
    P = rh.ref_params('img_based', 'sa_ldm', 'sa_ldm_clevrtex_params-res128')
    P.slot_dict['num_slots'] = 7
    model = im.build_model(P)
    det_fill_(model.state_dict().items(), skip=is_buffer_name)
    B = 2
    img, t, noise, x_T = make_inputs(B)
    G = dict(t=t, noise=noise, x_T=x_T, img_checksum=torch.stack([img.double().sum(), (img.double() ** 2).sum(), img[1, 2, 77, 5].double()]))

    # ---- a1-a4: encoder + slot attention (train-res masks and eval-res masks)
    model.train()
    with torch.no_grad():
        G['enc_out_sub8'] = model._get_encoder_out(img)[:, 3::8].contiguous()
        slots, masks = model.encode(img)
    G['slots'], G['masks_train'] = slots, masks
    G['masks_train_argmax'] = masks.argmax(1)
    model.eval()
    with torch.no_grad():
        slots_e, masks_e = model.encode(img)
    assert torch.equal(slots_e, slots)
    G['masks_eval_sub4'] = masks_e[:, :, 1::4, 2::4].contiguous()
    G['masks_eval_argmax'] = masks_e.argmax(1)

    dm = model.dm_decoder
    # ---- a6, a14, a15: VQ-VAE
    with torch.no_grad():
        x0 = dm.vae.encode(img)
        G['x0'] = x0
        zq, _, (_, _, idx) = dm.vae.vqvae.quantize(x0)
        G['x0_vq_idx'] = idx
        G['x0_vq'] = dm.vae.quantize(x0)
        G['x0_decoded_sub2'] = dm.vae.decode(x0)[:, :, 1::2, ::2].contiguous()
    # ---- a7-a11: q-sample, UNet eps prediction, loss
    with torch.no_grad():
        xt = dm._sample_xt_from_x0(x0=x0, t=t, noise=noise)
        G['x_t'] = xt
        G['eps_pred'] = dm.forward(xt, t, context=slots)
        G['denoise_loss'] = torch.nn.functional.mse_loss(G['eps_pred'], noise)
        tf = torch.tensor([123.456, 987.125])
        G['t_frac'] = tf
        G['eps_pred_frac'] = dm.forward(xt, tf, context=slots)
    # ---- gradients of the train step (dropout off: eval-mode UNet, train-mode SA has no RNG)
    model.train()
    dm.model.eval()   # disables the ResBlock dropout only (no BN anywhere)
    for p in model.parameters():
        p.grad = None
    out = model(dict(img=img))
    torch.manual_seed(99)
    # reproduce loss_function with our explicit t / noise
    x_noisy = dm._sample_xt_from_x0(x0=x0, t=t, noise=noise)
    pred = dm.forward(x_noisy, t, context=out['slots'])
    loss = torch.nn.functional.mse_loss(pred, noise)
    loss.backward()
    G['train_loss'] = loss.detach()
    named = dict(model.named_parameters())
    gn2 = 0.
    for n, p in named.items():
        if p.grad is not None:
            gn2 += float((p.grad.double() ** 2).sum())
    G['grad_global_norm'] = torch.tensor(gn2 ** 0.5)
    for n in ['init_latents', 'slot_attention.gru.bias_ih',               'encoder.conv1.weight', 'encoder.layer3.1.bn2.weight', 'dm_decoder.model.diffusion_model.out.2.weight',
              'dm_decoder.model.diffusion_model.input_blocks.0.0.weight',
              'dm_decoder.model.diffusion_model.middle_block.1.transformer_blocks.0.attn2.to_out.0.bias',
              'dm_decoder.model.diffusion_model.output_blocks.5.2.conv.bias',
              'dm_decoder.model.diffusion_model.input_blocks.4.0.in_layers.0.weight']:
        G['grad:' + n] = named[n].grad.detach().clone()
    G['grad_norms_names'] = np.array(sorted(n for n, p in named.items() if p.grad is not None))
    G['grad_norms'] = torch.tensor([float(named[n].grad.norm()) for n in G['grad_norms_names']])

    # ---- a12/a13: schedule tables + DPM-Solver++ 20-NFE trajectory
    model.eval()
    from slotdiffusion.img_based.models.ddpm import dpm_solver as ds
    ns = ds.NoiseScheduleVP(betas=dm.betas)
    tt = torch.tensor([1.0, 0.95005, 0.5, 0.123, 0.001])
    G['ns_t'] = tt
    G['ns_log_alpha'] = ns.marginal_log_mean_coeff(tt)
    G['ns_lambda'] = ns.marginal_lambda(tt)
    G['ns_inv_lambda'] = ns.inverse_lambda(G['ns_lambda'])
    solver = ds.DPM_Solver(lambda x, t: x, ns, algorithm_type='dpmsolver++')
    outer, orders = solver.get_orders_and_timesteps_for_singlestep_solver(
        20, 3, 'time_uniform', 1.0, 1e-3, 'cpu')
    G['dpm_outer'], G['dpm_orders'] = outer, torch.tensor(orders)
    with torch.no_grad():
        dm.model.vae = dm.vae
        model_fn = ds.model_wrapper(model=dm.model, noise_schedule=ns, model_type='noise',
                                    guidance_type='classifier-free', condition=slots)
        sampler = ds.DPM_Solver(model_fn, ns, algorithm_type='dpmsolver++',
                                correcting_x0_fn=False, vq_denoised=True)
        x, inter = sampler.sample(x_T.clone(), steps=20, order=3, method='singlestep',
                                  return_intermediate=True)
        dm.model.vae = None
        G['dpm_trace'] = torch.stack(inter, 0)
        G['dpm_final'] = x
        G['samples'] = dm.vae.decode(x)
        # first NFE in isolation (robust parity point): x0-pred at t=T before/after VQ
        t1 = torch.ones(B)
        eps1 = dm.forward(x_T, (t1 - 1e-3) * 1000., context=slots)
        G['nfe0_eps'] = eps1

    os.makedirs(OUT, exist_ok=True)
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in G.items()}
    for k in list(arrs):
        if arrs[k].dtype == np.int64 and k != 't':
            arrs[k] = arrs[k].astype(np.int16)
    np.savez_compressed(os.path.join(OUT, 'sadiff_b2.npz'), **arrs)
    for k, v in arrs.items():
        print(k, v.shape, v.dtype)



def main_ddim():
    """tests/golden/ddim_b2.npz: the reference's DDIM sampler (ddim.py:90-218, eta = 0, VQ-denoised)
    driven exactly as CondDDPM.generate_imgs(use_ddim=True) does (cond_ddpm.py:180-190) but with 50
    steps, B = 2, from the slots / x_T of sadiff_b2.npz."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    im = rh.ref_models('img_based')
    P = rh.ref_params('img_based', 'sa_ldm', 'sa_ldm_clevrtex_params-res128')
    P.slot_dict['num_slots'] = 7
    model = im.build_model(P)
    det_fill_(model.state_dict().items(), skip=is_buffer_name)
    model.eval()
    img, t, noise, x_T = make_inputs(2)
    with torch.no_grad():
        slots, _ = model.encode(img)
    dm = model.dm_decoder
    from slotdiffusion.video_based.models.ddpm import ddim as dd
    # the reference class moves every buffer to 'cuda' (ddim.py:31-35); keep them on the CPU here
    dd.DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    sampler = dd.DDIMSampler(dm, schedule=dm.beta_schedule)
    steps = 50
    import slotdiffusion.video_based.models.ddpm.ddim as ddmod
    ddmod.noise_like = lambda shape, device, repeat=False: x_T.clone()      # fixed x_T, eta = 0
    with torch.no_grad():
        x, inter = sampler.generate_imgs(steps, tuple(x_T.shape), conditioning=slots, eta=0.,
                                         verbose=False, log_every_t=10, ret_intermed=True)
    G = dict(ddim_timesteps=torch.tensor(sampler.ddim_timesteps), ddim_alphas=torch.as_tensor(sampler.ddim_alphas),
             ddim_alphas_prev=torch.as_tensor(sampler.ddim_alphas_prev), ddim_final=x, ddim_inter=inter,
             steps=torch.tensor(steps))
    with torch.no_grad():
        _, _, (_, _, idx) = dm.vae.vqvae.quantize(x)
        G['ddim_final_idx'] = idx
    np.savez_compressed(os.path.join(OUT, 'ddim_b2.npz'), **{k: np.asarray(v.numpy() if torch.is_tensor(v) else v) for k, v in G.items()})
    print('wrote ddim_b2.npz', {k: tuple(np.asarray(v).shape) for k, v in G.items()})


def main_anc():
    """tests/golden/anc_x0_b2.npz: the reference's ancestral sampler
    (cond_ddpm.py:55-132: _p_mean_variance / _p_sample / _sample_x0_from_noise, VQ-denoised LDM
    branch) as single steps with explicit x, t and noise, the posterior tables it reads, and the
    x0-prediction variant (pred_target='x0': loss target, _p_sample, DPM-Solver 'x_start')."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    im = rh.ref_models('img_based')
    import slotdiffusion.img_based.models.ddpm.cond_ddpm as cd
    G = {}
    for target in ('eps', 'x0'):
        P = rh.ref_params('img_based', 'sa_ldm', 'sa_ldm_clevrtex_params-res128')
        P.slot_dict['num_slots'] = 7
        P.dec_dict['diffusion_dict']['pred_target'] = target
        model = im.build_model(P)
        det_fill_(model.state_dict().items(), skip=is_buffer_name)
        model.eval()
        img, t, noise, x_T = make_inputs(2)
        with torch.no_grad():
            slots, _ = model.encode(img)
        dm = model.dm_decoder
        assert dm.pred_target == target and dm.vq_denoised and not dm.clip_denoised
        pre = target + '_'
        if target == 'eps':
            for k in ('posterior_mean_coef1', 'posterior_mean_coef2', 'posterior_log_variance_clipped',
                      'sqrt_recip_alphas_bar', 'sqrt_recipm1_alphas_bar'):
                G['tab_' + k] = getattr(dm, k)[[0, 1, 2, 250, 500, 998, 999]].clone()
        # single ancestral steps with explicit noise (x_{t-1} | x_t = x_T-like state)
        nz = torch.randn(x_T.shape, generator=torch.Generator().manual_seed(77))
        G['anc_noise'] = nz
        cd.noise_like = lambda shape, device, repeat=False: nz.clone()
        ts = [999, 500, 1, 0]
        G['anc_t'] = torch.tensor(ts)
        outs, means = [], []
        x = x_T.clone()
        with torch.no_grad():
            for tv in ts:
                tt = torch.full((2,), tv, dtype=torch.long)
                mean, _, logvar = dm._p_mean_variance(x.clone(), tt, slots, clip_denoised=False,
                                                      vq_denoised=True)
                means.append(mean)
                x = dm._p_sample(x, tt, slots, clip_denoised=False, vq_denoised=True)
                outs.append(x)
        G[pre + 'anc_mean'] = torch.stack(means, 0)
        G[pre + 'anc_x'] = torch.stack(outs, 0)
        if target == 'x0':
            # loss with the x0 target (ldm.py:59-83) at explicit t / noise, a few gradient norms
            model.train()
            dm.model.eval()
            with torch.no_grad():
                x0 = dm.vae.encode(img)
            out = model(dict(img=img))
            xt = dm._sample_xt_from_x0(x0=x0, t=t, noise=noise)
            pred = dm.forward(xt, t, context=out['slots'])
            loss = torch.nn.functional.mse_loss(pred, x0)
            loss.backward()
            G['x0_train_loss'] = loss.detach()
            named = dict(model.named_parameters())
            names = sorted(n for n, p in named.items() if p.grad is not None)[::16]
            G['x0_grad_norms_names'] = np.array(names)
            G['x0_grad_norms'] = torch.tensor([float(named[n].grad.norm()) for n in names])
            G['x0_pred'] = pred.detach()
            # DPM-Solver++ with model_type 'x_start' (cond_ddpm.py:160-166)
            model.eval()
            from slotdiffusion.img_based.models.ddpm import dpm_solver as ds
            ns = ds.NoiseScheduleVP(betas=dm.betas)
            with torch.no_grad():
                dm.model.vae = dm.vae
                model_fn = ds.model_wrapper(model=dm.model, noise_schedule=ns, model_type='x_start',
                                            guidance_type='classifier-free', condition=slots)
                sampler = ds.DPM_Solver(model_fn, ns, algorithm_type='dpmsolver++',
                                        correcting_x0_fn=False, vq_denoised=True)
                xx, inter = sampler.sample(x_T.clone(), steps=20, order=3, method='singlestep',
                                           return_intermediate=True)
                dm.model.vae = None
            G['x0_dpm_trace'] = torch.stack(inter, 0)
            G['x0_dpm_final'] = xx
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in G.items()}
    np.savez_compressed(os.path.join(OUT, 'anc_x0_b2.npz'), **arrs)
    print('wrote anc_x0_b2.npz', {k: tuple(np.asarray(v).shape) for k, v in arrs.items()})


def main_vpred():
    """tests/golden/vpred_b1t2.npz (pred_target='v'): the video reference model
    (only video_based accepts 'v': ddpm.py:79) with the MOVi-E config, a B=1 clip of T=2 frames --
    the v loss target (ldm.py:74-78) with a few gradient norms, single ancestral steps through
    _p_mean_variance's v branch (cond_ddpm.py:63-67) and DPM-Solver++ with model_type 'v'
    (cond_ddpm.py:163-164, dpm_solver.py:362-365)."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    vm = rh.ref_models('video_based')
    import slotdiffusion.video_based.models.ddpm.cond_ddpm as cd
    P = rh.ref_params('video_based', 'savi_ldm', 'savi_ldm_movie_params-res128')
    P.dec_dict['diffusion_dict']['pred_target'] = 'v'
    P.n_sample_frames = P.input_frames = 2
    model = vm.build_model(P)
    det_fill_(model.state_dict().items(), skip=is_buffer_name)
    T = 2
    img, t, noise, x_T = make_inputs(T, seed=17)
    clip = img.view(1, T, 3, 128, 128)
    dm = model.dm_decoder
    assert dm.pred_target == 'v' and dm.vq_denoised and not dm.clip_denoised
    G = dict(t=t, noise=noise, x_T=x_T)
    model.eval()
    with torch.no_grad():
        slots = model(dict(img=clip))['slots'].flatten(0, 1)
    G['slots'] = slots
    nz = torch.randn(x_T.shape, generator=torch.Generator().manual_seed(78))
    G['anc_noise'] = nz
    cd.noise_like = lambda shape, device, repeat=False: nz.clone()
    ts = [999, 500, 1, 0]
    G['anc_t'] = torch.tensor(ts)
    outs, means = [], []
    x = x_T.clone()
    with torch.no_grad():
        for tv in ts:
            tt = torch.full((T,), tv, dtype=torch.long)
            mean, _, _ = dm._p_mean_variance(x.clone(), tt, slots, clip_denoised=False, vq_denoised=True)
            means.append(mean)
            x = dm._p_sample(x, tt, slots, clip_denoised=False, vq_denoised=True)
            outs.append(x)
    G['v_anc_mean'], G['v_anc_x'] = torch.stack(means, 0), torch.stack(outs, 0)
    # loss with the v target at explicit t / noise
    model.train()
    dm.model.eval()
    model.predictor.eval()
    for p in model.parameters():
        p.grad = None
    out = model(dict(img=clip))
    with torch.no_grad():
        x0 = dm.vae.encode(img)
    xt = dm._sample_xt_from_x0(x0=x0, t=t, noise=noise)
    pred = dm.forward(xt, t, context=out['slots'].flatten(0, 1))
    a = dm.sqrt_alphas_bar[t].view(-1, 1, 1, 1)
    sg = dm.sqrt_one_minus_alphas_bar[t].view(-1, 1, 1, 1)
    loss = torch.nn.functional.mse_loss(pred, (a * noise - sg * x0).detach())
    # (the library path must give the same number: LDM.loss_function draws its own t / noise, so the
    # formula is restated here and cross-checked once with patched RNG below)
    loss.backward()
    G['v_train_loss'], G['v_pred'] = loss.detach(), pred.detach()
    named = dict(model.named_parameters())
    names = sorted(n for n, p in named.items() if p.grad is not None)[::16]
    G['v_grad_norms_names'] = np.array(names)
    G['v_grad_norms'] = torch.tensor([float(named[n].grad.norm()) for n in names])
    # cross-check against the reference's own loss_function with its RNG pinned to (t, noise)
    ri, rl = torch.randint, torch.randn_like
    try:
        torch.randint = lambda *a_, **k_: t.clone()
        torch.randn_like = lambda *_a, **_k: noise.clone()
        with torch.no_grad():
            ref_loss = dm.loss_function(dict(img=img, slots=out['slots'].flatten(0, 1).detach()))['denoise_loss']
    finally:
        torch.randint, torch.randn_like = ri, rl
    assert abs(float(ref_loss) - float(loss)) <= 1e-6 * max(1.0, float(loss)), (float(ref_loss), float(loss))
    # DPM-Solver++ with model_type 'v'
    model.eval()
    from slotdiffusion.video_based.models.ddpm import dpm_solver as ds
    nsch = ds.NoiseScheduleVP(betas=dm.betas)
    with torch.no_grad():
        dm.model.vae = dm.vae
        model_fn = ds.model_wrapper(model=dm.model, noise_schedule=nsch, model_type='v',
                                    guidance_type='classifier-free', condition=slots)
        sampler = ds.DPM_Solver(model_fn, nsch, algorithm_type='dpmsolver++', correcting_x0_fn=False,
                                vq_denoised=True)
        xx, inter = sampler.sample(x_T.clone(), steps=20, order=3, method='singlestep',
                                   return_intermediate=True)
        dm.model.vae = None
    G['v_dpm_trace'], G['v_dpm_final'] = torch.stack(inter, 0), xx
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in G.items()}
    np.savez_compressed(os.path.join(OUT, 'vpred_b1t2.npz'), **arrs)
    print('wrote vpred_b1t2.npz', {k: tuple(np.asarray(v).shape) for k, v in arrs.items()})


def dino_name_map(key):
    """transformers 4.27 ViTModel key (the reference era, used by this repository's checkpoints) ->
    the key of the installed transformers 5.x ViTModel."""
    k = key.replace('.encoder.layer.', '.layers.')
    for a, b in (('.attention.attention.query.', '.attention.q_proj.'), ('.attention.attention.key.', '.attention.k_proj.'),
                 ('.attention.attention.value.', '.attention.v_proj.'), ('.attention.output.dense.', '.attention.o_proj.'),
                 ('.intermediate.dense.', '.mlp.fc1.'), ('.output.dense.', '.mlp.fc2.')):
        k = k.replace(a, b)
    return k


def main_dino():
    """tests/golden/sadiff_dino_b1.npz (SURVEY 8(f) row 4 + BASELINE config 5): the reference
    img_based SADiffusion with the DINO ViT-S/8 encoder config (sa_ldm_dino_coco_params-res224:
    224 x 224 images, 28 x 28 feature tokens, 7 slots of 256, latent 56 x 56), B = 1.
    `ViTModel.from_pretrained` needs the HF hub: it is replaced by `ViTModel(ViTConfig(...))` of the
    installed transformers with the dino-vits8 architecture (hidden 384, 12 layers, 6 heads, MLP 1536,
    patch 8, qkv bias) -- random architecture-true weights, overwritten by the deterministic fill."""
    from transformers import ViTConfig, ViTModel            # the REAL classes, before the harness mocks the module
    torch.manual_seed(0)
    torch.set_num_threads(8)
    im = rh.ref_models('img_based')
    import slotdiffusion.img_based.models.dino as rdino
    import slotdiffusion.video_based.models.dino as vdino      # (the class img_based actually builds)

    class _ViT:
        @staticmethod
        def from_pretrained(name):
            assert name == 'facebook/dino-vits8', name
            return ViTModel(ViTConfig(hidden_size=384, num_hidden_layers=12, num_attention_heads=6,
                                      intermediate_size=1536, patch_size=8, image_size=224, qkv_bias=True))
    rdino.ViTModel = vdino.ViTModel = _ViT
    P = rh.ref_params('img_based', 'sa_ldm', 'sa_ldm_dino_coco_params-res224')
    model = im.build_model(P)
    det_fill_(model.state_dict().items(), skip=is_buffer_name)
    # this repository's model (4.27 key names): same fill recipe; its DINO values go into the HF module
    from slotdiffusion_amd import compat
    from slotdiffusion_amd.models import build_model as my_build
    mine = my_build(compat.Params(**{k: getattr(P, k) for k in ('model', 'resolution', 'slot_dict', 'enc_dict',
                                                                 'dec_dict', 'loss_dict')}))
    det_fill_(mine.state_dict().items(), skip=is_buffer_name)
    ref_sd = model.state_dict()
    my_sd = mine.state_dict()
    assert len(ref_sd) == len(my_sd), (len(ref_sd), len(my_sd))
    n_dino = 0
    with torch.no_grad():
        for k, v in my_sd.items():
            rk = dino_name_map(k) if k.startswith('encoder.dino.') else k
            assert rk in ref_sd and ref_sd[rk].shape == v.shape, (k, rk)
            if k.startswith('encoder.dino.'):
                ref_sd[rk].copy_(v)
                n_dino += 1
            elif not is_buffer_name(k):
                assert torch.equal(ref_sd[rk], v), k           # same fill everywhere else
    assert n_dino == 200
    g = torch.Generator().manual_seed(21)
    img = (torch.randn(1, 3, 224, 224, generator=g) * 0.5).clamp(-1, 1)
    t = torch.tensor([437])
    noise = torch.randn(1, 3, 56, 56, generator=g)
    G = dict(t=t, noise=noise, img_checksum=torch.stack([img.double().sum(), (img.double() ** 2).sum()]))
    model.train()
    dm = model.dm_decoder
    dm.model.eval()
    with torch.no_grad():
        feat = model.encoder(img)                           # [1, 384, 28, 28]
    G['dino_feat_sub'] = feat[:, ::8].contiguous()
    for p in model.parameters():
        p.grad = None
    out = model(dict(img=img))
    G['slots'], G['masks_train_argmax'] = out['slots'].detach(), out['masks'].detach().argmax(1)
    with torch.no_grad():
        x0 = dm.vae.encode(img)
    G['x0'] = x0
    xt = dm._sample_xt_from_x0(x0=x0, t=t, noise=noise)
    pred = dm.forward(xt, t, context=out['slots'])
    loss = torch.nn.functional.mse_loss(pred, noise)
    loss.backward()
    G['eps_pred'], G['train_loss'] = pred.detach(), loss.detach()
    named = dict(model.named_parameters())
    names = sorted(n for n, p in named.items() if p.grad is not None)[::8]
    G['grad_norms_names'] = np.array(names)
    G['grad_norms'] = torch.tensor([float(named[n].grad.norm()) for n in names])
    assert all(named[n].grad is None for n in named if n.startswith('encoder.dino.'))
    model.eval()
    with torch.no_grad():
        oe = model(dict(img=img))
    G['masks_eval_argmax'] = oe['masks'].argmax(1)
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in G.items()}
    for k in list(arrs):
        if arrs[k].dtype == np.int64 and k != 't':
            arrs[k] = arrs[k].astype(np.int16)
    np.savez_compressed(os.path.join(OUT, 'sadiff_dino_b1.npz'), **arrs)
    print('wrote sadiff_dino_b1.npz', {k: tuple(np.asarray(v).shape) for k, v in arrs.items()})


def metric_inputs():
    """Synthetic id maps for the metric fixtures: B=4 images 32x32; blobs of ids 0..5 (gt) against
    shifted / merged / split predictions; image 2 holds only background; image 3 has fewer
    predicted segments than objects."""
    g = torch.Generator().manual_seed(4321)
    H = W = 32
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    gt = torch.zeros(4, H, W, dtype=torch.long)
    pred = torch.zeros(4, H, W, dtype=torch.long)
    for b in (0, 1, 3):
        for k in range(1, 6):
            cy, cx = torch.randint(4, 28, (2,), generator=g).tolist()
            r = int(torch.randint(3, 7, (1,), generator=g))
            gt[b][((yy - cy) ** 2 + (xx - cx) ** 2) <= r * r] = k
            sh = int(torch.randint(-2, 3, (1,), generator=g))
            m = ((yy - cy - sh) ** 2 + (xx - cx + sh) ** 2) <= (r + (k % 2)) ** 2
            pred[b][m] = (k + 1) % 7 if b != 3 else 1 + (k % 2)
    pred[2] = torch.randint(0, 3, (H, W), generator=g)
    return gt, pred


def main_metrics():
    """tests/golden/metrics_b4.npz: the reference's segmentation metrics
    (video_based/models/eval_utils.py:119-333) on metric_inputs().  PSNR is not generated here
    (skimage is absent from the image; its definition is restated in the oracle)."""
    rh.ref_models('video_based')
    from slotdiffusion.video_based.models import eval_utils as eu
    gt, pred = metric_inputs()
    G = dict(gt=gt.to(torch.int16), pred=pred.to(torch.int16),
             ari_per_image=eu.adjusted_rand_index(gt, pred, False),
             fari_per_image=eu.adjusted_rand_index(gt, pred, True),
             ari=torch.tensor(eu.ARI_metric(gt, pred)), fari=torch.tensor(eu.fARI_metric(gt, pred)),
             miou=torch.tensor(float(eu.miou_metric(gt, pred))),
             fmiou=torch.tensor(float(eu.fmiou_metric(gt, pred))),
             mbo=torch.tensor(float(eu.mbo_metric(gt, pred))))
    # video-shaped ids [B,T,H,W] for the ARI
    gtv, pv = gt.view(2, 2, 32, 32), pred.view(2, 2, 32, 32)
    G['ari_video'] = eu.adjusted_rand_index(gtv, pv, False)
    G['fari_video'] = eu.adjusted_rand_index(gtv, pv, True)
    np.savez_compressed(os.path.join(OUT, 'metrics_b4.npz'), **{k: v.numpy() for k, v in G.items()})
    print('wrote metrics_b4.npz', {k: (tuple(v.shape), v.flatten()[:4].tolist()) for k, v in G.items() if k not in ('gt', 'pred')})


def main_sa():
    """tests/golden/sa_b2.npz: plain Slot Attention auto-encoder (registry 'SA', BASELINE config 0;
    SURVEY 8(a) row a16): slots, recon, masks, loss and parameter-gradient norms at B=2."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    im = rh.ref_models('img_based')
    P = rh.ref_params('img_based', 'sa', 'sa_clevrtex_params-res128')
    P.slot_dict['num_slots'] = 7
    model = im.build_model(P)
    keys = list(model.state_dict().keys())
    det_fill_(model.state_dict().items(), skip=is_buffer_name)
    img = make_inputs(2)[0]
    model.train()
    out = model(dict(img=img))
    loss = model.calc_train_loss(dict(img=img), out)['img_recon_loss']
    loss.backward()
    G = dict(slots=out['slots'].detach(), recon_img_sub2=out['recon_img'].detach()[:, :, 1::2, ::2].contiguous(),
             recon_checksum=torch.stack([out['recon_img'].detach().double().sum(),
                                         (out['recon_img'].detach().double() ** 2).sum()]),
             masks_sub4=out['masks'].detach()[:, :, 0, ::4, 1::4].contiguous(),
             masks_argmax=out['masks'].detach()[:, :, 0].argmax(1),
             recons_sub4=out['recons'].detach()[:, :, :, 1::4, ::4].contiguous(),
             img_recon_loss=loss.detach())
    names = [n for n, p in model.named_parameters()]
    G['grad_norms'] = torch.stack([p.grad.double().norm() for _, p in model.named_parameters()]).float()
    for n, p in model.named_parameters():
        if n in ('decoder.0.0.weight', 'decoder.3.0.bias', 'decoder.4.weight', 'decoder_pos_embedding.dense.weight',
                 'init_latents', 'slot_attention.project_q.1.weight'):
            G['grad/' + n] = (p.grad.detach()[::4, ::4] if n == 'decoder.0.0.weight' else p.grad.detach()).clone()
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, 'sa_b2.npz'), **{k: v.numpy() for k, v in G.items()},
                        state_dict_keys=np.array(keys), param_names=np.array(names))
    print('wrote sa_b2.npz', {k: tuple(v.shape) for k, v in G.items()})


def main_video(cfg_name='savi_ldm_movie_params-res128', num_slots=None, T=3, out_name='savidiff_b1t3.npz',
               seed=11):
    """video_based SAViDiffusion (MOVi-E config, 15 slots, 2 iterations), B=1 clip of T=3 frames.
    `video11x6`: BASELINE config 2 -- the MOVi-D config with num_slots=11 and 6-frame clips
    (n_sample_frames / input_frames = 6, the values BASELINE.json asks for)."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    vm = rh.ref_models('video_based')
    P = rh.ref_params('video_based', 'savi_ldm', cfg_name)
    if num_slots is not None:
        P.slot_dict['num_slots'] = num_slots
    P.n_sample_frames = P.input_frames = T
    model = vm.build_model(P)
    det_fill_(model.state_dict().items(), skip=is_buffer_name)
    B = 1
    img, t, noise, x_T = make_inputs(B * T, seed=seed)
    img = img.view(B, T, 3, 128, 128)
    G = dict(t=t, noise=noise, img_checksum=torch.stack([img.double().sum(), (img.double() ** 2).sum()]))
    model.train()
    dm = model.dm_decoder
    dm.model.eval()                       # UNet ResBlock dropout off
    model.predictor.eval()                # predictor dropout off (nn.TransformerEncoderLayer p=0.1)
    with torch.no_grad():
        G['pred_of_init'] = model.predictor(model.init_latents.detach())
    for p in model.parameters():
        p.grad = None
    out = model(dict(img=img))
    G['slots'], G['masks_train_argmax'] = out['slots'].detach(), out['masks'].detach().argmax(2)
    G['masks_train_sub'] = out['masks'].detach()[:, :, :, ::2, 1::2].contiguous()
    with torch.no_grad():
        x0 = dm.vae.encode(img.flatten(0, 1))
    x_noisy = dm._sample_xt_from_x0(x0=x0, t=t, noise=noise)
    pred = dm.forward(x_noisy, t, context=out['slots'].flatten(0, 1))
    loss = torch.nn.functional.mse_loss(pred, noise)
    loss.backward()
    G['train_loss'] = loss.detach()
    named = dict(model.named_parameters())
    names = sorted(n for n, p in named.items() if p.grad is not None)
    G['grad_norms_names'] = np.array(names)
    G['grad_norms'] = torch.tensor([float(named[n].grad.norm()) for n in names])
    for n in ['init_latents', 'predictor.transformer_encoder.layers.0.self_attn.in_proj_weight',
              'predictor.transformer_encoder.layers.1.linear2.bias',
              'predictor.transformer_encoder.layers.0.norm1.weight']:
        G['grad:' + n] = named[n].grad.detach().clone()
    model.eval()
    with torch.no_grad():
        oe = model(dict(img=img))
    G['masks_eval_argmax'] = oe['masks'].argmax(2)
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in G.items()}
    for k in list(arrs):
        if arrs[k].dtype == np.int64 and k != 't':
            arrs[k] = arrs[k].astype(np.int16)
    np.savez_compressed(os.path.join(OUT, out_name), **arrs)
    keys = {'params': [[k, list(v.shape)] for k, v in model.named_parameters()],
            'frozen': [k for k, v in model.named_parameters() if not v.requires_grad]}
    for k, v in arrs.items():
        print(k, v.shape, v.dtype)


def main_ema():
    """tests/golden/ema_lit.npz: the reference's LitEma (video_based/models/ddpm/ema.py:7-52) run for
    4 updates on a small module (two trainable tensors + one frozen) whose parameters move between
    updates by a fixed recipe -- with the num_updates warm-up of the decay (use_num_upates=True, decay
    0.9999: effective decays 2/11, 3/12, 4/13, 5/14) and without it (decay 0.95).  Stored: the
    parameter values before every update and the shadows after it."""
    import importlib
    rh.install()
    ema_mod = importlib.import_module('slotdiffusion.video_based.models.ddpm.ema')

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            g = torch.Generator().manual_seed(5)
            self.a = torch.nn.Parameter(torch.randn(7, 33, generator=g))
            self.sub = torch.nn.Linear(5, 3)
            with torch.no_grad():
                self.sub.weight.copy_(torch.randn(3, 5, generator=g))
                self.sub.bias.copy_(torch.randn(3, generator=g))
            self.sub.bias.requires_grad_(False)        # frozen: no shadow (ema.py:48-49)

    G = {}
    for tag, decay, use_n in (('warm', 0.9999, True), ('flat', 0.95, False)):
        m = Tiny()
        ema = ema_mod.LitEma(m, decay=decay, use_num_upates=use_n)
        g = torch.Generator().manual_seed(9)
        for it in range(4):
            with torch.no_grad():
                m.a.add_(0.1 * torch.randn(m.a.shape, generator=g))
                m.sub.weight.mul_(1.0 + 0.05 * (it + 1))
            G[f'{tag}_p_a_{it}'] = m.a.detach().clone()
            G[f'{tag}_p_w_{it}'] = m.sub.weight.detach().clone()
            ema(m)
            sh = dict(ema.named_buffers())
            G[f'{tag}_s_a_{it}'] = sh[ema.m_name2s_name['a']].detach().clone()
            G[f'{tag}_s_w_{it}'] = sh[ema.m_name2s_name['sub.weight']].detach().clone()
        G[f'{tag}_num_updates'] = torch.tensor(int(ema.num_updates))
        assert 'sub.bias' not in ema.m_name2s_name
    m0 = Tiny()
    G['init_a'], G['init_w'] = m0.a.detach().clone(), m0.sub.weight.detach().clone()
    np.savez_compressed(os.path.join(OUT, 'ema_lit.npz'), **{k: v.numpy() for k, v in G.items()})
    for k, v in G.items():
        print(k, tuple(v.shape))


def main_savi():
    """tests/golden/savi_b1t3.npz: video_based SAVi baseline (registry 'SAVi', MOVi-E config: 15
    slots, 2 iterations, transformer predictor, spatial-broadcast transposed-conv decoder), B=1 clip
    of T=3 frames: slots, reconstruction, masks, loss, gradient norms, and calc_eval_loss metrics
    against synthetic GT masks."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    vm = rh.ref_models('video_based')
    P = rh.ref_params('video_based', 'savi', 'savi_movie_params-res128')
    model = vm.build_model(P)
    keys = list(model.state_dict().keys())
    det_fill_(model.state_dict().items(), skip=is_buffer_name)
    B, T = 1, 3
    img = make_inputs(B * T, seed=11)[0].view(B, T, 3, 128, 128)
    model.train()
    model.predictor.eval()                # predictor dropout off
    out = model(dict(img=img))
    loss = model.calc_train_loss(dict(img=img), out)['img_recon_loss']
    loss.backward()
    G = dict(slots=out['slots'].detach(),
             recon_img_sub2=out['recon_img'].detach()[:, :, :, 1::2, ::2].contiguous(),
             recon_checksum=torch.stack([out['recon_img'].detach().double().sum(),
                                         (out['recon_img'].detach().double() ** 2).sum()]),
             masks_sub4=out['masks'].detach()[:, :, :, 0, ::4, 1::4].contiguous(),
             masks_argmax=out['masks'].detach()[:, :, :, 0].argmax(2),
             img_recon_loss=loss.detach())
    named = dict(model.named_parameters())
    names = sorted(n for n, p in named.items() if p.grad is not None)
    G['grad_norms_names'] = np.array(names)
    G['grad_norms'] = torch.tensor([float(named[n].grad.norm()) for n in names])
    # eval loss with GT masks: blobs that follow the predicted segmentation loosely
    gt = (G['masks_argmax'] % 5).clone()
    gt[:, :, :40] = 0
    model.eval()
    import slotdiffusion.video_based.models.savi as sv
    with torch.no_grad():
        oe = model(dict(img=img))
        ev = model.calc_eval_loss(dict(img=img, masks=gt), oe)
    G['gt_masks'] = gt
    for k, v in ev.items():
        G['eval_' + k] = v.detach().float().cpu()
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in G.items()}
    for k in list(arrs):
        if arrs[k].dtype == np.int64:
            arrs[k] = arrs[k].astype(np.int16)
    np.savez_compressed(os.path.join(OUT, 'savi_b1t3.npz'), **arrs, state_dict_keys=np.array(keys))
    print('wrote savi_b1t3.npz', {k: tuple(v.shape) for k, v in arrs.items()})
    print({k: float(v) for k, v in ev.items()})


def main_vqvae():
    """tests/golden/vqvae_b2.npz: the stand-alone VQ-VAE (registry 'VQVAE', CLEVRTex config) in
    eval: recon, token ids, quantizer / reconstruction losses of calc_eval_loss at B=2, with the
    perceptual weight set to 0 (lpips is absent) -- plus the L1 value the configured loss would
    take (loss.py:27, restated)."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    im = rh.ref_models('img_based')
    P = rh.ref_params('img_based', 'sa_ldm', 'vqvae_clevrtex_params-res128')
    P.vq_dict['percept_loss_w'] = 0.
    model = im.build_model(P)
    keys = [k for k in model.state_dict().keys() if not k.startswith('loss.')]
    det_fill_([(k, v) for k, v in model.state_dict().items() if not k.startswith('loss.')], skip=is_buffer_name)
    model.eval()
    img = make_inputs(2)[0]
    with torch.no_grad():
        out = model(dict(img=img))
        ev = model.calc_eval_loss(dict(img=img), out)
    G = dict(recon_sub2=out['recon'][:, :, 1::2, ::2].contiguous(),
             recon_checksum=torch.stack([out['recon'].double().sum(), (out['recon'].double() ** 2).sum()]),
             token_id=out['token_id'].to(torch.int16), quant_loss=out['quant_loss'],
             recon_loss=ev['recon_loss'], recon_mse=ev['recon_mse'], percept_loss=ev['percept_loss'],
             recon_l1=torch.abs(img - out['recon']).mean(),
             pre_vq=model.encode(img))
    # stage-1 training step: recon (MSE) + quantizer loss, parameter gradients
    model.train()
    for p in model.parameters():
        p.grad = None
    out = model(dict(img=img))
    ld = model.calc_train_loss(dict(img=img), out)
    (ld['recon_loss'] + ld['quant_loss']).backward()
    named = {n: p for n, p in model.named_parameters() if not n.startswith('loss.')}
    names = sorted(n for n, p in named.items() if p.grad is not None)
    G['train_recon_loss'], G['train_quant_loss'] = ld['recon_loss'].detach(), ld['quant_loss'].detach()
    G['grad_norms'] = torch.tensor([float(named[n].grad.double().norm()) for n in names])
    for n in ('quantize.embedding.weight', 'encoder.conv_in.weight', 'decoder.mid.attn_1.k.weight',
              'encoder.mid.attn_1.proj_out.bias', 'decoder.conv_out.bias'):
        G['grad/' + n] = named[n].grad.detach().clone()
    # the L1 form (percept weight > 0 in the shipped config) at the same point: gradient norms only
    for p in model.parameters():
        p.grad = None
    out = model(dict(img=img))
    (torch.abs(img - out['recon']).mean() + out['quant_loss']).backward()
    G['grad_norms_l1'] = torch.tensor([float(named[n].grad.double().norm()) for n in names])
    np.savez_compressed(os.path.join(OUT, 'vqvae_b2.npz'), **{k: v.detach().numpy() for k, v in G.items()},
                        grad_norms_names=np.array(names),
                        state_dict_keys=np.array(keys))
    print('wrote vqvae_b2.npz', {k: (tuple(v.shape), float(v.flatten()[0])) for k, v in G.items()})


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'ddim':
        main_ddim()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'sa':
        main_sa()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'vqvae':
        main_vqvae()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'savi':
        main_savi()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'metrics':
        main_metrics()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'anc':
        main_anc()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'dino':
        main_dino()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'vpred':
        main_vpred()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'ema':
        main_ema()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'video15x6':
        # BASELINE config 3's shape: MOVi-E config (15 slots) on 6-frame clips
        main_video('savi_ldm_movie_params-res128', num_slots=15, T=6, out_name='savidiff_b1t6_n15.npz',
                   seed=17)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'video11x6':
        main_video('savi_ldm_movid_params-res128', num_slots=11, T=6, out_name='savidiff_b1t6_n11.npz',
                   seed=13)
    elif len(sys.argv) > 1 and sys.argv[1] == 'video':
        main_video()
    else:
        main()
