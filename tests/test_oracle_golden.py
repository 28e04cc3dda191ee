"""Pins the CPU oracle (oracle/slotdiff_oracle.py) against tensors captured from the real
reference (tests/golden/sadiff_b2.npz, produced by tools/gen_golden.py in the build container).
CPU only."""
import numpy as np
import torch

from oracle import slotdiff_oracle as O
from slotdiffusion_amd import spec
from tests import common as C

torch.set_num_threads(8)
_cache = {}


def ctx():
    if not _cache:
        cfg = C.clevrtex_cfg()
        _cache.update(cfg=cfg, W=C.oracle_weights(cfg), G=C.load_golden(),
                      rplan=spec.resnet18_plan(False),
                      uplan=spec.unet_plan(cfg['dec_dict']['unet_dict']),
                      ed=cfg['dec_dict']['vae_dict']['enc_dec_dict'])
        img, t, noise, x_T = C.make_inputs(2)
        _cache.update(img=img)
    return _cache


def close(a, b, atol, rtol=0.):
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a.double() - b.double()).abs()
    lim = atol + rtol * b.double().abs()
    assert bool((err <= lim).all()), f'max err {err.max().item():.3e} (atol {atol})'


def test_spec_matches_reference_keys():
    keys = C.load_keys()
    cfg = C.clevrtex_cfg()
    sp = spec.sa_diffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'])
    ref = keys['img_based/SADiffusion/clevrtex-7slot']
    assert [(p.name, list(p.shape)) for p in sp] == [(k, s) for k, s, _ in ref['state']]
    frozen = set(ref['frozen'])
    mine = [p.name for p in sp if p.trainable and not p.init.startswith('buf')]
    assert mine == [k for k, _ in ref['params'] if k not in frozen]


def test_inputs_regenerate():
    c = ctx()
    img = c['img']
    chk = torch.stack([img.double().sum(), (img.double() ** 2).sum(), img[1, 2, 77, 5].double()])
    assert torch.equal(chk, c['G']['img_checksum'])


def test_encoder_and_slot_attention():
    c = ctx()
    W, G = c['W'], c['G']
    with torch.no_grad():
        eo = O.encoder_out(W, c['img'], c['rplan'])
        close(eo[:, 3::8], G['enc_out_sub8'], 2e-5)
        slots, masks = O.sa_encode(W, c['img'], c['rplan'], 3, training=True)
        close(slots, G['slots'], 2e-5)
        close(masks, G['masks_train'], 1e-5)
        assert torch.equal(masks.argmax(1), G['masks_train_argmax'].long())
        _, masks_e = O.sa_encode(W, c['img'], c['rplan'], 3, training=False)
        close(masks_e[:, :, 1::4, 2::4], G['masks_eval_sub4'], 1e-5)
        assert torch.equal(masks_e.argmax(1), G['masks_eval_argmax'].long())


def test_vqvae():
    c = ctx()
    W, G, ed = c['W'], c['G'], c['ed']
    with torch.no_grad():
        x0 = O.vae_encode(W, c['img'], ed)
        close(x0, G['x0'], 2e-5)
        zq, idx = O.vq_quantize(W, G['x0'])
        assert torch.equal(idx, G['x0_vq_idx'].long())
        close(zq, G['x0_vq'], 0.)
        dec = O.vae_decode(W, G['x0'], ed)
        close(dec[:, :, 1::2, ::2], G['x0_decoded_sub2'], 5e-5)


def test_unet_eps_and_loss():
    c = ctx()
    W, G = c['W'], c['G']
    with torch.no_grad():
        xt = O.q_sample(W, G['x0'], G['t'], G['noise'])
        close(xt, G['x_t'], 1e-6)
        eps = O.unet_forward(W, c['uplan'], G['x_t'], G['t'], G['slots'])
        close(eps, G['eps_pred'], 5e-5)
        loss = torch.nn.functional.mse_loss(eps, G['noise'])
        assert abs(float(loss) - float(G['denoise_loss'])) < 1e-5
        eps_f = O.unet_forward(W, c['uplan'], G['x_t'], G['t_frac'], G['slots'])
        close(eps_f, G['eps_pred_frac'], 5e-5)


def test_noise_schedule_and_orders():
    c = ctx()
    W, G = c['W'], c['G']
    ns = O.NoiseScheduleDiscrete(W['dm_decoder.betas'])
    close(ns.log_mean_coeff(G['ns_t']), G['ns_log_alpha'], 1e-6, 1e-6)
    close(ns.lam(G['ns_t']), G['ns_lambda'], 2e-6, 2e-6)
    close(ns.inverse_lambda(G['ns_lambda']), G['ns_inv_lambda'], 1e-6)
    outer, orders = O.dpm_orders_and_timesteps(20, 3, 1.0, 1e-3)
    close(outer, G['dpm_outer'], 0.)
    assert orders == G['dpm_orders'].tolist() == [3] * 6 + [2]


def test_train_step_gradients():
    c = ctx()
    G = c['G']
    W = {k: v.clone() for k, v in c['W'].items()}
    keys = C.load_keys()['img_based/SADiffusion/clevrtex-7slot']
    frozen = set(keys['frozen'])
    train = [k for k, _ in keys['params'] if k not in frozen]
    for k in train:
        W[k].requires_grad_(True)
    slots, _ = O.sa_encode(W, c['img'], c['rplan'], 3, training=True)
    loss, _, _ = O.ldm_loss(W, c['uplan'], c['ed'], c['img'], slots, G['t'], G['noise'])
    loss.backward()
    assert abs(float(loss.detach()) - float(G['train_loss'])) < 1e-5
    gn = sum(float((W[k].grad.double() ** 2).sum()) for k in train) ** 0.5
    assert abs(gn - float(G['grad_global_norm'])) < 1e-4 * float(G['grad_global_norm'])
    for k in G:
        if k.startswith('grad:'):
            g = W[k[5:]].grad
            ref = G[k]
            close(g, ref, 2e-5 * float(ref.abs().max()) + 1e-9)
    names = list(G['grad_norms_names'])
    mine = torch.tensor([float(W[n].grad.norm()) for n in names])
    close(mine, G['grad_norms'], 1e-7, 2e-4)


def test_dpm_solver_trajectory():
    c = ctx()
    W, G = c['W'], c['G']
    with torch.no_grad():
        eps1 = O.unet_forward(W, c['uplan'], G['x_T'], (torch.ones(2) - 1e-3) * 1000., G['slots'])
        close(eps1, G['nfe0_eps'], 5e-5)
        trace = []
        x, samples = O.ldm_sample(W, c['uplan'], c['ed'], G['slots'], G['x_T'], trace=trace)
        # north_star's bar (fp32, fixed seeds): all 7 recorded solver states and the final latent within
        # 1e-4 of the reference's trajectory (20 NFEs, 20 VQ quantisations), identical VQ codes, recon
        # PSNR within 1e-4 dB
        close(torch.stack(trace, 0), G['dpm_trace'], 1e-4)
        close(x, G['dpm_final'], 1e-4)
        assert torch.equal(O.vq_quantize(W, x)[1], O.vq_quantize(W, G['dpm_final'])[1])
        img = C.make_inputs(2)[0]
        assert float((O.psnr(samples, img) - O.psnr(G['samples'], img)).abs().max()) <= 1e-4
        assert float(O.psnr(samples, G['samples']).min()) > 80.


def test_video_model_oracle():
    """SAViDiffusion (MOVi-E config): transformer predictor, recurrence over frames, masks."""
    cfg = C.movie_cfg()
    G = C.load_golden('savidiff_b1t3.npz')
    W = C.oracle_weights_video(cfg)
    img = C.make_inputs(3, seed=11)[0].view(1, 3, 3, 128, 128)
    assert torch.equal(torch.stack([img.double().sum(), (img.double() ** 2).sum()]), G['img_checksum'])
    rplan = spec.resnet18_plan(False)
    with torch.no_grad():
        close(O.transformer_predictor(W, W['init_latents'], 2, 4), G['pred_of_init'], 1e-5)
        slots, masks = O.savi_encode(W, img, rplan, 2, True, 2, 4)
        close(slots, G['slots'], 5e-5)
        close(masks[:, :, :, ::2, 1::2], G['masks_train_sub'], 1e-5)
        assert torch.equal(masks.argmax(2), G['masks_train_argmax'].long())
        _, me = O.savi_encode(W, img, rplan, 2, False, 2, 4)
        assert torch.equal(me.argmax(2), G['masks_eval_argmax'].long())


def test_video_model_oracle_cfg2_11slots_6frames():
    """BASELINE config 2 (MOVi-D, 11 slots x 6 frames): the oracle against the reference fixture
    tests/golden/savidiff_b1t6_n11.npz (tools/gen_golden.py video11x6)."""
    cfg = C.movid_cfg()
    G = C.load_golden('savidiff_b1t6_n11.npz')
    W = C.oracle_weights_video(cfg)
    img = C.make_inputs(6, seed=13)[0].view(1, 6, 3, 128, 128)
    assert torch.equal(torch.stack([img.double().sum(), (img.double() ** 2).sum()]), G['img_checksum'])
    rplan = spec.resnet18_plan(False)
    with torch.no_grad():
        close(O.transformer_predictor(W, W['init_latents'], 2, 4), G['pred_of_init'], 1e-5)
        slots, masks = O.savi_encode(W, img, rplan, 2, True, 2, 4)
        assert slots.shape == (1, 6, 11, 192)
        close(slots, G['slots'], 5e-5)
        close(masks[:, :, :, ::2, 1::2], G['masks_train_sub'], 1e-5)
        assert torch.equal(masks.argmax(2), G['masks_train_argmax'].long())
        _, me = O.savi_encode(W, img, rplan, 2, False, 2, 4)
        assert torch.equal(me.argmax(2), G['masks_eval_argmax'].long())


def test_video_model_oracle_cfg3_15slots_6frames():
    """BASELINE config 3's shape (MOVi-E config: 15 slots, on 6-frame clips): the oracle against the
    reference fixture tests/golden/savidiff_b1t6_n15.npz (tools/gen_golden.py video15x6)."""
    cfg = C.movie_cfg()
    G = C.load_golden('savidiff_b1t6_n15.npz')
    W = C.oracle_weights_video(cfg)
    img = C.make_inputs(6, seed=17)[0].view(1, 6, 3, 128, 128)
    assert torch.equal(torch.stack([img.double().sum(), (img.double() ** 2).sum()]), G['img_checksum'])
    rplan = spec.resnet18_plan(False)
    with torch.no_grad():
        slots, masks = O.savi_encode(W, img, rplan, 2, True, 2, 4)
        assert slots.shape == (1, 6, 15, 192)
        close(slots, G['slots'], 5e-5)
        close(masks[:, :, :, ::2, 1::2], G['masks_train_sub'], 1e-5)
        assert torch.equal(masks.argmax(2), G['masks_train_argmax'].long())
        _, me = O.savi_encode(W, img, rplan, 2, False, 2, 4)
        assert torch.equal(me.argmax(2), G['masks_eval_argmax'].long())


def test_oracle_adam_is_torch_adam_with_clip():
    """Row a17 pin: O.clip_and_adam against the optimiser the reference configures (img_based/
    method.py:235-285: torch.optim.Adam, two lr groups, no weight decay) behind nerv's
    clip_grad_norm_(max_norm=clip_grad) -- 3 steps, fresh gradients per step, one group whose norm
    exceeds the clip."""
    g = torch.Generator().manual_seed(3)
    shapes = [(37, 5), (64,), (3, 3, 8, 4), (1,)]
    P0 = [torch.randn(sh, generator=g) for sh in shapes]
    lrs = [1e-4, 1e-4, 2e-4, 2e-4]
    tp = [torch.nn.Parameter(p.clone()) for p in P0]
    opt = torch.optim.Adam([dict(params=tp[:2], lr=1e-4, weight_decay=0.),
                            dict(params=tp[2:], lr=2e-4, weight_decay=0.)])
    P = [p.clone() for p in P0]
    M, V = [torch.zeros_like(p) for p in P], [torch.zeros_like(p) for p in P]
    for it in range(3):
        grads = [torch.randn(sh, generator=g) * (3.0 if it != 1 else 0.01) for sh in shapes]
        for p, gr in zip(tp, grads):
            p.grad = gr.clone()
        total_t = torch.nn.utils.clip_grad_norm_(tp, 1.0)
        opt.step()
        total = O.clip_and_adam(P, [gr.clone() for gr in grads], M, V, it + 1, lrs, clip=1.0)
        assert abs(float(total) - float(total_t)) <= 1e-6 * float(total_t)
        for a, b in zip(P, tp):
            assert float((a - b.detach()).abs().max()) <= 1e-7, it
    st = opt.state_dict()['state']
    for i, (m_, v_) in enumerate(zip(M, V)):
        assert float((m_ - st[i]['exp_avg']).abs().max()) <= 1e-7
        assert float((v_ - st[i]['exp_avg_sq']).abs().max()) <= 1e-7


def test_oracle_ema_matches_reference_litema():
    """Row a18 pin: O.ema_update against the reference's LitEma run captured in tests/golden/ema_lit.npz
    (tools/gen_golden.py ema; ddpm/ema.py:29-52): decay warm-up min(decay, (1+n)/(10+n)) and the plain
    fixed-decay mode, 4 updates each."""
    G = C.load_golden('ema_lit.npz')
    for tag, decay, n0 in (('warm', 0.9999, 0), ('flat', 0.95, -1)):
        shadow = [G['init_a'].clone(), G['init_w'].clone()]
        n = n0
        for it in range(4):
            n = O.ema_update(shadow, [G[f'{tag}_p_a_{it}'], G[f'{tag}_p_w_{it}']], n, decay)
            close(shadow[0], G[f'{tag}_s_a_{it}'], 1e-7)
            close(shadow[1], G[f'{tag}_s_w_{it}'], 1e-7)
        assert n == int(G[f'{tag}_num_updates'])


def test_plain_sa_oracle_matches_reference():
    """Row a16 (SA.decode, config 0): the oracle's spatial-broadcast decoder + reconstruction loss
    and their gradients against the reference run captured in tests/golden/sa_b2.npz; the spec's
    key order must equal the reference state_dict."""
    from slotdiffusion_amd import spec
    cfg = C.sa_plain_cfg()
    G = C.load_golden('sa_b2.npz')
    sp = spec.sa_model(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'])
    assert [p.name for p in sp] == [str(k) for k in G['state_dict_keys']]
    W = C.oracle_weights_sa(cfg)
    names = [str(n) for n in G['param_names']]
    for n in names:
        W[n].requires_grad_(True)
    img = C.make_inputs(2)[0]
    plan = spec.resnet18_plan(False)
    dplan = spec.sa_decoder_plan(cfg['resolution'], cfg['dec_dict'])
    loss, recon, masks, slots = O.sa_forward_loss(W, img, plan, dplan, cfg['dec_dict']['dec_resolution'],
                                                  cfg['slot_dict']['num_iterations'])
    loss.backward()
    loss, recon, masks, slots = loss.detach(), recon.detach(), masks.detach(), slots.detach()
    assert float((slots - G['slots']).abs().max()) <= 2e-5
    assert float((recon[:, :, 1::2, ::2] - G['recon_img_sub2']).abs().max()) <= 2e-5
    assert float((masks[:, :, 0, ::4, 1::4] - G['masks_sub4']).abs().max()) <= 2e-5
    assert torch.equal(masks[:, :, 0].argmax(1), G['masks_argmax'])
    assert abs(float(loss) - float(G['img_recon_loss'])) <= 1e-6 * max(1.0, float(G['img_recon_loss']))
    gn = torch.stack([W[n].grad.double().norm() for n in names]).float()
    # (project_q.0.bias has a mathematically zero gradient: softmax over slots is shift invariant)
    rel = ((gn - G['grad_norms']).abs() / (G['grad_norms'].abs() + 1e-9))
    assert float(rel.max()) <= 5e-3, float(rel.max())
    g0 = W['decoder.0.0.weight'].grad[::4, ::4]
    assert float((g0 - G['grad/decoder.0.0.weight']).abs().max()) <= 2e-3 * float(G['grad/decoder.0.0.weight'].abs().max())


def test_ddim_oracle_matches_reference():
    """DDIM (eta = 0, 50 steps, VQ-denoised) trajectory of the oracle against the
    reference sampler's (tests/golden/ddim_b2.npz); schedule tables exact."""
    cfg = C.clevrtex_cfg()
    G, D = C.load_golden(), C.load_golden('ddim_b2.npz')
    W = C.oracle_weights(cfg)
    ab = W['dm_decoder.alphas_bar']
    ts, a, a_prev, sig = O.ddim_schedule(ab, int(D['steps']))
    assert torch.equal(ts, D['ddim_timesteps'].long())
    assert torch.equal(a, D['ddim_alphas']) and torch.equal(a_prev, D['ddim_alphas_prev'])
    plan = spec.unet_plan(cfg['dec_dict']['unet_dict'])
    slots = G['slots']
    eps_fn = lambda xc, t: O.unet_forward(W, plan, xc, t, slots)
    q_fn = lambda x0: O.vq_quantize(W, x0)[0]
    with torch.no_grad():
        x, inter = O.ddim_sample(eps_fn, q_fn, ab, G['x_T'], int(D['steps']), log_every_t=10)
    assert inter.shape == D['ddim_inter'].shape
    assert float((inter - D['ddim_inter']).abs().max()) <= 1e-4
    assert float((x - D['ddim_final']).abs().max()) <= 1e-4
    idx = O.vq_quantize(W, x)[1]
    assert float((idx == D['ddim_final_idx'].long()).float().mean()) >= 0.999


def test_ancestral_and_x0_oracle_matches_reference():
    """ancestral sampler steps (eps and x0 prediction), posterior tables, the x0
    training target and the 'x_start' DPM-Solver wrapper of the oracle against the reference
    (tests/golden/anc_x0_b2.npz)."""
    cfg = C.clevrtex_cfg()
    G, A = C.load_golden(), C.load_golden('anc_x0_b2.npz')
    W = C.oracle_weights(cfg)
    sel = torch.tensor([0, 1, 2, 250, 500, 998, 999])
    for k in ('posterior_mean_coef1', 'posterior_mean_coef2', 'posterior_log_variance_clipped',
              'sqrt_recip_alphas_bar', 'sqrt_recipm1_alphas_bar'):
        assert torch.equal(W['dm_decoder.' + k][sel], A['tab_' + k]), k
    plan = spec.unet_plan(cfg['dec_dict']['unet_dict'])
    slots = G['slots']
    fn = lambda xc, t: O.unet_forward(W, plan, xc, t, slots)
    q_fn = lambda x0: O.vq_quantize(W, x0)[0]
    for target in ('eps', 'x0'):
        x = G['x_T']
        with torch.no_grad():
            for j, tv in enumerate(A['anc_t'].tolist()):
                t = torch.full((2,), int(tv), dtype=torch.long)
                mean, _ = O.p_mean(W, fn(x, t), x, t, q_fn, target)
                assert float((mean - A[target + '_anc_mean'][j]).abs().max()) <= 2e-4, (target, tv)
                x = O.p_sample(W, fn, x, t, A['anc_noise'], q_fn, target)
                assert float((x - A[target + '_anc_x'][j]).abs().max()) <= 2e-4, (target, tv)
                x = A[target + '_anc_x'][j]          # continue from the reference state
    # x0 target of the loss
    ed = cfg['dec_dict']['vae_dict']['enc_dec_dict']
    with torch.no_grad():
        loss, pred, _ = O.ldm_loss(W, plan, ed, C.make_inputs(2)[0], slots, G['t'].long(), G['noise'],
                                   pred_target='x0')
    assert float((pred - A['x0_pred']).abs().max()) <= 2e-4
    assert abs(float(loss) - float(A['x0_train_loss'])) <= 1e-5 * max(1.0, float(A['x0_train_loss']))
    # DPM-Solver++ with an x_start model: first outer step tight, whole trace loose (VQ ties)
    tr = []
    with torch.no_grad():
        O.dpm_solver_sample(fn, q_fn, W['dm_decoder.betas'], G['x_T'], steps=20, order=3, trace=tr,
                            model_type='x_start')
    tr = torch.stack(tr, 0)
    assert tr.shape == A['x0_dpm_trace'].shape
    assert float((tr[0] - A['x0_dpm_trace'][0]).abs().max()) <= 2e-4
    assert float((tr - A['x0_dpm_trace']).abs().mean()) <= 1e-2


def test_v_prediction_oracle_matches_reference():
    """SURVEY 8(f) row 2, pred_target='v' (video_based only): loss target, _p_mean_variance's v branch
    and DPM-Solver's model_type 'v' of the oracle against the reference (tests/golden/vpred_b1t2.npz)."""
    cfg = C.movie_cfg()
    A = C.load_golden('vpred_b1t2.npz')
    W = C.oracle_weights_video(cfg)
    plan = spec.unet_plan(cfg['dec_dict']['unet_dict'])
    ed = cfg['dec_dict']['vae_dict']['enc_dec_dict']
    slots = A['slots']
    fn = lambda xc, t: O.unet_forward(W, plan, xc, t, slots)
    q_fn = lambda x0: O.vq_quantize(W, x0)[0]
    x = A['x_T']
    with torch.no_grad():
        for j, tv in enumerate(A['anc_t'].tolist()):
            t = torch.full((2,), int(tv), dtype=torch.long)
            mean, _ = O.p_mean(W, fn(x, t), x, t, q_fn, 'v')
            assert float((mean - A['v_anc_mean'][j]).abs().max()) <= 2e-4, tv
            x = O.p_sample(W, fn, x, t, A['anc_noise'], q_fn, 'v')
            assert float((x - A['v_anc_x'][j]).abs().max()) <= 2e-4, tv
            x = A['v_anc_x'][j]
        img = C.make_inputs(2, seed=17)[0]
        loss, pred, _ = O.ldm_loss(W, plan, ed, img, slots, A['t'].long(), A['noise'], pred_target='v')
    # (slots here are the eval-mode ones of the fixture; the fixture's loss used the train-mode
    # forward of the same weights -- identical without dropout)
    assert float((pred - A['v_pred']).abs().max()) <= 2e-4
    assert abs(float(loss) - float(A['v_train_loss'])) <= 1e-5 * max(1.0, float(A['v_train_loss']))
    tr = []
    with torch.no_grad():
        O.dpm_solver_sample(fn, q_fn, W['dm_decoder.betas'], A['x_T'], steps=20, order=3, trace=tr,
                            model_type='v')
    tr = torch.stack(tr, 0)
    assert tr.shape == A['v_dpm_trace'].shape
    assert float((tr[0] - A['v_dpm_trace'][0]).abs().max()) <= 2e-4
    assert float((tr - A['v_dpm_trace']).abs().mean()) <= 1e-2


def test_eval_metrics_oracle_matches_reference():
    """ARI / FG-ARI / Hungarian mIoU / mBO of the oracle against the reference's
    eval_utils on tests/golden/metrics_b4.npz (incl. the background-only image -> nan handling and
    the fewer-predictions-than-objects case)."""
    M = C.load_golden('metrics_b4.npz')
    gt, pred = M['gt'].long(), M['pred'].long()
    assert torch.equal(O.adjusted_rand_index(gt, pred, False), M['ari_per_image'])
    assert torch.equal(O.adjusted_rand_index(gt, pred, True), M['fari_per_image'])
    assert torch.equal(O.adjusted_rand_index(gt.view(2, 2, 32, 32), pred.view(2, 2, 32, 32), False), M['ari_video'])
    r = O.seg_metrics(gt, pred)
    for k in ('ari', 'fari', 'miou', 'fmiou', 'mbo'):
        assert abs(float(r[k]) - float(M[k])) <= 1e-6, (k, r[k], float(M[k]))
    x = torch.rand(2, 3, 16, 16, generator=torch.Generator().manual_seed(3))
    y = (x + 0.05 * torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(4))).clamp(0, 1)
    assert abs(O.psnr_metric(x, y) - float(np.mean([10 * np.log10(1.0 / float(((x[i] - y[i]).double() ** 2).mean())) for i in range(2)]))) < 1e-9


def test_savi_oracle_matches_reference():
    """video SA baseline (registry 'SAVi'): oracle forward / loss / gradients
    and the eval metrics against the reference run in tests/golden/savi_b1t3.npz; spec key order =
    the reference state_dict."""
    cfg = C.savi_cfg()
    G = C.load_golden('savi_b1t3.npz')
    sp = spec.savi_model(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'], cfg['pred_dict'])
    assert [p.name for p in sp] == [str(k) for k in G['state_dict_keys']]
    W = C.oracle_weights_savi(cfg)
    names = [str(n) for n in G['grad_norms_names']]
    for n in names:
        W[n].requires_grad_(True)
    img = C.make_inputs(3, seed=11)[0].view(1, 3, 3, 128, 128)
    loss, recon, masks, slots = O.savi_forward_loss(
        W, img, spec.resnet18_plan(False), spec.sa_decoder_plan(cfg['resolution'], cfg['dec_dict']),
        cfg['dec_dict']['dec_resolution'], cfg['slot_dict']['num_iterations'],
        cfg['pred_dict']['pred_num_layers'], cfg['pred_dict']['pred_num_heads'])
    loss.backward()
    loss, recon, masks, slots = loss.detach(), recon.detach(), masks.detach(), slots.detach()
    assert float((slots - G['slots']).abs().max()) <= 2e-5
    assert float((recon[:, :, :, 1::2, ::2] - G['recon_img_sub2']).abs().max()) <= 2e-5
    assert float((masks[:, :, :, 0, ::4, 1::4] - G['masks_sub4']).abs().max()) <= 2e-5
    am = masks[:, :, :, 0].argmax(2)
    assert float((am == G['masks_argmax'].long()).float().mean()) >= 0.9999
    assert abs(float(loss) - float(G['img_recon_loss'])) <= 1e-6
    gn = torch.stack([W[n].grad.double().norm() for n in names]).float()
    rel = (gn - G['grad_norms']).abs() / (G['grad_norms'].abs() + 1e-7)
    assert float(rel.max()) <= 5e-3, float(rel.max())
    r = O.seg_metrics(G['gt_masks'].long().flatten(1, 2), G['masks_argmax'].long().flatten(1, 2))
    for k in ('ari', 'fari', 'miou', 'fmiou', 'mbo'):
        assert abs(float(r[k]) - float(G['eval_' + k])) <= 1e-6, k


def test_vqvae_standalone_oracle_matches_reference():
    """Registry model 'VQVAE' in eval (VQVAE.forward + calc_eval_loss): the oracle against the
    reference run in tests/golden/vqvae_b2.npz; spec key order = the reference state_dict."""
    cfg = C.clevrtex_cfg()
    va = cfg['dec_dict']['vae_dict']
    V = C.load_golden('vqvae_b2.npz')
    sp = spec.vqvae_model(va['enc_dec_dict'], va['vq_dict'])
    assert [p.name for p in sp] == [str(k) for k in V['state_dict_keys']]
    W = C.oracle_weights_vqvae(cfg)
    img = C.make_inputs(2)[0]
    with torch.no_grad():
        r = O.vqvae_forward(W, img, va['enc_dec_dict'])
        r1 = O.vqvae_forward(W, img, va['enc_dec_dict'], percept_loss_w=1.0)
    assert torch.equal(r['token_id'], V['token_id'].long())
    assert float((r['recon'][:, :, 1::2, ::2] - V['recon_sub2']).abs().max()) <= 2e-5
    for k in ('quant_loss', 'recon_loss', 'recon_mse'):
        assert abs(float(r[k]) - float(V[k])) <= 1e-6 * max(1.0, abs(float(V[k]))), k
    assert abs(float(r1['recon_loss']) - float(V['recon_l1'])) <= 1e-6


def test_vqvae_training_gradients_oracle_matches_reference():
    """gradients of recon_loss + quant_loss through decoder, straight-through
    quantizer and encoder of the oracle against the reference's (tests/golden/vqvae_b2.npz), for
    the MSE and the L1 reconstruction loss."""
    cfg = C.clevrtex_cfg()
    va = cfg['dec_dict']['vae_dict']
    V = C.load_golden('vqvae_b2.npz')
    names = [str(n) for n in V['grad_norms_names']]
    img = C.make_inputs(2)[0]
    for key, w in (('grad_norms', 0.), ('grad_norms_l1', 1.)):
        W = C.oracle_weights_vqvae(cfg)
        for n in names:
            W[n].requires_grad_(True)
        r = O.vqvae_forward(W, img, va['enc_dec_dict'], percept_loss_w=w)
        (r['recon_loss'] + r['quant_loss']).backward()
        gn = torch.stack([W[n].grad.double().norm() for n in names]).float()
        rel = (gn - V[key]).abs() / (V[key].abs() + 1e-9)
        assert float(rel.max()) <= 5e-3, (key, float(rel.max()))
        if w == 0.:
            assert abs(float(r['recon_loss'].detach()) - float(V['train_recon_loss'])) <= 1e-6
            for n in ('quantize.embedding.weight', 'decoder.mid.attn_1.k.weight'):
                ref = V['grad/' + n]
                assert float((W[n].grad - ref).abs().max()) <= 2e-3 * float(ref.abs().max()) + 1e-9, n


def test_ssim_restatement_and_slot_shuffling():
    """SSIM (parity unpinned: skimage is absent): the oracle's scipy restatement against a direct
    evaluation of the published definition, its basic properties, and the compositional-generation
    slot shuffling against the reference's in-place loop (test_comp_gen.py:25-31)."""
    import numpy as np
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 24, 20, generator=g).numpy()
    y = np.clip(x + 0.1 * torch.randn(2, 3, 24, 20, generator=g).numpy(), 0, 1)
    s = O.ssim_metric(x, y)
    # direct definition: 11x11 outer-product Gaussian window at every interior pixel
    w = np.exp(-0.5 * (np.arange(-5, 6) ** 2) / 1.5 ** 2)
    w /= w.sum()
    W2 = np.outer(w, w)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for i in range(2):
        ch = []
        for c in range(3):
            a, b = x[i, c] * 255., y[i, c] * 255.
            acc = []
            for cy in range(5, 24 - 5):
                for cx in range(5, 20 - 5):
                    pa, pb = a[cy - 5:cy + 6, cx - 5:cx + 6], b[cy - 5:cy + 6, cx - 5:cx + 6]
                    ux, uy = (W2 * pa).sum(), (W2 * pb).sum()
                    vx, vy = (W2 * pa * pa).sum() - ux * ux, (W2 * pb * pb).sum() - uy * uy
                    vxy = (W2 * pa * pb).sum() - ux * uy
                    acc.append(((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2)))
            ch.append(np.mean(acc))
        vals.append(np.mean(ch))
    assert abs(s - float(np.mean(vals))) < 1e-8
    assert abs(O.ssim_metric(x, x) - 1.0) < 1e-12 and 0.0 < s < 1.0
    assert O.ssim_metric(x, y) == O.ssim_metric(y, x)
    # slot shuffling: the reference's in-place loop
    slots = torch.randn(5, 2, 7, 4, generator=g)
    ref = slots.clone()
    for i in range(7):
        ref[:, :, i] = torch.cat([ref[i:, :, i], ref[:i, :, i]])
    assert torch.equal(O.shuffle_slots(slots), ref)
    from slotdiffusion_amd import metrics
    assert torch.equal(metrics.shuffle_slots(slots), ref)
    assert torch.equal(metrics.shuffle_slots(slots[:, 0]), O.shuffle_slots(slots[:, 0]))


def test_dino_coco_config_oracle_matches_reference():
    """SURVEY 8(f) row 4 / BASELINE config 5 (COCO 224^2, DINO ViT-S/8 encoder, 7 slots of 256, latent
    56^2): the oracle's ViT restatement + the usual slot / VQ-VAE / UNet path against the reference run
    in tests/golden/sadiff_dino_b1.npz (tools/gen_golden.py dino: the reference model with
    transformers' ViTModel class standing in for the hub download)."""
    cfg = C.dino_coco_cfg()
    G = C.load_golden('sadiff_dino_b1.npz')
    W = C.oracle_weights(cfg)
    img, noise = C.dino_inputs()
    assert torch.equal(torch.stack([img.double().sum(), (img.double() ** 2).sum()]), G['img_checksum'])
    assert torch.equal(noise, G['noise'])
    meta, vres = spec.encoder_plan(cfg['resolution'], cfg['enc_dict'])
    assert vres == (28, 28) and meta['ntok'] == 785
    with torch.no_grad():
        feat = O.dino_vit_forward(W, img, meta)
        close(feat[:, ::8], G['dino_feat_sub'], 2e-4)
        slots, masks = O.sa_encode(W, img, meta, cfg['slot_dict']['num_iterations'], training=True)
        close(slots, G['slots'], 2e-4)
        assert torch.equal(masks.argmax(1), G['masks_train_argmax'].long())
        _, me = O.sa_encode(W, img, meta, cfg['slot_dict']['num_iterations'], training=False)
        assert float((me.argmax(1) == G['masks_eval_argmax'].long()).float().mean()) > 0.9995
        plan = spec.unet_plan(cfg['dec_dict']['unet_dict'])
        ed = cfg['dec_dict']['vae_dict']['enc_dec_dict']
        loss, pred, x0 = O.ldm_loss(W, plan, ed, img, G['slots'], G['t'].long(), noise)
    close(x0, G['x0'], 2e-4)
    close(pred, G['eps_pred'], 5e-4)
    assert abs(float(loss) - float(G['train_loss'])) <= 1e-4 * max(1.0, float(G['train_loss']))
