"""End-to-end parity of the HIP path against golden tensors captured from the real reference
(tests/golden/sadiff_b2.npz): slot encoder + Slot Attention masks, VQ-VAE encode / quantise /
decode, UNet eps-prediction, denoising loss and the 20-NFE DPM-Solver++ sampler.

fp32 compute path: index outputs bit-exact; float outputs within 1e-4.
bf16 compute path: deviations are reported and bounded loosely (it is the throughput path)."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from tests import common as C
from tests.detfill import det_fill_, is_buffer_name

pytestmark = pytest.mark.gpu
_cache = {}
REPORT = {}


def ctx(dtype=torch.float32):
    if 'G' not in _cache:
        _cache['G'] = C.load_golden()
        _cache['img'] = C.make_inputs(2)[0]
    if dtype not in _cache:
        from slotdiffusion_amd.models import SADiffusion
        cfg = C.clevrtex_cfg()
        m = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                        cfg['loss_dict'], compute_dtype=dtype)
        det_fill_(m.state_dict().items(), skip=is_buffer_name)
        m.train_dropout = 0.0
        m = m.cuda().eval()
        m.use_graph = False
        _cache[dtype] = m
    return _cache[dtype], _cache['G'], _cache['img'].cuda()


def maxerr(a, b):
    return float((a.float().cpu() - b.float()).abs().max())


def _dump():
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_report.json', 'w') as f:
        json.dump(REPORT, f, indent=1)


def test_encode_slots_and_masks_fp32():
    m, G, img = ctx()
    m.train()
    slots, masks = m.encode(img)
    m.eval()
    REPORT['slots_maxerr'] = maxerr(slots, G['slots'])
    REPORT['masks_train_maxerr'] = maxerr(masks, G['masks_train'])
    agree = float((masks.cpu().argmax(1) == G['masks_train_argmax'].long()).float().mean())
    REPORT['masks_train_argmax_agree'] = agree
    slots_e, masks_e = m.encode(img)
    REPORT['masks_eval_maxerr'] = maxerr(masks_e[:, :, 1::4, 2::4], G['masks_eval_sub4'])
    agree_e = float((masks_e.cpu().argmax(1) == G['masks_eval_argmax'].long()).float().mean())
    REPORT['masks_eval_argmax_agree'] = agree_e
    _dump()
    assert REPORT['slots_maxerr'] <= 1e-4
    assert REPORT['masks_train_maxerr'] <= 1e-4 and REPORT['masks_eval_maxerr'] <= 1e-4
    assert agree == 1.0 and agree_e == 1.0          # segmentation indices bit-exact


def test_vqvae_fp32():
    m, G, img = ctx()
    vae = m.dm_decoder.vae
    x0 = vae.encode(img)
    REPORT['x0_maxerr'] = maxerr(x0, G['x0'])
    idx = vae.quantize_indices(G['x0'].cuda())
    REPORT['vq_idx_agree'] = float((idx.cpu() == G['x0_vq_idx'].long()).float().mean())
    zq = vae.quantize(G['x0'].cuda())
    REPORT['x0_vq_maxerr'] = maxerr(zq, G['x0_vq'])
    dec = vae.decode(G['x0'].cuda())
    REPORT['x0_decoded_maxerr'] = maxerr(dec[:, :, 1::2, ::2], G['x0_decoded_sub2'])
    _dump()
    assert REPORT['x0_maxerr'] <= 1e-4
    assert REPORT['vq_idx_agree'] == 1.0 and REPORT['x0_vq_maxerr'] == 0.0
    assert REPORT['x0_decoded_maxerr'] <= 1e-4        # (measured 2.1e-5: the same bar as every other fp32 tensor)


def test_unet_eps_and_loss_fp32():
    m, G, img = ctx()
    xt = m._latent_nhwc(G['x_t'].cuda())
    eps = m._unet_eps(xt, G['t'].float().cuda(), G['slots'].cuda())
    from slotdiffusion_amd import ops
    eps = ops.nhwc_to_nchw(eps, 3)
    REPORT['eps_maxerr'] = maxerr(eps, G['eps_pred'])
    REPORT['eps_mse_vs_ref'] = float(((eps.cpu() - G['eps_pred']) ** 2).mean())
    epsf = ops.nhwc_to_nchw(m._unet_eps(xt, G['t_frac'].cuda(), G['slots'].cuda()), 3)
    REPORT['eps_frac_maxerr'] = maxerr(epsf, G['eps_pred_frac'])
    loss = m.calc_train_loss(dict(img=img, t=G['t'].cuda(), noise=G['noise'].cuda()),
                             dict(slots=G['slots'].cuda()))['denoise_loss']
    REPORT['loss'] = float(loss)
    REPORT['loss_ref'] = float(G['denoise_loss'])
    _dump()
    assert REPORT['eps_maxerr'] <= 1e-4 and REPORT['eps_frac_maxerr'] <= 1e-4
    assert abs(REPORT['loss'] - REPORT['loss_ref']) <= 1e-4      # eps-MSE within 1e-4


def test_dpm_solver_sampling_fp32():
    m, G, img = ctx()
    from slotdiffusion_amd import ops
    dm = m.dm_decoder
    x, inter = dm.generate_imgs(cond=G['slots'].cuda(), batch_size=2, x_T=G['x_T'].cuda(),
                                ret_intermed=True)
    REPORT['dpm_step0_maxerr'] = maxerr(inter[0], G['dpm_trace'][0])
    REPORT['dpm_trace_maxerr'] = [maxerr(inter[i], G['dpm_trace'][i]) for i in range(7)]
    REPORT['dpm_final_frac_gt_1e-3'] = float(((x.cpu() - G['dpm_final']).abs() > 1e-3).float().mean())
    samples = dm.vae.decode(x)
    from oracle import slotdiff_oracle as O
    ps = O.psnr(samples.cpu(), G['samples'])
    REPORT['samples_psnr_vs_ref_db'] = [float(v) for v in ps]
    # recon PSNR (vs the input image) must agree with the reference's recon PSNR
    REPORT['recon_psnr_ours'] = [float(v) for v in O.psnr(samples.cpu(), img.cpu())]
    REPORT['recon_psnr_ref'] = [float(v) for v in O.psnr(G['samples'], img.cpu())]
    idx_ours = dm.vae.quantize_indices(x)
    idx_ref = dm.vae.quantize_indices(G['dpm_final'].cuda())
    REPORT['final_code_agree'] = float((idx_ours == idx_ref).float().mean())
    REPORT['dpm_final_maxerr'] = maxerr(x, G['dpm_final'])
    _dump()
    # north_star's bar, fp32 on fixed seeds: every solver state and the final latent within 1e-4 of the
    # reference's, the same VQ codes, recon PSNR within 1e-4 dB (measured: 1e-6, 100 %, identical)
    assert max(REPORT['dpm_trace_maxerr']) <= 1e-4, REPORT['dpm_trace_maxerr']
    assert REPORT['dpm_final_maxerr'] <= 1e-4
    assert REPORT['final_code_agree'] == 1.0
    assert max(abs(a - b) for a, b in zip(REPORT['recon_psnr_ours'], REPORT['recon_psnr_ref'])) <= 1e-4
    assert min(REPORT['samples_psnr_vs_ref_db']) > 80.


def test_log_images_api_fp32():
    m, G, img = ctx()
    torch.manual_seed(0)
    out = m(dict(img=img), log_images=True, use_dpm=True, same_noise=True, ret_intermed=False)
    assert out['samples'].shape == (2, 3, 128, 128) and out['masks'].shape == (2, 7, 128, 128)
    assert torch.isfinite(out['samples']).all()


def test_bf16_path_deviation():
    m, G, img = ctx(torch.bfloat16)
    from slotdiffusion_amd import ops
    slots, masks = m.encode(img)
    REPORT['bf16_slots_maxerr'] = maxerr(slots, G['slots'])
    REPORT['bf16_masks_eval_argmax_agree'] = float(
        (masks.cpu().argmax(1) == G['masks_eval_argmax'].long()).float().mean())
    xt = m._latent_nhwc(G['x_t'].cuda())
    eps = ops.nhwc_to_nchw(m._unet_eps(xt, G['t'].float().cuda(), G['slots'].cuda()), 3)
    ref = G['eps_pred']
    REPORT['bf16_eps_rel_l2'] = float((eps.cpu() - ref).norm() / ref.norm())
    x0 = m.dm_decoder.vae.encode(img)
    REPORT['bf16_x0_rel_l2'] = float((x0.cpu() - G['x0']).norm() / G['x0'].norm())
    _dump()
    # measured on MI355X: eps rel-L2 1.3 %, mask agreement 99.5 % -- bounds leave ~1.5x head-room
    assert REPORT['bf16_eps_rel_l2'] < 0.02 and REPORT['bf16_x0_rel_l2'] < 0.02
    assert REPORT['bf16_masks_eval_argmax_agree'] > 0.99


def _video_parity(cfg, fixture, T, seed, tag):
    from slotdiffusion_amd.models import SAViDiffusion
    G = C.load_golden(fixture)
    m = SAViDiffusion(cfg['resolution'], T, cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                      cfg['pred_dict'], cfg['loss_dict'], compute_dtype=torch.float32)
    det_fill_(m.state_dict().items(), skip=is_buffer_name)
    m.train_dropout = m.pred_dropout = 0.0
    m = m.cuda()
    img = C.make_inputs(T, seed=seed)[0].view(1, T, 3, 128, 128).cuda()
    m.train()
    m.grad_arena().zero_()
    out = m(dict(img=img))
    R = {}
    R[tag + '_slots_maxerr'] = maxerr(out['slots'].detach(), G['slots'])
    R[tag + '_masks_argmax_agree'] = float(
        (out['masks'].cpu().argmax(2) == G['masks_train_argmax'].long()).float().mean())
    loss = m.calc_train_loss(dict(img=img, t=G['t'].cuda(), noise=G['noise'].cuda()), out)[
        'denoise_loss']
    loss.backward()
    R[tag + '_loss'] = float(loss.detach())
    R[tag + '_loss_ref'] = float(G['train_loss'])
    named = dict(m.named_parameters())
    names = [str(n) for n in G['grad_norms_names']]
    mine = torch.tensor([float(named[n].grad.norm()) for n in names])
    ref = G['grad_norms']
    big = ref > 1e-6
    rel = ((mine - ref).abs() / (ref.abs() + 1e-12))[big]
    R[tag + '_grad_norm_max_rel_vs_reference'] = float(rel.max())
    errs = {k[5:]: float((named[k[5:]].grad.float().cpu() - G[k]).abs().max() / G[k].abs().max())
            for k in G if k.startswith('grad:')}
    R[tag + '_grad_tensor_rel_err'] = errs
    m.eval()
    oe = m(dict(img=img))
    R[tag + '_masks_eval_argmax_agree'] = float(
        (oe['masks'].cpu().argmax(2) == G['masks_eval_argmax'].long()).float().mean())
    REPORT.update(R)
    _dump()
    assert R[tag + '_slots_maxerr'] <= 1e-4 and R[tag + '_masks_argmax_agree'] == 1.0
    assert R[tag + '_masks_eval_argmax_agree'] == 1.0
    assert abs(R[tag + '_loss'] - R[tag + '_loss_ref']) <= 1e-4
    assert R[tag + '_grad_norm_max_rel_vs_reference'] <= 2e-2 and max(errs.values()) <= 2e-2


def test_video_model_fp32():
    """SAViDiffusion (MOVi-E config, 15 slots, T=3): predictor, per-frame Slot Attention, masks, and
    the train-step gradients of every tensor against the reference's (tests/golden/savidiff_b1t3.npz)."""
    _video_parity(C.movie_cfg(), 'savidiff_b1t3.npz', 3, 11, 'video')


def test_video_model_cfg2_11slots_6frames_fp32():
    """BASELINE config 2: video SAVi+LDM on the MOVi-D config with 11 slots and 6-frame clips
    (tests/golden/savidiff_b1t6_n11.npz, generated from the reference by tools/gen_golden.py
    video11x6): slots, train / eval mask argmax, loss and all parameter-gradient norms."""
    _video_parity(C.movid_cfg(), 'savidiff_b1t6_n11.npz', 6, 13, 'video_cfg2')


def test_video_model_cfg3_15slots_6frames_fp32():
    """BASELINE config 3's shape: the MOVi-E config (15 slots) on 6-frame clips
    (tests/golden/savidiff_b1t6_n15.npz, tools/gen_golden.py video15x6)."""
    _video_parity(C.movie_cfg(), 'savidiff_b1t6_n15.npz', 6, 17, 'video_cfg3')


def test_video_model_bf16_deviation():
    """The benchmarked dtype on the video model (BASELINE config 2: 11 slots x 6 frames): slots, mask
    agreement, the denoiser's eps on the reference slots and the gradient norms of the bf16 compute path
    (head-dim-48 predictor attention on the VALU kernels, per-frame recurrence) against the fp32
    reference fixture.  eps rel-L2 < 2 % as on the image model (measured on MI355X: 1.15 %; loss 0.07 %,
    gradient-norm median 0.54 %); mask agreement > 98 % -- measured 98.9 % on the training-resolution
    masks: the per-frame recurrence carries bf16 slot differences through six frames, the image model's
    single frame sits at 99.5 %."""
    from slotdiffusion_amd import ops
    from slotdiffusion_amd.models import SAViDiffusion
    cfg, T = C.movid_cfg(), 6
    G = C.load_golden('savidiff_b1t6_n11.npz')
    m = SAViDiffusion(cfg['resolution'], T, cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                      cfg['pred_dict'], cfg['loss_dict'], compute_dtype=torch.bfloat16)
    det_fill_(m.state_dict().items(), skip=is_buffer_name)
    m.train_dropout = m.pred_dropout = 0.0
    m = m.cuda()
    img = C.make_inputs(T, seed=13)[0].view(1, T, 3, 128, 128).cuda()
    m.train()
    m.grad_arena().zero_()
    out = m(dict(img=img))
    R = {}
    R['video_bf16_slots_rel_l2'] = float((out['slots'].detach().float().cpu() - G['slots']).norm() / G['slots'].norm())
    R['video_bf16_masks_train_agree'] = float(
        (out['masks'].cpu().argmax(2) == G['masks_train_argmax'].long()).float().mean())
    loss = m.calc_train_loss(dict(img=img, t=G['t'].cuda(), noise=G['noise'].cuda()), out)['denoise_loss']
    loss.backward()
    R['video_bf16_loss_rel'] = abs(float(loss.detach()) - float(G['train_loss'])) / float(G['train_loss'])
    named = dict(m.named_parameters())
    names = [str(n) for n in G['grad_norms_names']]
    mine = torch.tensor([float(named[n].grad.norm()) for n in names])
    ref = G['grad_norms']
    big = ref > 1e-6
    rel = ((mine - ref).abs() / (ref.abs() + 1e-12))[big]
    R['video_bf16_grad_norm_rel_median'] = float(rel.median())
    R['video_bf16_grad_norm_rel_p90'] = float(rel.kthvalue(max(1, int(0.9 * rel.numel()))).values)
    # eps on the REFERENCE's slots: fp32 model of the same weights as the yardstick (the fixture stores
    # the loss, not eps itself)
    m32 = SAViDiffusion(cfg['resolution'], T, cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                        cfg['pred_dict'], cfg['loss_dict'], compute_dtype=torch.float32)
    det_fill_(m32.state_dict().items(), skip=is_buffer_name)
    m32 = m32.cuda().eval()
    m.eval()
    with torch.no_grad():
        slots = G['slots'].flatten(0, 1).cuda()
        xt = m._latent_nhwc(C.make_inputs(T, seed=13)[3][:T].cuda())
        tt = G['t'].float().cuda()
        e16 = ops.nhwc_to_nchw(m._unet_eps(xt, tt, slots), 3).float()
        e32 = ops.nhwc_to_nchw(m32._unet_eps(xt, tt, slots), 3)
        R['video_bf16_eps_rel_l2'] = float((e16 - e32).norm() / e32.norm())
        oe = m(dict(img=img))
    R['video_bf16_masks_eval_agree'] = float(
        (oe['masks'].cpu().argmax(2) == G['masks_eval_argmax'].long()).float().mean())
    REPORT.update(R)
    _dump()
    assert R['video_bf16_eps_rel_l2'] < 0.02, R
    assert R['video_bf16_masks_train_agree'] > 0.98 and R['video_bf16_masks_eval_agree'] > 0.98, R
    assert R['video_bf16_loss_rel'] < 0.02 and R['video_bf16_grad_norm_rel_median'] < 0.05, R


def _plain_sa(dtype):
    from slotdiffusion_amd.models import SA
    cfg = C.sa_plain_cfg()
    m = SA(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'], cfg['loss_dict'],
           compute_dtype=dtype)
    det_fill_(m.state_dict().items(), skip=is_buffer_name)
    return m.cuda().train()


def test_plain_sa_autoencoder_fp32():
    """Row a16 / BASELINE config 0: plain Slot Attention auto-encoder (transposed-conv decoder,
    slot softmax compositing, reconstruction loss) forward + backward against the reference run
    in tests/golden/sa_b2.npz."""
    G = C.load_golden('sa_b2.npz')
    img = C.make_inputs(2)[0].cuda()
    m = _plain_sa(torch.float32)
    m.grad_arena().zero_()
    out = m(dict(img=img))
    loss = m.calc_train_loss(dict(img=img), out)['img_recon_loss']
    loss.backward()
    torch.cuda.synchronize()
    REPORT['sa_slots'] = maxerr(out['slots'], G['slots'])
    REPORT['sa_recon'] = maxerr(out['recon_img'][:, :, 1::2, ::2], G['recon_img_sub2'])
    REPORT['sa_masks'] = maxerr(out['masks'][:, :, 0, ::4, 1::4], G['masks_sub4'])
    REPORT['sa_recons'] = maxerr(out['recons'][:, :, :, 1::4, ::4], G['recons_sub4'])
    REPORT['sa_loss_rel'] = abs(float(loss) - float(G['img_recon_loss'])) / float(G['img_recon_loss'])
    agree = float((out['masks'][:, :, 0].argmax(1).cpu() == G['masks_argmax']).float().mean())
    REPORT['sa_mask_argmax_agree'] = agree
    names = [str(n) for n in G['param_names']]
    named = dict(m.named_parameters())
    gn = torch.stack([named[n].grad.double().norm().cpu() for n in names]).float()
    rel = (gn - G['grad_norms']).abs() / (G['grad_norms'].abs() + 1e-9)
    REPORT['sa_gradnorm_rel_max'] = float(rel.max())
    g0 = named['decoder.0.0.weight'].grad.cpu()[::4, ::4]
    REPORT['sa_deconv_wgrad_rel'] = float((g0 - G['grad/decoder.0.0.weight']).abs().max() /
                                          G['grad/decoder.0.0.weight'].abs().max())
    for k in ('decoder.3.0.bias', 'decoder.4.weight', 'decoder_pos_embedding.dense.weight', 'init_latents'):
        ref = G['grad/' + k]
        REPORT['sa_grad/' + k] = float((named[k].grad.cpu() - ref).abs().max() / ref.abs().max())
    _dump()
    assert REPORT['sa_slots'] <= 1e-4 and REPORT['sa_recon'] <= 1e-4 and REPORT['sa_masks'] <= 1e-4
    assert REPORT['sa_recons'] <= 1e-4 and REPORT['sa_loss_rel'] <= 1e-5
    assert agree == 1.0
    assert REPORT['sa_gradnorm_rel_max'] <= 1e-2
    assert REPORT['sa_deconv_wgrad_rel'] <= 5e-3
    for k in ('decoder.3.0.bias', 'decoder.4.weight', 'decoder_pos_embedding.dense.weight', 'init_latents'):
        assert REPORT['sa_grad/' + k] <= 1e-2, k


def test_plain_sa_autoencoder_bf16_close():
    G = C.load_golden('sa_b2.npz')
    img = C.make_inputs(2)[0].cuda()
    m = _plain_sa(torch.bfloat16)
    m.grad_arena().zero_()
    out = m(dict(img=img))
    loss = m.calc_train_loss(dict(img=img), out)['img_recon_loss']
    loss.backward()
    REPORT['sa_bf16_loss_rel'] = abs(float(loss) - float(G['img_recon_loss'])) / float(G['img_recon_loss'])
    named = dict(m.named_parameters())
    names = [str(n) for n in G['param_names']]
    gn = torch.stack([named[n].grad.double().norm().cpu() for n in names]).float()
    big = G['grad_norms'] > 1e-6 * G['grad_norms'].max()
    rel = ((gn - G['grad_norms']).abs() / G['grad_norms'].abs())[big]
    REPORT['sa_bf16_gradnorm_rel_median'] = float(rel.median())
    _dump()
    assert REPORT['sa_bf16_loss_rel'] <= 2e-2
    assert float(rel.median()) <= 5e-2


def test_ddim_sampler_fp32():
    """DDIM (eta 0, 50 steps, VQ-denoised; generate_imgs(use_ddim=True)) against the reference run
    in tests/golden/ddim_b2.npz.  A long trajectory through the discrete VQ denoiser is chaotic (a
    1e-6 difference in eps, amplified by 1/sqrt(a_t) ~ 70 at large t, flips a code now and then and
    eps feeds back into x), so the parity points are: the schedule, the first logged state, and
    single steps restarted from the reference's own intermediate states."""
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd import dpm, ops, spec
    m, G, img = ctx()
    D = C.load_golden('ddim_b2.npz')
    steps = int(D['steps'])
    plan = dpm.ddim_plan(m.dm_decoder.alphas_bar.detach().float().cpu(), steps)
    assert [st['t'] for st in plan] == list(reversed(D['ddim_timesteps'].tolist()))
    slots = G['slots'].cuda()
    out, inter = m.dm_decoder.generate_imgs(slots, batch_size=2, ret_intermed=True, use_dpm=False,
                                            use_ddim=True, x_T=G['x_T'].cuda(), ddim_steps=steps,
                                            log_every_t=10)
    assert inter.shape == D['ddim_inter'].shape
    REPORT['ddim_first_logged_state'] = maxerr(inter[1], D['ddim_inter'][1])
    assert maxerr(inter[0], D['ddim_inter'][0]) == 0.0 and REPORT['ddim_first_logged_state'] <= 1e-4
    cfg = C.clevrtex_cfg()
    W = C.oracle_weights(cfg)
    uplan = spec.unet_plan(cfg['dec_dict']['unet_dict'])
    worst = 1.0
    for k, index in ((2, 40), (4, 20), (6, 0)):          # inter[k] = state after step `index`
        if index == 0:
            continue
        st = [s for s in plan if s['index'] == index - 1]
        x_ref_in = D['ddim_inter'][k]
        x_dev = next(m._ddim_steps(ops.nchw_to_nhwc(x_ref_in.cuda(), torch.float32, 4), slots, st))[0]
        t = torch.full((2,), st[0]['t'], dtype=torch.long)
        with torch.no_grad():
            e = O.unet_forward(W, uplan, x_ref_in, t, G['slots'])
            x0 = O.vq_quantize(W, (x_ref_in - st[0]['som'] * e) / st[0]['sqrt_a'])[0]
            x_cpu = st[0]['sqrt_a_prev'] * x0 + st[0]['dir'] * e
        ok = ((ops.nhwc_to_nchw(x_dev, 3).cpu() - x_cpu).abs() <= 1e-4).float().mean()
        worst = min(worst, float(ok))
    REPORT['ddim_single_step_agree'] = worst
    _dump()
    assert worst >= 0.995


def test_ancestral_sampler_and_x0_target_fp32():
    """ancestral sampling steps (generate_imgs(use_dpm=False, use_ddim=False)
    path) and the x0-prediction variant (loss target, ancestral step, DPM-Solver 'x_start') against
    the reference run in tests/golden/anc_x0_b2.npz.  Single steps with the fixture's explicit
    noise are the parity points (each step is restarted from the reference state: the VQ denoiser
    makes long trajectories chaotic)."""
    from slotdiffusion_amd import ops
    from slotdiffusion_amd.models import SADiffusion
    m, G, img = ctx()
    A = C.load_golden('anc_x0_b2.npz')
    slots = G['slots'].cuda()
    ts = [int(v) for v in A['anc_t'].tolist()]
    nz = A['anc_noise'].cuda()

    def steps_of(model, key):
        agree = 1.0
        x_in = G['x_T']
        for j, tv in enumerate(ts):
            x = ops.nchw_to_nhwc(x_in.cuda(), torch.float32, 4)
            x_dev = next(model._ancestral_steps(x, slots, [tv], noises=[nz]))[0]
            ok = ((ops.nhwc_to_nchw(x_dev, 3).cpu() - A[key][j]).abs() <= 1e-4).float().mean()
            agree = min(agree, float(ok))
            x_in = A[key][j]
        return agree

    REPORT['ancestral_eps_step_agree'] = steps_of(m, 'eps_anc_x')
    # the full sampler API on a 12-timestep schedule (same weights): runs, logs T//4-spaced states
    cfg = C.clevrtex_cfg()
    cfg['dec_dict']['diffusion_dict']['timesteps'] = 12
    ms = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                     cfg['loss_dict'], compute_dtype=torch.float32)
    det_fill_(ms.state_dict().items(), skip=is_buffer_name)
    ms = ms.cuda().eval()
    ms.use_graph = False
    out, inter = ms.dm_decoder.generate_imgs(slots, batch_size=2, ret_intermed=True, use_dpm=False,
                                             use_ddim=False, x_T=G['x_T'].cuda(), log_every_t=4)
    assert out.shape == (2, 3, 32, 32) and inter.shape[0] == 1 + 4 and bool(torch.isfinite(out).all())
    assert maxerr(inter[0], G['x_T']) == 0.0 and maxerr(inter[-1], out.cpu()) == 0.0
    del ms
    # ---- x0-prediction model
    cfg = C.clevrtex_cfg()
    cfg['dec_dict']['diffusion_dict']['pred_target'] = 'x0'
    mx = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                     cfg['loss_dict'], compute_dtype=torch.float32)
    det_fill_(mx.state_dict().items(), skip=is_buffer_name)
    mx.train_dropout = 0.0
    mx = mx.cuda().eval()
    mx.use_graph = False
    REPORT['ancestral_x0_step_agree'] = steps_of(mx, 'x0_anc_x')
    x, tr = mx.dm_decoder.generate_imgs(cond=slots, batch_size=2, x_T=G['x_T'].cuda(), ret_intermed=True)
    REPORT['x0_dpm_step0_maxerr'] = maxerr(tr[0], A['x0_dpm_trace'][0])
    REPORT['x0_dpm_final_frac_gt_1e-3'] = float(((x.cpu() - A['x0_dpm_final']).abs() > 1e-3).float().mean())
    # training loss with the x0 target + gradient norms
    mx.train()
    for p in mx.parameters():
        p.grad = None
    data = dict(img=img, t=G['t'].long().cuda(), noise=G['noise'].cuda())
    loss = mx.calc_train_loss(data, mx(data))['denoise_loss']
    loss.backward()
    named = dict(mx.named_parameters())
    names = [str(n) for n in A['x0_grad_norms_names']]
    mine = torch.tensor([float(named[n].grad.norm()) for n in names])
    rel = (mine - A['x0_grad_norms']).abs() / (A['x0_grad_norms'].abs() + 1e-12)
    REPORT['x0_train_loss_err'] = abs(float(loss.detach()) - float(A['x0_train_loss']))
    REPORT['x0_grad_norm_max_rel'] = float(rel.max())
    _dump()
    assert REPORT['ancestral_eps_step_agree'] >= 0.995 and REPORT['ancestral_x0_step_agree'] >= 0.995
    assert REPORT['x0_dpm_step0_maxerr'] <= 1e-4 and REPORT['x0_dpm_final_frac_gt_1e-3'] <= 0.02
    assert REPORT['x0_train_loss_err'] <= 1e-5 * max(1.0, float(A['x0_train_loss']))
    assert REPORT['x0_grad_norm_max_rel'] <= 2e-3


def test_v_prediction_target_fp32():
    """SURVEY 8(f) row 2, pred_target='v': v loss target (+ gradient norms), ancestral step through
    the v branch, DPM-Solver++ with the 'v' wrapper -- SAViDiffusion against the video reference run
    in tests/golden/vpred_b1t2.npz (only video_based accepts 'v')."""
    from slotdiffusion_amd import ops
    from slotdiffusion_amd.models import SAViDiffusion
    cfg = C.movie_cfg()
    cfg['dec_dict']['diffusion_dict']['pred_target'] = 'v'
    A = C.load_golden('vpred_b1t2.npz')
    m = SAViDiffusion(cfg['resolution'], 2, cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                      cfg['pred_dict'], cfg['loss_dict'], compute_dtype=torch.float32)
    det_fill_(m.state_dict().items(), skip=is_buffer_name)
    m.train_dropout = m.pred_dropout = 0.0
    m = m.cuda().eval()
    m.use_graph = False
    assert m.dm_decoder.pred_target == 'v'
    slots = A['slots'].cuda()
    nz = A['anc_noise'].cuda()
    agree, x_in = 1.0, A['x_T']
    for j, tv in enumerate(int(v) for v in A['anc_t'].tolist()):
        x = ops.nchw_to_nhwc(x_in.cuda(), torch.float32, 4)
        x_dev = next(m._ancestral_steps(x, slots, [tv], noises=[nz]))[0]
        ok = ((ops.nhwc_to_nchw(x_dev, 3).cpu() - A['v_anc_x'][j]).abs() <= 1e-4).float().mean()
        agree = min(agree, float(ok))
        x_in = A['v_anc_x'][j]
    REPORT['v_ancestral_step_agree'] = agree
    x, tr = m.dm_decoder.generate_imgs(cond=slots, batch_size=2, x_T=A['x_T'].cuda(), ret_intermed=True)
    REPORT['v_dpm_step0_maxerr'] = maxerr(tr[0], A['v_dpm_trace'][0])
    REPORT['v_dpm_final_frac_gt_1e-3'] = float(((x.cpu() - A['v_dpm_final']).abs() > 1e-3).float().mean())
    m.train()
    for p in m.parameters():
        p.grad = None
    img = C.make_inputs(2, seed=17)[0].view(1, 2, 3, 128, 128).cuda()
    data = dict(img=img, t=A['t'].long().cuda(), noise=A['noise'].cuda())
    loss = m.calc_train_loss(data, m(data))['denoise_loss']
    loss.backward()
    named = dict(m.named_parameters())
    names = [str(n) for n in A['v_grad_norms_names']]
    mine = torch.tensor([float(named[n].grad.norm()) for n in names])
    big = A['v_grad_norms'] > 1e-6
    rel = ((mine - A['v_grad_norms']).abs() / (A['v_grad_norms'].abs() + 1e-12))[big]
    REPORT['v_train_loss_err'] = abs(float(loss.detach()) - float(A['v_train_loss']))
    REPORT['v_grad_norm_max_rel'] = float(rel.max())
    _dump()
    assert REPORT['v_ancestral_step_agree'] >= 0.995
    assert REPORT['v_dpm_step0_maxerr'] <= 1e-4 and REPORT['v_dpm_final_frac_gt_1e-3'] <= 0.02
    assert REPORT['v_train_loss_err'] <= 1e-5 * max(1.0, float(A['v_train_loss']))
    assert REPORT['v_grad_norm_max_rel'] <= 2e-2


def test_savi_video_baseline_fp32():
    """Registry model 'SAVi' (per-frame recurrence + spatial-broadcast decoder on every frame):
    forward, loss, gradients and calc_eval_loss metrics against the reference
    (tests/golden/savi_b1t3.npz)."""
    from slotdiffusion_amd.models import SAVi, build_model
    cfg = C.savi_cfg()
    G = C.load_golden('savi_b1t3.npz')

    class P:
        model = 'SAVi'
        resolution, input_frames = cfg['resolution'], cfg['clip_len']
        slot_dict, enc_dict, dec_dict = cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict']
        pred_dict, loss_dict = cfg['pred_dict'], cfg['loss_dict']
    m = build_model(P)
    assert isinstance(m, SAVi) and list(m.state_dict().keys()) == [str(k) for k in G['state_dict_keys']]
    m.set_compute_dtype(torch.float32)
    det_fill_(m.state_dict().items(), skip=is_buffer_name)
    m = m.cuda().train()
    m.pred_dropout = 0.0
    img = C.make_inputs(3, seed=11)[0].view(1, 3, 3, 128, 128).cuda()
    out = m(dict(img=img))
    loss = m.calc_train_loss(dict(img=img), out)['img_recon_loss']
    loss.backward()
    REPORT['savi_slots_maxerr'] = maxerr(out['slots'].detach(), G['slots'])
    REPORT['savi_recon_maxerr'] = maxerr(out['recon_img'].detach()[:, :, :, 1::2, ::2], G['recon_img_sub2'])
    REPORT['savi_masks_maxerr'] = maxerr(out['masks'][:, :, :, 0, ::4, 1::4], G['masks_sub4'])
    am = out['masks'][:, :, :, 0].argmax(2).cpu()
    REPORT['savi_argmax_agree'] = float((am == G['masks_argmax'].long()).float().mean())
    REPORT['savi_loss_err'] = abs(float(loss.detach()) - float(G['img_recon_loss']))
    named = dict(m.named_parameters())
    names = [str(n) for n in G['grad_norms_names']]
    mine = torch.tensor([float(named[n].grad.norm()) for n in names])
    rel = (mine - G['grad_norms']).abs() / (G['grad_norms'].abs() + 1e-7)
    REPORT['savi_grad_norm_max_rel'] = float(rel.max())
    m.eval()
    with torch.no_grad():
        oe = m(dict(img=img))
        ev = m.calc_eval_loss(dict(img=img, masks=G['gt_masks'].long().cuda()), oe)
    REPORT['savi_eval'] = {k: float(v) for k, v in ev.items()}
    _dump()
    assert REPORT['savi_slots_maxerr'] <= 5e-5 and REPORT['savi_recon_maxerr'] <= 5e-5
    assert REPORT['savi_masks_maxerr'] <= 5e-5 and REPORT['savi_argmax_agree'] >= 0.9999
    assert REPORT['savi_loss_err'] <= 1e-6 and REPORT['savi_grad_norm_max_rel'] <= 5e-3
    for k in ('ari', 'fari', 'miou', 'fmiou', 'mbo'):
        assert abs(float(ev[k]) - float(G['eval_' + k])) <= 2e-4, (k, float(ev[k]), float(G['eval_' + k]))
    assert abs(float(ev['img_recon_loss']) - float(G['eval_img_recon_loss'])) <= 1e-6


def test_vqvae_standalone_eval_fp32():
    """Registry model 'VQVAE' (encode / quantize / decode + calc_eval_loss) against the reference
    (tests/golden/vqvae_b2.npz); training mode refuses to run (no silent partial path)."""
    from slotdiffusion_amd.models import VQVAE, build_model
    cfg = C.clevrtex_cfg()
    va = cfg['dec_dict']['vae_dict']
    V = C.load_golden('vqvae_b2.npz')

    class P:
        model = 'VQVAE'
        enc_dec_dict, vq_dict = va['enc_dec_dict'], dict(va['vq_dict'], percept_loss_w=0.)
    m = build_model(P)
    assert isinstance(m, VQVAE) and list(m.state_dict().keys()) == [str(k) for k in V['state_dict_keys']]
    m.set_compute_dtype(torch.float32)
    det_fill_(m.state_dict().items(), skip=is_buffer_name)
    m = m.cuda().eval()
    img = C.make_inputs(2)[0].cuda()
    out = m(dict(img=img))
    ev = m.calc_eval_loss(dict(img=img), out)
    REPORT['vqvae_token_agree'] = float((out['token_id'].cpu() == V['token_id'].long()).float().mean())
    REPORT['vqvae_recon_maxerr'] = maxerr(out['recon'][:, :, 1::2, ::2], V['recon_sub2'])
    REPORT['vqvae_prevq_maxerr'] = maxerr(m.encode(img), V['pre_vq'])
    REPORT['vqvae_losses'] = {k: float(v) for k, v in ev.items()}
    _dump()
    assert REPORT['vqvae_token_agree'] == 1.0 and REPORT['vqvae_recon_maxerr'] <= 5e-5
    assert REPORT['vqvae_prevq_maxerr'] <= 1e-5
    for k in ('quant_loss', 'recon_loss', 'recon_mse'):
        assert abs(float(ev[k]) - float(V[k])) <= 2e-6 * max(1.0, abs(float(V[k]))), (k, float(ev[k]))
    m.percept_loss_w = 1.0                         # the configured loss takes the L1 form
    assert abs(float(m.calc_train_loss(dict(img=img), out)['recon_loss']) - float(V['recon_l1'])) <= 2e-6
    assert maxerr(m.quantize_decode(m.encode(img)), out['recon'].cpu()) <= 1e-6
    clip = img.view(1, 2, 3, 128, 128)             # temporal wrapper
    assert m.encode(clip).shape == (1, 2, 3, 32, 32)


def test_vqvae_stage1_training_gradients():
    """VQ-VAE stage-1 training step on the GPU (decoder, straight-through
    quantizer + commitment loss, encoder, single-head AttnBlocks): parameter gradients against the
    reference's (tests/golden/vqvae_b2.npz) in fp32 (MSE and L1 reconstruction loss); bf16 close;
    one fused Adam step runs."""
    from slotdiffusion_amd.models import VQVAE
    from slotdiffusion_amd.optim import FusedAdam
    cfg = C.clevrtex_cfg()
    va = cfg['dec_dict']['vae_dict']
    V = C.load_golden('vqvae_b2.npz')
    names = [str(n) for n in V['grad_norms_names']]
    img = C.make_inputs(2)[0].cuda()

    def grads(dtype, w):
        m = VQVAE(va['enc_dec_dict'], dict(va['vq_dict'], percept_loss_w=w), compute_dtype=dtype)
        det_fill_(m.state_dict().items(), skip=is_buffer_name)
        m = m.cuda().train()
        m.grad_arena().zero_()
        out = m(dict(img=img))
        ld = m.calc_train_loss(dict(img=img), out)
        (ld['recon_loss'] + ld['quant_loss']).backward()
        named = dict(m.named_parameters())
        return m, ld, named, torch.tensor([float(named[n].grad.double().norm()) for n in names])

    m, ld, named, gn = grads(torch.float32, 0.)
    # (attn_1.k.bias has a mathematically zero gradient -- softmax is shift invariant per query --
    #  so the comparison carries an absolute floor of 1e-6 of the largest norm)
    floor = 1e-6 * float(V['grad_norms'].max())
    rel = (gn - V['grad_norms']).abs() / (V['grad_norms'].abs() + floor / 5e-3)
    REPORT['vqvae_train_grad_norm_max_rel'] = float(rel.max())
    REPORT['vqvae_train_loss_err'] = abs(float(ld['recon_loss'].detach()) - float(V['train_recon_loss']))
    worst = 0.0
    for n in ('quantize.embedding.weight', 'encoder.conv_in.weight', 'decoder.mid.attn_1.k.weight',
              'encoder.mid.attn_1.proj_out.bias', 'decoder.conv_out.bias'):
        ref = V['grad/' + n]
        worst = max(worst, float((named[n].grad.cpu() - ref).abs().max()) / (float(ref.abs().max()) + 1e-12))
    REPORT['vqvae_train_grad_max_rel_elem'] = worst
    opt = FusedAdam(m, lr=1e-4, dec_lr=1e-4, clip_grad=1.0)
    before = m.arena().clone()
    opt.step()
    assert float((m.arena() - before).abs().max()) > 0 and bool(torch.isfinite(m.arena()).all())
    _, _, _, gn1 = grads(torch.float32, 1.)
    rel1 = (gn1 - V['grad_norms_l1']).abs() / (V['grad_norms_l1'].abs() + floor / 5e-3)
    REPORT['vqvae_train_l1_grad_norm_max_rel'] = float(rel1.max())
    _, _, _, gnb = grads(torch.bfloat16, 0.)
    relb = (gnb - V['grad_norms']).abs() / (V['grad_norms'].abs() + floor / 5e-3)
    REPORT['vqvae_train_bf16_grad_norm_median_rel'] = float(relb.median())
    _dump()
    assert REPORT['vqvae_train_loss_err'] <= 1e-6 and REPORT['vqvae_train_grad_norm_max_rel'] <= 5e-3
    assert worst <= 5e-3 and REPORT['vqvae_train_l1_grad_norm_max_rel'] <= 5e-3
    assert REPORT['vqvae_train_bf16_grad_norm_median_rel'] <= 0.1


def test_dino_coco_config_fp32():
    """SURVEY 8(f) row 4 / BASELINE config 5: img_based SADiffusion with the frozen DINO ViT-S/8 encoder
    (224 x 224, 785 tokens through the long-sequence GEMM attention, 28 x 28 x 384 features), 7 slots
    of 256, latent 56 x 56 with the UNet's 28 x 28 self-attention (784 keys: ops.attention_long and
    its backward) -- against the reference run in tests/golden/sadiff_dino_b1.npz."""
    from slotdiffusion_amd.models import SADiffusion
    from slotdiffusion_amd import ops, engine
    cfg = C.dino_coco_cfg()
    G = C.load_golden('sadiff_dino_b1.npz')
    m = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                    cfg['loss_dict'], compute_dtype=torch.float32)
    det_fill_(m.state_dict().items(), skip=is_buffer_name)
    m.train_dropout = 0.0
    m = m.cuda()
    m.use_graph = False
    img, noise = C.dino_inputs()
    img = img.cuda()
    with torch.no_grad():
        feat, (gh, gw) = engine.dino_encoder(m.K(), m._to_nhwc(img), m.rplan)
    feat = feat.view(1, gh, gw, -1).permute(0, 3, 1, 2)
    R = {'dino_feat_maxerr': maxerr(feat[:, ::8], G['dino_feat_sub']),
         'dino_feat_scale': float(G['dino_feat_sub'].abs().max())}
    m.train()
    m.grad_arena().zero_()
    out = m(dict(img=img))
    R['slots_maxerr'] = maxerr(out['slots'], G['slots'])
    R['masks_train_argmax_agree'] = float(
        (out['masks'].cpu().argmax(1) == G['masks_train_argmax'].long()).float().mean())
    x0 = m.dm_decoder.vae.encode(img)
    R['x0_maxerr'] = maxerr(x0, G['x0'])
    loss = m.calc_train_loss(dict(img=img, t=G['t'].cuda(), noise=noise.cuda()), out)['denoise_loss']
    loss.backward()
    R['loss'], R['loss_ref'] = float(loss), float(G['train_loss'])
    named = dict(m.named_parameters())
    names = [str(n) for n in G['grad_norms_names']]
    mine = torch.tensor([float(named[n].grad.norm()) for n in names])
    big = G['grad_norms'] > 1e-6 * float(G['grad_norms'].max())
    rel = ((mine - G['grad_norms']).abs() / (G['grad_norms'].abs() + 1e-12))[big]
    R['grad_norm_max_rel'], R['grad_norm_median_rel'] = float(rel.max()), float(rel.median())
    assert all(named[n].grad is None or not named[n].requires_grad for n in named
               if n.startswith('encoder.dino.'))                      # the ViT is frozen
    m.eval()
    with torch.no_grad():
        xt = m._latent_nhwc(C_xt(G, x0, noise).cuda())
        eps = ops.nhwc_to_nchw(m._unet_eps(xt, G['t'].float().cuda(), G['slots'].cuda()), 3)
        R['eps_maxerr'] = maxerr(eps, G['eps_pred'])
        oe = m(dict(img=img))
    R['masks_eval_argmax_agree'] = float(
        (oe['masks'].cpu().argmax(1) == G['masks_eval_argmax'].long()).float().mean())
    REPORT['dino_coco224'] = R
    _dump()
    assert R['dino_feat_maxerr'] <= 2e-4 * max(1.0, R['dino_feat_scale'])
    assert R['slots_maxerr'] <= 2e-4 and R['masks_train_argmax_agree'] == 1.0
    assert R['x0_maxerr'] <= 1e-4 and R['eps_maxerr'] <= 2e-4
    assert abs(R['loss'] - R['loss_ref']) <= 1e-4 * max(1.0, R['loss_ref'])
    assert R['masks_eval_argmax_agree'] > 0.9995
    assert R['grad_norm_max_rel'] <= 2e-2 and R['grad_norm_median_rel'] <= 2e-3


def C_xt(G, x0, noise):
    """x_t = sqrt(acp[t]) x0 + sqrt(1 - acp[t]) noise with the linear-beta DDPM schedule of the config."""
    from slotdiffusion_amd.module import ddpm_schedule
    cfg = C.dino_coco_cfg()
    dd = {k: v for k, v in cfg['dec_dict']['diffusion_dict'].items()
          if k in ('timesteps', 'beta_schedule', 'linear_start', 'linear_end')}
    s = ddpm_schedule(**dd)
    t = int(G['t'][0])
    a = torch.tensor(s['sqrt_alphas_bar'], dtype=torch.float32)[t]
    b = torch.tensor(s['sqrt_one_minus_alphas_bar'], dtype=torch.float32)[t]
    return a * G['x0'] + b * noise


def test_fp8_unet_coco_config_deviation():
    """BASELINE config 5 ("fp8 MFMA UNet"): the COCO-224 DINO model with e4m3fn operands on the
    denoiser's 3x3 convolutions (compute 'fp8' = bf16 storage + fp8 conv operands) against the
    reference run in sadiff_dino_b1.npz, next to the plain bf16 path: deviations reported and
    bounded (the fp8 bound leaves ~1.5x head-room over the value measured on MI355X)."""
    from slotdiffusion_amd.models import SADiffusion
    from slotdiffusion_amd import ops
    cfg = C.dino_coco_cfg()
    G = C.load_golden('sadiff_dino_b1.npz')
    img, noise = C.dino_inputs()
    R = {}
    for mode in ('bf16', 'fp8'):
        m = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                        cfg['loss_dict'], compute_dtype=torch.bfloat16)
        det_fill_(m.state_dict().items(), skip=is_buffer_name)
        m = m.cuda().eval()
        m.set_compute_dtype(mode)
        m.use_graph = False
        assert m.fp8_unet == (mode == 'fp8') and m.compute_dtype == torch.bfloat16
        with torch.no_grad():
            xt = m._latent_nhwc(C_xt(G, None, noise).cuda())
            eps = ops.nhwc_to_nchw(m._unet_eps(xt, G['t'].float().cuda(), G['slots'].cuda()), 3)
        R[mode + '_eps_rel_l2'] = float((eps.cpu() - G['eps_pred']).norm() / G['eps_pred'].norm())
        if mode == 'fp8':
            from slotdiffusion_amd import _lib
            with _lib.KernelTimer() as kt:
                with torch.no_grad():
                    m._unet_eps(xt, G['t'].float().cuda(), G['slots'].cuda())
            torch.cuda.synchronize()
            R['fp8_igemm_launches'] = sum(1 for r in kt.records if r[0] == 'sdmi_igemm' and r[3].get('fp8'))
            R['fp8_quant_launches'] = sum(1 for r in kt.records if r[0] == 'sdmi_quant_fp8')
    REPORT['fp8_coco224'] = R
    _dump()
    # the ResBlock convolutions really ran on fp8 operands, written by the GroupNorms in front of them
    assert R['fp8_igemm_launches'] >= 40 and R['fp8_quant_launches'] <= 8
    assert R['bf16_eps_rel_l2'] < 0.03
    assert R['fp8_eps_rel_l2'] < 0.10



def test_fp8_coco_config_train_step_deviation():
    """`--config coco224 --dtype fp8 --mode train` (BASELINE configs[4]): one train step of the
    COCO-224 DINO model with e4m3fn operands in the forward 3x3 convolutions of the denoiser (weights quantised on
    the device at 448 / amax, kern.WeightBank.w8_dev) against the reference run in sadiff_dino_b1.npz and the bf16
    step: loss within 5 %, eps of the training provider within 10 % rel-L2, gradient direction kept."""
    from slotdiffusion_amd.models import SADiffusion
    from slotdiffusion_amd import ops, _lib
    cfg = C.dino_coco_cfg()
    G = C.load_golden('sadiff_dino_b1.npz')
    img, noise = C.dino_inputs()
    img = img.cuda()
    R, grads = {}, {}
    for mode in ('bf16', 'fp8'):
        m = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                        cfg['loss_dict'], compute_dtype=torch.bfloat16)
        det_fill_(m.state_dict().items(), skip=is_buffer_name)
        m.train_dropout = 0.0
        m = m.cuda()
        m.set_compute_dtype(mode)
        m.use_graph = False
        m.train()
        m.grad_arena().zero_()
        with _lib.KernelTimer() as kt:
            out = m(dict(img=img))
            loss = m.calc_train_loss(dict(img=img, t=G['t'].cuda(), noise=noise.cuda()), out)['denoise_loss']
            loss.backward()
        torch.cuda.synchronize()
        R[mode + '_loss'] = float(loss)
        grads[mode] = m.grad_arena().float().clone()
        n8 = sum(1 for r in kt.records if r[0] == 'sdmi_igemm' and r[3].get('fp8'))
        assert (n8 >= 30) if mode == 'fp8' else (n8 == 0)
        xt = m._latent_nhwc(C_xt(G, None, noise).cuda())
        with torch.enable_grad():
            e = m._unet_eps(xt, G['t'].float().cuda(), G['slots'].cuda().requires_grad_(True), m.KG())
        eps = ops.nhwc_to_nchw(e.detach(), 3).float().cpu()
        m.bank().join()
        R[mode + '_eps_rel_l2'] = float((eps - G['eps_pred']).norm() / G['eps_pred'].norm())
    R['loss_ref'] = float(G['train_loss'])
    R['fp8_grad_cosine_vs_bf16'] = float(F.cosine_similarity(grads['fp8'], grads['bf16'], dim=0))
    REPORT['fp8_coco224_train'] = R
    _dump()
    assert abs(R['fp8_loss'] - R['loss_ref']) / R['loss_ref'] < 0.05
    assert abs(R['fp8_loss'] - R['bf16_loss']) / R['bf16_loss'] < 0.05
    assert R['fp8_eps_rel_l2'] < 0.10 and R['bf16_eps_rel_l2'] < 0.03
    assert R['fp8_grad_cosine_vs_bf16'] > 0.95
