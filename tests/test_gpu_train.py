"""Train-step parity on a real MI355X: gradients of every trainable tensor (HIP backward kernels)
against the reference's autograd gradients captured in tests/golden/sadiff_b2.npz, the fused
clip+Adam step against the oracle's restatement of torch.optim.Adam, and per-kernel backward
checks against torch-CPU autograd."""
import json
import math
import os

import pytest
import torch
import torch.nn.functional as F

from tests import common as C
from tests.detfill import det_fill_, is_buffer_name

pytestmark = pytest.mark.gpu
DEV = 'cuda'
REPORT = {}


def _dump():
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/train_parity_report.json', 'w') as f:
        json.dump(REPORT, f, indent=1)


def _model(dtype):
    from slotdiffusion_amd.models import SADiffusion
    cfg = C.clevrtex_cfg()
    m = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                    cfg['loss_dict'], compute_dtype=dtype)
    det_fill_(m.state_dict().items(), skip=is_buffer_name)
    m.train_dropout = 0.0       # RNG streams cannot match torch's; parity runs without dropout
    return m.cuda()


def _train_backward(m, G, img):
    m.train()
    m.grad_arena().zero_()
    out = m(dict(img=img))
    loss = m.calc_train_loss(dict(img=img, t=G['t'].cuda(), noise=G['noise'].cuda()), out)[
        'denoise_loss']
    loss.backward()
    return loss


def test_train_step_gradients_fp32():
    G = C.load_golden()
    img = C.make_inputs(2)[0].cuda()
    m = _model(torch.float32)
    loss = _train_backward(m, G, img)
    REPORT['train_loss'] = float(loss)
    REPORT['train_loss_ref'] = float(G['train_loss'])
    named = dict(m.named_parameters())
    gn = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in named.values()
                       if p.requires_grad))
    REPORT['grad_global_norm'] = gn
    REPORT['grad_global_norm_ref'] = float(G['grad_global_norm'])
    names = [str(n) for n in G['grad_norms_names']]
    mine = torch.tensor([float(named[n].grad.norm()) for n in names])
    E = C.load_golden('oracle_fp64_grads.npz')          # exact (fp64 oracle) gradients
    big = G['grad_norms'] > 1e-6                          # skip numerically-zero gradients

    def rel_to(ref):
        return ((mine - ref).abs() / (ref.abs() + 1e-12))[big]
    rel_ref, rel_exact = rel_to(G['grad_norms']), rel_to(E['grad_norms'])
    REPORT['grad_norm_max_rel_vs_reference_fp32'] = float(rel_ref.max())
    REPORT['grad_norm_max_rel_vs_exact_fp64'] = float(rel_exact.max())
    REPORT['reference_fp32_vs_exact_fp64_max_rel'] = float(
        ((G['grad_norms'] - E['grad_norms']).abs() / (E['grad_norms'].abs() + 1e-12))[big].max())
    errs, errs_exact = {}, {}
    for k in G:
        if k.startswith('grad:'):
            g = named[k[5:]].grad.float().cpu()
            errs[k[5:]] = float((g - G[k]).abs().max() / (G[k].abs().max() + 1e-30))
            errs_exact[k[5:]] = float((g - E[k]).abs().max() / (E[k].abs().max() + 1e-30))
    REPORT['grad_tensor_rel_err_vs_reference_fp32'] = errs
    REPORT['grad_tensor_rel_err_vs_exact_fp64'] = errs_exact
    _dump()
    assert abs(REPORT['train_loss'] - REPORT['train_loss_ref']) <= 1e-4
    # the reference's own fp32 gradients are 0.1-0.5 % off the exact ones on the slot-encoder
    # side (see tests/gen_oracle_fp64_grads.py); the HIP backward must match the exact gradients
    # tightly and the reference within that conditioning band
    assert REPORT['grad_norm_max_rel_vs_exact_fp64'] <= 2e-4
    assert max(errs_exact.values()) <= 5e-4, errs_exact
    assert REPORT['grad_norm_max_rel_vs_reference_fp32'] <= 1e-2
    assert max(errs.values()) <= 1e-2, errs
    assert abs(gn - REPORT['grad_global_norm_ref']) <= 1e-3 * REPORT['grad_global_norm_ref']


def test_adam_clip_step_matches_oracle():
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd.optim import FusedAdam
    G = C.load_golden()
    img = C.make_inputs(2)[0].cuda()
    m = _model(torch.float32)
    opt = FusedAdam(m, lr=1e-4, dec_lr=2e-4, clip_grad=1.0)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    for it in range(2):
        opt.zero_grad()
        _train_backward(m, G, img)
        named = dict(m.named_parameters())
        if it == 0:
            P = [named[n].detach().float().cpu().clone() for n in names]
            M = [torch.zeros_like(p) for p in P]
            V = [torch.zeros_like(p) for p in P]
        grads = [named[n].grad.detach().float().cpu().clone() for n in names]
        lrs = [2e-4 if 'dm_decoder' in n else 1e-4 for n in names]
        O.clip_and_adam(P, grads, M, V, it + 1, lrs, clip=1.0)
        opt.step()
    named = dict(m.named_parameters())
    worst = 0.
    for n, p in zip(names, P):
        worst = max(worst, float((named[n].detach().cpu() - p).abs().max()))
    REPORT['adam_param_maxdiff_after_2_steps'] = worst
    _dump()
    assert worst <= 2e-6


@pytest.mark.parametrize('g16', [False, True])
@pytest.mark.parametrize('n,off', [(4099, 0), (1003, 1), (70001, 8)])
def test_adam_reads_summed_gradients_scaled(n, off, g16):
    """Data-parallel form of the update (SdmiAdamArgs.gscale / g_dtype): the kernels read the SUM over ranks where the
    all-reduce left it -- fp32 arena or bf16 wire buffer -- and apply 1 / world themselves; equal to the oracle's
    clip + Adam on the averaged gradients."""
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd import _lib
    world = 8
    g = torch.Generator().manual_seed(n + 7)
    P, Gs = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3 * world      # Gs: summed gradients
    if g16:
        Gs = Gs.bfloat16().float()
    M, V = torch.randn(n, generator=g) * 0.1, torch.rand(n, generator=g) * 0.1
    buf = lambda t: torch.cat([torch.zeros(off), t]).cuda()
    p, m, v = buf(P), buf(M), buf(V)
    gr = buf(Gs).bfloat16() if g16 else buf(Gs)
    partial = torch.zeros(1024, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    gd = _lib.BF16 if g16 else _lib.F32
    _lib.call('sdmi_sqsum_partial', st, g=gr[off:].data_ptr(), partial=partial.data_ptr(), n=n, nblk=1024, g_dtype=gd)
    _lib.call('sdmi_adam_clip', st, p=p[off:].data_ptr(), g=gr[off:].data_ptr(), m=m[off:].data_ptr(),
              v=v[off:].data_ptr(), shadow_bf16=0, sq_partial=partial.data_ptr(), nblk=1024, n=n, lr=1e-3,
              beta1=0.9, beta2=0.999, eps=1e-8, clip=1.0, step=3, lr_dev=0, step_dev=0, gscale=1.0 / world,
              g_dtype=gd)
    Pl, Ml, Vl = [P.clone()], [M.clone()], [V.clone()]
    O.clip_and_adam(Pl, [Gs / world], Ml, Vl, 3, [1e-3], clip=1.0)
    assert float((p[off:].cpu() - Pl[0]).abs().max()) <= 2e-6
    assert float((m[off:].cpu() - Ml[0]).abs().max()) <= 2e-6 and float((v[off:].cpu() - Vl[0]).abs().max()) <= 2e-6
    assert float(p[:off].abs().sum()) == 0.


@pytest.mark.parametrize('n,off', [(4099, 0), (1003, 1), (8, 4), (3, 0)])
def test_adam_kernel_tails_and_alignment(n, off):
    """sdmi_adam_clip on ranges whose length is not a multiple of four and whose start is not 16-byte
    aligned (the vector kernel's scalar tail / the scalar kernel) against the oracle's Adam."""
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd import _lib
    g = torch.Generator().manual_seed(n)
    P, Gd = torch.randn(n, generator=g), torch.randn(n, generator=g) * 3
    M, V = torch.randn(n, generator=g) * 0.1, torch.rand(n, generator=g) * 0.1
    buf = lambda t: torch.cat([torch.zeros(off), t]).cuda()
    p, gr, m, v = buf(P), buf(Gd), buf(M), buf(V)
    shadow = torch.zeros(n + off, dtype=torch.bfloat16, device=DEV)
    partial = torch.zeros(1024, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.call('sdmi_sqsum_partial', st, g=gr[off:].data_ptr(), partial=partial.data_ptr(), n=n, nblk=1024)
    _lib.call('sdmi_adam_clip', st, p=p[off:].data_ptr(), g=gr[off:].data_ptr(), m=m[off:].data_ptr(),
              v=v[off:].data_ptr(), shadow_bf16=shadow[off:].data_ptr(), sq_partial=partial.data_ptr(),
              nblk=1024, n=n, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, clip=1.0, step=3, lr_dev=0,
              step_dev=0)
    Pl, Ml, Vl = [P.clone()], [M.clone()], [V.clone()]
    O.clip_and_adam(Pl, [Gd.clone()], Ml, Vl, 3, [1e-3], clip=1.0)
    assert float((p[off:].cpu() - Pl[0]).abs().max()) <= 1e-6
    assert float((m[off:].cpu() - Ml[0]).abs().max()) <= 1e-6
    assert float((v[off:].cpu() - Vl[0]).abs().max()) <= 1e-6
    assert torch.equal(shadow[off:].cpu(), p[off:].cpu().to(torch.bfloat16))
    assert float(p[:off].abs().sum()) == 0.0 and float(shadow[:off].float().abs().sum()) == 0.0


def test_graphed_train_step_matches_eager():
    """The HIP-graph replay of zero-grad + fwd + bwd + clip + Adam must walk exactly the same
    trajectory as the eager step (same kernels, same order): 4 steps, fixed t / noise, dropout
    off; then with dropout on, two replays must draw different masks (device-side seed word)."""
    from slotdiffusion_amd.optim import FusedAdam, GraphedTrainStep
    G = C.load_golden()
    img = C.make_inputs(2)[0].cuda()
    batch = dict(img=img, t=G['t'].cuda(), noise=G['noise'].cuda())
    finals, losses = [], []
    for graphed in (False, True):
        m = _model(torch.float32)
        m.train()
        opt = FusedAdam(m, lr=1e-4, dec_lr=2e-4, clip_grad=1.0, total_steps=50, warmup_pct=0.2)
        ls = []
        if graphed:
            gs = GraphedTrainStep(m, opt, batch)       # (its warm-up passes are rolled back)
            assert opt.step_count == 0 and int(opt.step_dev) == 0
            for i in range(4):
                ls.append(float(gs(batch)))
        else:
            for _ in range(4):
                opt.zero_grad()
                out = m(batch)
                loss = m.calc_train_loss(batch, out)['denoise_loss']
                loss.backward()
                opt.step()
                ls.append(float(loss))
        assert opt.step_count == 4 and int(opt.step_dev) == 4
        finals.append(m.arena().detach().clone())
        losses.append(ls)
    diff = float((finals[0] - finals[1]).abs().max())
    REPORT['graphed_vs_eager_param_maxdiff'] = diff
    REPORT['graphed_losses'] = losses
    _dump()
    assert diff == 0.0
    assert losses[0] == losses[1]
    # dropout on: the in-graph seed word advances, so replays differ
    m = _model(torch.float32)
    m.train_dropout = 0.1
    m.train()
    opt = FusedAdam(m, lr=0.0, dec_lr=0.0, clip_grad=1.0)
    gs = GraphedTrainStep(m, opt, batch)
    l1, l2 = float(gs(batch)), float(gs(batch))
    assert l1 != l2 and abs(l1 - l2) < 0.2 * abs(l1)


def test_eager_steps_draw_new_dropout_masks():
    """Eager training (SDMI_GRAPH=0, Method._eager_step, plain model(batch) loops): every training
    forward starts a new dropout step, so two consecutive steps on the same batch / t / noise see
    different masks; the same step replayed from the same seed word is reproducible."""
    G = C.load_golden()
    img = C.make_inputs(2)[0].cuda()
    batch = dict(img=img, t=G['t'].cuda(), noise=G['noise'].cuda())
    m = _model(torch.float32)
    m.train_dropout = 0.1
    m.train()
    ls = []
    for _ in range(3):
        m.grad_arena().zero_()
        loss = m.calc_train_loss(batch, m(batch))['denoise_loss']
        loss.backward()
        ls.append(float(loss))
    assert int(m.step_seed) == 3
    assert len(set(ls)) == 3, ls
    m.step_seed.fill_(1)                       # rewind: step 2 again
    l2 = float(m.calc_train_loss(batch, m(batch))['denoise_loss'])
    assert l2 == ls[1]
    # ranks and run seeds key the generator
    from slotdiffusion_amd.kern import KernGrad
    os.environ['RANK'] = '1'
    try:
        assert KernGrad(m.bank()).seed != m.KG().seed
    finally:
        del os.environ['RANK']


def test_eval_loss_draws_fresh_timesteps_and_noise():
    """calc_eval_loss (no_grad / eval mode) draws t and noise like the reference does for every batch
    (ldm.py:65-69): consecutive validation calls see different timesteps / noise, before and after
    training steps, and they do not advance the training stream's seed word."""
    img = C.make_inputs(2)[0].cuda()
    m = _model(torch.float32)
    m.eval()
    ldm = m.dm_decoder
    with torch.no_grad():
        d1 = ldm._draw_tn(64, 32, 32, img.device)
        d2 = ldm._draw_tn(64, 32, 32, img.device)
        assert not torch.equal(d1[0], d2[0]) and not torch.equal(d1[4], d2[4])
        assert int(d1[0].min()) >= 0 and int(d1[0].max()) < 1000
        assert torch.equal(d1[2], ldm.sqrt_alphas_bar[d1[0]])
        out = m(dict(img=img))
        l1 = float(m.calc_eval_loss(dict(img=img), out)['denoise_loss'])
        l2 = float(m.calc_eval_loss(dict(img=img), out)['denoise_loss'])
    assert l1 != l2
    assert m.step_seed is None or int(m.step_seed) == 0
    assert int(m.eval_seed) == 4


def test_checkpoint_resume_continues_the_trajectory(tmp_path):
    """Method.save / fit(resume_from): weights, Adam moments, step (bias correction + schedule
    position), iteration and the dropout seed word are restored -- 3 steps + save + 2 steps equals
    a fresh process resumed from the checkpoint for 2 steps, bit for bit (eager steps)."""
    from slotdiffusion_amd import img_based as task
    os.environ['SDMI_GRAPH'] = '0'
    try:
        def make():
            P = C.make_params('SADiffusion')
            P.max_epochs = 2
            model = task.build_model(P)
            det_fill_(model.state_dict().items(), skip=is_buffer_name)
            model = model.cuda()
            model.set_compute_dtype('fp32')
            dm = task.build_dataset(P)
            dm.steps_per_epoch = 5
            return task.build_method(model=model, datamodule=dm, params=P, ckp_path=None,
                                     local_rank=0, use_ddp=False, use_fp16=False)
        a = make()
        a.fit(max_steps=5)
        path = str(tmp_path / 'ckp.pth')
        a.save(path)
        ref_m, ref_v = a.optimizer.m.clone(), a.optimizer.v.clone()
        b = make()
        b.fit(resume_from=path, max_steps=5)       # restores; it == 5 -> returns after restoring
        assert b.it == 5 and b.optimizer.step_count == 5 and int(b.optimizer.step_dev) == 5
        assert torch.equal(b.optimizer.m, ref_m) and torch.equal(b.optimizer.v, ref_v)
        assert torch.equal(b.model.arena(), a.model.arena())
        assert int(b.model.step_seed) == int(a.model.step_seed)
        # both continue for 2 more steps of epoch 1 (the checkpoint carries the device RNG state the
        # t / noise draws continue from; `a` is rewound to it as well)
        rng = torch.load(path, map_location='cpu')['cuda_rng_state']
        for meth in (a, b):
            torch.cuda.set_rng_state(rng)
            for i, batch in enumerate(meth.datamodule.train_loader(1)):
                if i == 2:
                    break
                meth._eager_step(batch)
        assert torch.equal(b.model.arena(), a.model.arena())
        assert a.optimizer.lr_scale(a.optimizer.step_count) == b.optimizer.lr_scale(b.optimizer.step_count)
    finally:
        del os.environ['SDMI_GRAPH']


def test_mid_epoch_resume_skips_consumed_batches(tmp_path):
    """A checkpoint written in the middle of an epoch resumes behind the batches it consumed: 3 steps +
    save + resume for 2 steps equals 5 uninterrupted steps bit for bit, and the run executes exactly
    max_epochs * steps_per_epoch steps (eager steps)."""
    from slotdiffusion_amd import img_based as task
    os.environ['SDMI_GRAPH'] = '0'
    try:
        def make():
            P = C.make_params('SADiffusion')
            P.max_epochs = 1
            model = task.build_model(P)
            det_fill_(model.state_dict().items(), skip=is_buffer_name)
            model = model.cuda()
            model.set_compute_dtype('fp32')
            dm = task.build_dataset(P)
            dm.steps_per_epoch = 5
            return task.build_method(model=model, datamodule=dm, params=P, ckp_path=None,
                                     local_rank=0, use_ddp=False, use_fp16=False)
        a = make()
        a.fit(max_steps=3)
        path = str(tmp_path / 'mid.pth')
        a.save(path)
        b = make()
        b.fit(resume_from=path)                    # runs to the end of the (only) epoch
        c = make()
        c.fit()
        assert b.it == 5 and c.it == 5 and b.optimizer.step_count == 5
        assert [float(x) for x in b.history] == [float(x) for x in c.history[3:]]
        assert torch.equal(b.model.arena(), c.model.arena())
    finally:
        del os.environ['SDMI_GRAPH']


def test_train_step_bf16_gradients_close():
    G = C.load_golden()
    img = C.make_inputs(2)[0].cuda()
    m = _model(torch.bfloat16)
    loss = _train_backward(m, G, img)
    named = dict(m.named_parameters())
    names = [str(n) for n in G['grad_norms_names']]
    mine = torch.tensor([float(named[n].grad.norm()) for n in names])
    ref = G['grad_norms']
    rel = ((mine - ref).abs() / (ref.abs() + 1e-12))
    REPORT['bf16_train_loss'] = float(loss)
    REPORT['bf16_grad_norm_median_rel'] = float(rel.median())
    REPORT['bf16_grad_norm_p95_rel'] = float(rel.kthvalue(int(0.95 * len(rel))).values)
    cos = []
    for k in G:
        if k.startswith('grad:'):
            g = named[k[5:]].grad.float().cpu().flatten()
            r = G[k].flatten()
            cos.append(float(F.cosine_similarity(g, r, dim=0)))
    REPORT['bf16_grad_cosine_min'] = min(cos)
    _dump()
    assert abs(float(loss) - float(G['train_loss'])) < 0.05
    # measured on MI355X (round 5): median 0.6 %, p95 0.9 %, worst per-tensor cosine 0.9888 -- bars at ~2-3x the
    # measurement, so one layer drifting cannot hide behind the others
    assert REPORT['bf16_grad_norm_median_rel'] < 0.02 and REPORT['bf16_grad_norm_p95_rel'] < 0.03 and min(cos) > 0.98


# ---- per-kernel backward checks ----------------------------------------------------------
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', [(2, 32, 64, 16, 3, 1, (1, 1, 1, 1), False),
                                  (2, 64, 48, 16, 3, 2, (1, 1, 1, 1), False),
                                  (2, 32, 64, 8, 3, 1, (1, 1, 1, 1), True),
                                  (2, 64, 128, 16, 1, 2, (0, 0, 0, 0), False),
                                  (3, 128, 3, 16, 3, 1, (1, 1, 1, 1), False),
                                  (2, 96, 192, 8, 1, 1, (0, 0, 0, 0), False),
                                  (2, 64, 64, 32, 5, 1, (2, 2, 2, 2), False),     # 5x5 "same"
                                  (3, 64, 64, 64, 3, 1, (1, 1, 1, 1), False),     # direct 3x3 c64 kernel (splits > 1)
                                  (1, 64, 64, 128, 3, 1, (1, 1, 1, 1), False),    # ... two tiles per image row
                                  (1, 16, 32, 128, 3, 1, (1, 1, 1, 1), False),    # W > m step
                                  (2, 32, 64, 12, 3, 1, (1, 1, 1, 1), False),     # not 2^n
                                  (5, 72, 200, 8, 1, 1, (0, 0, 0, 0), False)])    # 1x1 tails
def test_wgrad_and_dgrad_kernels(case, dtype):
    from slotdiffusion_amd import _lib, ops
    from slotdiffusion_amd.kern import _DT
    B, Cin, Cout, H, k, stride, pad, ups = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    q = lambda t: t.to(dtype).float()
    x = q(torch.randn(B, Cin, H, H, generator=g)).requires_grad_(True)
    w = q(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).requires_grad_(True)
    xin = F.interpolate(x, scale_factor=2, mode='nearest') if ups else x
    y = F.conv2d(F.pad(xin, (pad[2], pad[3], pad[0], pad[1])), w, None, stride=stride)
    dy = q(torch.randn(y.shape, generator=g))
    y.backward(dy)
    vec = ops.vec_of(dtype)
    npad = (Cout + vec - 1) // vec * vec
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    dyd = F.pad(dy.permute(0, 2, 3, 1), (0, npad - Cout)).contiguous().to(dtype).cuda()
    Ho = y.shape[2]
    M, K = B * Ho * Ho, k * k * Cin
    def run(splits, accumulate=0, init=0.0):
        ws = torch.empty(splits * (Cout * K + Cout), device='cuda')
        dw_ = torch.full((Cout, K), init, device='cuda')
        db_ = torch.full((Cout,), init, device='cuda')
        _lib.call('sdmi_wgrad', torch.cuda.current_stream().cuda_stream, a=xd.data_ptr(),
                  dy=dyd.data_ptr(), dw=dw_.data_ptr(), dbias=db_.data_ptr(),
                  workspace=ws.data_ptr(), dtype=_DT[dtype], M=M, N=Cout, K=K, lda=Cin, ldy=npad,
                  B=B, H=H, W=H, Cin=Cin, Ho=Ho, Wo=Ho, KH=k, KW=k, stride=stride, pad_t=pad[0],
                  pad_l=pad[2], ups=int(ups), splits=splits, accumulate=accumulate)
        return dw_, db_

    dw, db = run(4)
    scale = max(1.0, float(dw.abs().max()))
    dw1, db1 = run(1, accumulate=1, init=0.5)      # one launch, accumulating straight into dW
    assert float((dw1 - 0.5 - dw).abs().max()) <= 1e-5 * scale + 1e-5
    assert float((db1 - 0.5 - db).abs().max()) <= 1e-4 * max(1.0, float(db.abs().max()))
    dwa, dba = run(3, accumulate=1, init=-1.0)
    assert float((dwa + 1.0 - dw).abs().max()) <= 1e-5 * scale + 1e-5
    assert float((dba + 1.0 - db).abs().max()) <= 1e-4 * max(1.0, float(db.abs().max()))
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(Cout, K)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    e = float((dw.cpu() - ref_dw).norm() / ref_dw.norm())
    assert e <= tol, f'wgrad rel err {e}'
    eb = float((db.cpu() - dy.sum((0, 2, 3))).norm() / dy.sum((0, 2, 3)).norm())
    assert eb <= tol, f'dbias rel err {eb}'
    # dgrad through the autograd Function used by the engine
    from slotdiffusion_amd.kern import WeightBank  # noqa: F401  (API presence)
    wp = w.detach().permute(0, 2, 3, 1).reshape(Cout, K).contiguous().to(dtype).cuda()
    wd = torch.zeros(Cin * k * k, npad, dtype=dtype, device='cuda')
    _lib.call('sdmi_pack_dgrad', torch.cuda.current_stream().cuda_stream, src=wp.data_ptr(),
              dst=wd.data_ptr(), dtype=_DT[dtype], Cout=Cout, KH=k, KW=k, Cin=Cin, CoutPad=npad)
    Hs = 2 * H if ups else H
    out = torch.empty(B, Hs, Hs, Cin, dtype=dtype, device='cuda')
    _lib.call('sdmi_igemm', torch.cuda.current_stream().cuda_stream, a=dyd.data_ptr(),
              w=wd.data_ptr(), out=out.data_ptr(), dtype=_DT[dtype], out_dtype=_DT[dtype],
              M=B * Hs * Hs, N=Cin, K=k * k * npad, lda=npad, ldw=k * k * npad, ldc=Cin, B=B, H=Ho,
              W=Ho, Cin=npad, Ho=Hs, Wo=Hs, KH=k, KW=k, stride=1, pad_t=k - 1 - pad[0],
              pad_l=k - 1 - pad[2], act=0, alpha=1.0, split_k=1, batch=1,
              zins=(stride if stride > 1 else 0))
    if ups:
        dx = torch.empty(B, H, H, Cin, dtype=dtype, device='cuda')
        _lib.call('sdmi_pool2x2_sum', torch.cuda.current_stream().cuda_stream, x=out.data_ptr(),
                  y=dx.data_ptr(), dtype=_DT[dtype], B=B, H=H, W=H, C=Cin)
    else:
        dx = out
    ref_dx = x.grad.permute(0, 2, 3, 1)
    e = float((dx.float().cpu() - ref_dx).norm() / ref_dx.norm())
    assert e <= tol, f'dgrad rel err {e}'



@pytest.mark.parametrize('case', [(8, 128, 128, 32, 32),      # 4 channel pairs x 32 slots, one tile per slot
                                  (20, 128, 192, 32, 22),     # 6 pairs, ragged tiles per slot (80 tiles over 22 slots)
                                  (64, 128, 64, 16, 64),      # W = 16: a tile is a whole image
                                  (4, 64, 128, 64, 64)])      # W = 64: four image rows per tile
def test_wgrad3x3_halo_kernel(case):
    """Direct 3x3 weight gradient on (64 output x 64 input channel) pairs (csrc/wgrad.hip: wgrad3x3_halo_kernel) against
    torch's convolution backward; the implicit-GEMM kernel (SDMI_WGRAD_HALO=0 is a process-level switch, so: a launch
    whose split count keeps it off the direct kernel) must agree to fp32 accumulation-order noise."""
    from slotdiffusion_amd import _lib
    from slotdiffusion_amd.kern import _DT
    B, Cin, Cout, H, splits = case
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(sum(case))
    q = lambda t: t.to(dtype).float()
    x = q(torch.randn(B, Cin, H, H, generator=g)).requires_grad_(True)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    y = F.conv2d(x, w, None, padding=1)
    dy = q(torch.randn(y.shape, generator=g))
    y.backward(dy)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    M, K = B * H * H, 9 * Cin

    def run(splits, accumulate=0, init=0.0):
        ws = torch.empty(splits * (Cout * K + Cout), device='cuda')
        dw_ = torch.full((Cout, K), init, device='cuda')
        db_ = torch.full((Cout,), init, device='cuda')
        _lib.call('sdmi_wgrad', torch.cuda.current_stream().cuda_stream, a=xd.data_ptr(), dy=dyd.data_ptr(),
                  dw=dw_.data_ptr(), dbias=db_.data_ptr(), workspace=ws.data_ptr(), dtype=_DT[dtype], M=M, N=Cout, K=K,
                  lda=Cin, ldy=Cout, B=B, H=H, W=H, Cin=Cin, Ho=H, Wo=H, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, ups=0,
                  splits=splits, accumulate=accumulate)
        return dw_, db_

    assert (Cout // 64) * (Cin // 64) * splits >= 128 and M // 256 >= splits      # the direct kernel's dispatch condition
    dw, db = run(splits)
    ref_dw = w.grad.permute(0, 2, 3, 1).reshape(Cout, K)
    e = float((dw.cpu() - ref_dw).norm() / ref_dw.norm())
    assert e <= 2e-3, f'wgrad rel err {e}'        # (bf16 operands, fp32 accumulation: the products are exact)
    ref_db = dy.sum((0, 2, 3))
    assert float((db.cpu() - ref_db).norm() / ref_db.norm()) <= 1e-4
    dwg, dbg = run(2)                               # implicit-GEMM kernel (2 slots: below the direct kernel's grid)
    assert float((dw - dwg).norm() / dwg.norm()) <= 1e-5 and float((db - dbg).norm() / dbg.norm()) <= 1e-5
    dwa, dba = run(splits, accumulate=1, init=-1.0)
    assert float((dwa + 1.0 - dw).abs().max()) <= 1e-5 * max(1.0, float(dw.abs().max())) + 1e-5
    assert torch.equal(run(splits)[0], dw)          # deterministic fold


def test_wgrad3x3_one_split_writes_the_gradient():
    """splits = 1 means 'written straight into dw' at the C ABI.  The direct channel-pair kernel always leaves M-split
    partials for a fold, so a one-split problem with >= 128 channel pairs (512 -> 1024 channels: 128 pairs) must NOT
    reach it (weight and bias gradients were silently lost on exactly this shape)."""
    from slotdiffusion_amd import _lib
    from slotdiffusion_amd.kern import _DT
    B, Cin, Cout, H = 1, 512, 1024, 16
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, H, H, Cin, generator=g).to(dtype)
    dy = torch.randn(B, H, H, Cout, generator=g).to(dtype)
    xd, dyd = x.cuda(), dy.cuda()
    M, K = B * H * H, 9 * Cin
    assert (Cout // 64) * (Cin // 64) >= 128 and M // 256 >= 1
    dw = torch.full((Cout, K), 7.0, device='cuda')
    db = torch.full((Cout,), 7.0, device='cuda')
    ws = torch.empty(Cout * K + Cout, device='cuda')          # (one slot: the entry point wants a workspace pointer)
    _lib.call('sdmi_wgrad', torch.cuda.current_stream().cuda_stream, a=xd.data_ptr(), dy=dyd.data_ptr(),
              dw=dw.data_ptr(), dbias=db.data_ptr(), workspace=ws.data_ptr(), dtype=_DT[dtype], M=M, N=Cout, K=K, lda=Cin, ldy=Cout,
              B=B, H=H, W=H, Cin=Cin, Ho=H, Wo=H, KH=3, KW=3, stride=1, pad_t=1, pad_l=1, ups=0, splits=1, accumulate=0)
    xp = F.pad(x.float().permute(0, 3, 1, 2), (1, 1, 1, 1))
    cols = F.unfold(xp, 3).view(B, Cin, 9, H * H)                          # [B, Cin, tap, pixel]
    ref = torch.einsum('bpn,bctp->ntc', dy.float().view(B, H * H, Cout), cols).reshape(Cout, K)
    assert float((dw.cpu() - ref).norm() / ref.norm()) <= 2e-3
    ref_db = dy.float().sum((0, 1, 2))
    assert float((db.cpu() - ref_db).norm() / ref_db.norm()) <= 1e-4


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('rows,C', [(100, 192), (1000, 256), (333, 384), (64, 512), (50, 1024),
                                    (7, 64)])
def test_layernorm_bwd_kernel(rows, C, dtype):
    from slotdiffusion_amd import _lib, ops
    from slotdiffusion_amd.kern import _DT
    g = torch.Generator().manual_seed(rows + C)
    q = lambda t: t.to(dtype).float()
    x = q(torch.randn(rows, C, generator=g) * 2 + 0.5).requires_grad_(True)
    gam = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_(True)
    bet = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    dy = q(torch.randn(rows, C, generator=g))
    F.layer_norm(x, (C,), gam, bet, 1e-5).backward(dy)
    xd, dyd = x.detach().to(dtype).cuda(), dy.to(dtype).cuda()
    stats = torch.empty(rows, 2, device='cuda')
    ops.layer_norm(xd, gam.detach().cuda(), bet.detach().cuda(), stats=stats)
    nblk = max(1, min(512, rows // 16))
    partial = torch.empty(nblk * C * 2, device='cuda')
    dx = torch.empty_like(xd)
    dgam, dbet = torch.full((C,), 0.5, device='cuda'), torch.full((C,), -0.5, device='cuda')
    _lib.call('sdmi_layernorm_bwd', torch.cuda.current_stream().cuda_stream, x=xd.data_ptr(),
              dy=dyd.data_ptr(), dx=dx.data_ptr(), gamma=gam.detach().cuda().data_ptr(),
              stats=stats.data_ptr(), dgamma=dgam.data_ptr(), dbeta=dbet.data_ptr(),
              partial=partial.data_ptr(), dtype=_DT[dtype], rows=rows, C=C, nblk=nblk,
              accumulate=1)
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    assert rel(dx, x.grad) <= tol
    assert rel(dgam - 0.5, gam.grad) <= tol
    assert rel(dbet + 0.5, bet.grad) <= tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,HW,C,act,res', [(2, 256, 64, 'silu', False), (3, 64, 384, 'silu', False),
                                            (2, 1024, 128, 'relu', True), (2, 16, 1024, None, False),
                                            (5, 100, 96, 'silu', True)])
def test_groupnorm_bwd_kernel(B, HW, C, act, res, dtype):
    from slotdiffusion_amd import _lib, ops
    from slotdiffusion_amd.kern import _DT
    g = torch.Generator().manual_seed(B * HW + C)
    q = lambda t: t.to(dtype).float()
    x = q(torch.randn(B, HW, C, generator=g) * 1.5 + 0.3).requires_grad_(True)
    r = q(torch.randn(B, HW, C, generator=g)).requires_grad_(True) if res else None
    gam = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_(True)
    bet = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    dy = q(torch.randn(B, HW, C, generator=g))
    z = F.group_norm(x.permute(0, 2, 1), 32, gam, bet, 1e-5).permute(0, 2, 1)
    if res:
        z = z + r
    y = {'silu': F.silu, 'relu': F.relu, None: (lambda t: t)}[act](z)
    y.backward(dy)
    xd, dyd = x.detach().to(dtype).cuda(), dy.to(dtype).cuda()
    rd = r.detach().to(dtype).cuda() if res else None
    gd, bd = gam.detach().cuda(), bet.detach().cuda()
    _, stats = ops.group_norm(xd, gd, bd, eps=1e-5, act=act, residual=rd, return_stats=True)
    nsplit = max(1, min(16, HW // 64))
    partial = torch.empty(B * nsplit * C * 2 + B * 32 * 2, device='cuda')
    dx = torch.empty_like(xd)
    dres = torch.empty_like(xd) if res else None
    dgam, dbet = torch.full((C,), 0.25, device='cuda'), torch.full((C,), -0.25, device='cuda')
    _lib.call('sdmi_groupnorm_bwd', torch.cuda.current_stream().cuda_stream, x=xd.data_ptr(),
              dy=dyd.data_ptr(), dx=dx.data_ptr(), gamma=gd.data_ptr(), beta=bd.data_ptr(),
              stats=stats.data_ptr(), dgamma=dgam.data_ptr(), dbeta=dbet.data_ptr(),
              partial=partial.data_ptr(), dtype=_DT[dtype], B=B, HW=HW, C=C, groups=32,
              act=_lib.ACT[act], nsplit=nsplit, residual=(rd.data_ptr() if res else 0),
              dresidual=(dres.data_ptr() if res else 0), accumulate=1)
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    assert rel(dx, x.grad) <= tol
    assert rel(dgam - 0.25, gam.grad) <= tol
    assert rel(dbet + 0.25, bet.grad) <= tol
    if res:
        assert rel(dres, r.grad) <= tol


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,HW,C', [(2, 256, 128), (2, 1024, 128), (3, 64, 384), (2, 8192, 64)])
def test_groupnorm_fanout_and_fused_dropout(B, HW, C, dtype):
    """The training-mode GroupNorm Function with its fused extras against torch autograd:
    (a) alias outputs -- the gradients of x's other consumers come back into the backward and are
        summed inside the kernel (dextra0 / dextra1) -- single-pass and two-pass geometries;
    (b) dropout fused behind the activation: the keep mask regenerated in backward equals the
        forward's, keep rate = 1 - p, survivors are scaled by 1 / (1 - p)."""
    from slotdiffusion_amd.kern import GroupNormFn
    g = torch.Generator().manual_seed(B * HW + C)
    q = lambda t: t.to(dtype).float()
    x0 = q(torch.randn(B, HW, C, generator=g) * 1.5 + 0.3)
    gam, bet = 1 + 0.3 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    dy, d1, d2 = (q(torch.randn(B, HW, C, generator=g)) for _ in range(3))

    class WB:                                   # the slice of WeightBank GroupNormFn touches
        class model:
            @staticmethod
            def grad_arena():
                return WB.g
            _offsets = {'n.weight': (0, C), 'n.bias': (C, C)}
        t = {'n.weight': gam.cuda(), 'n.bias': bet.cuda()}
        g = torch.zeros(2 * C, device='cuda')
        defer_colsum = False

        @staticmethod
        def f(k):
            return WB.t[k]
    # (a) fan-out: y, a1, a2 = GN(x);  loss = <y, dy> + <a1, d1> + <a2, d2>
    xr = x0.clone().requires_grad_(True)
    yr = F.silu(F.group_norm(xr.permute(0, 2, 1), 32, gam, bet, 1e-5).permute(0, 2, 1))
    (yr * dy).sum().backward()
    ref_dx = xr.grad + d1 + d2
    xd = x0.to(dtype).cuda().requires_grad_(True)
    anchor = torch.zeros(1, device='cuda', requires_grad=True)
    y, a1, a2 = GroupNormFn.apply(xd, None, anchor, WB, 'n', 1e-5, 'silu', 2, None)
    torch.autograd.backward([y, a1, a2], [dy.to(dtype).cuda(), d1.to(dtype).cuda(), d2.to(dtype).cuda()])
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    assert rel(xd.grad, ref_dx) <= tol
    assert a1.data_ptr() == xd.data_ptr()
    # (b) fused dropout
    p = 0.1
    seed_dev = torch.full((1,), 5, dtype=torch.int64, device='cuda')
    xd2 = x0.to(dtype).cuda().requires_grad_(True)
    yd = GroupNormFn.apply(xd2, None, anchor, WB, 'n', 1e-5, 'silu', 0, (p, 1234, seed_dev))
    y_plain = y.detach().float().cpu()
    ydc = yd.detach().float().cpu()
    keep = ydc != 0
    frac = 1.0 - float(keep.float().mean()) - float((y_plain == 0).float().mean())
    assert abs(frac - p) < 0.01, frac
    assert rel(ydc[keep], (y_plain / (1 - p))[keep]) <= (1e-6 if dtype == torch.float32 else 1e-2)
    yd.backward(dy.to(dtype).cuda())
    xr2 = x0.clone().requires_grad_(True)
    yr2 = F.silu(F.group_norm(xr2.permute(0, 2, 1), 32, gam, bet, 1e-5).permute(0, 2, 1))
    (yr2 * (dy * keep.float() / (1 - p))).sum().backward()
    assert rel(xd2.grad, xr2.grad) <= tol
    # another seed word -> another mask
    seed_dev.fill_(6)
    yd2 = GroupNormFn.apply(xd2.detach(), None, anchor, WB, 'n', 1e-5, 'silu', 0, (p, 1234, seed_dev))
    assert float(((yd2 != 0) != (yd != 0)).float().mean()) > 0.1


def test_wgrad_group_matches_single_launches():
    """sdmi_wgrad_group: several 1x1 / linear bf16 problems (different shapes, ragged M, with and
    without bias, with and without M-splits) in one launch equal the single-problem launches bit for
    bit and torch's fp32 result within bf16 tolerance; accumulate adds to the existing gradient."""
    import ctypes
    from slotdiffusion_amd import _lib
    g = torch.Generator().manual_seed(11)
    shapes = [(1024, 512, 512, True, 1), (1000, 1536, 512, False, 1), (4096, 384, 1536, True, 4),
              (448, 256, 192, True, 1), (16384, 256, 256, True, 8), (777, 128, 136, False, 2)]
    st = torch.cuda.current_stream().cuda_stream
    probs, refs, keep = [], [], []
    for M, N, K, bias, splits in shapes:
        x = (torch.randn(M, K, generator=g)).bfloat16().cuda()
        dy = (torch.randn(M, N, generator=g) / 8).bfloat16().cuda()
        init = torch.randn(N, K, generator=g).cuda()
        binit = torch.randn(N, generator=g).cuda()
        ws = torch.empty(splits * (N * K + N), device='cuda')
        kw = dict(a=x.data_ptr(), dy=dy.data_ptr(), dtype=_lib.BF16, M=M, N=N, K=K, lda=K, ldy=N, B=M, H=1,
                  W=1, Cin=K, Ho=1, Wo=1, KH=1, KW=1, stride=1, pad_t=0, pad_l=0, ups=0, splits=splits,
                  accumulate=1, workspace=ws.data_ptr())
        outs = []
        for _ in range(2):                                 # [0]: single launches, [1]: grouped
            dw, db = init.clone(), binit.clone()
            outs.append((dw, db))
        _lib.call('sdmi_wgrad', st, dw=outs[0][0].data_ptr(), dbias=(outs[0][1].data_ptr() if bias else 0), **kw)
        probs.append(dict(kw, dw=outs[1][0].data_ptr(), dbias=(outs[1][1].data_ptr() if bias else 0)))
        refs.append((init.cpu() + dy.float().cpu().t() @ x.float().cpu(),
                     binit.cpu() + dy.float().cpu().sum(0), outs, bias))
        keep += [x, dy, ws]
    arr = (_lib.CSTRUCT['SdmiWgradArgs'] * len(probs))()
    for a, kw in zip(arr, probs):
        for k, v in kw.items():
            setattr(a, k, v)
    _lib.call('sdmi_wgrad_group', st, problems=ctypes.addressof(arr), n=len(probs))
    torch.cuda.synchronize()
    for (rw, rb, outs, bias), sh in zip(refs, shapes):
        assert torch.equal(outs[0][0], outs[1][0]), sh
        assert float((outs[1][0].cpu() - rw).norm() / rw.norm()) < 2e-3, sh
        if bias:
            assert torch.equal(outs[0][1], outs[1][1]), sh
            assert float((outs[1][1].cpu() - rb).norm() / rb.norm()) < 2e-3, sh
    # argument validation: fp32 / 3x3 / narrow problems are refused without a launch
    bad = (_lib.CSTRUCT['SdmiWgradArgs'] * 1)()
    for k, v in dict(probs[0], N=32).items():
        setattr(bad[0], k, v)
    with pytest.raises(_lib.SdmiError):
        _lib.call('sdmi_wgrad_group', st, problems=ctypes.addressof(bad), n=1)


def test_wgrad_group_fp32_matches_single_launches():
    """sdmi_wgrad_group with exact-fp32 problems (the Slot Attention / predictor layers: M = images x slots rows, any
    N / K, with and without bias, one with M-splits): one launch equals the single-problem launches bit for bit and
    torch's fp32 result to 1e-5; mixing dtypes in a group is refused."""
    import ctypes
    from slotdiffusion_amd import _lib
    g = torch.Generator().manual_seed(12)
    shapes = [(448, 192, 192, True, 1), (448, 576, 192, True, 1), (448, 192, 384, False, 1), (448, 384, 192, True, 1),
              (240, 768, 192, True, 1), (100, 64, 72, False, 1), (1000, 200, 136, True, 2), (448, 1024, 192, True, 1)]
    st = torch.cuda.current_stream().cuda_stream
    probs, refs, keep = [], [], []
    for M, N, K, bias, splits in shapes:
        x = torch.randn(M, K, generator=g).cuda()
        dy = (torch.randn(M, N, generator=g) / 8).cuda()
        init = torch.randn(N, K, generator=g).cuda()
        binit = torch.randn(N, generator=g).cuda()
        ws = torch.empty(splits * (N * K + N), device='cuda')
        kw = dict(a=x.data_ptr(), dy=dy.data_ptr(), dtype=_lib.F32, M=M, N=N, K=K, lda=K, ldy=N, B=M, H=1,
                  W=1, Cin=K, Ho=1, Wo=1, KH=1, KW=1, stride=1, pad_t=0, pad_l=0, ups=0, splits=splits,
                  accumulate=1, workspace=ws.data_ptr())
        outs = [(init.clone(), binit.clone()) for _ in range(2)]           # [0]: single launches, [1]: grouped
        _lib.call('sdmi_wgrad', st, dw=outs[0][0].data_ptr(), dbias=(outs[0][1].data_ptr() if bias else 0), **kw)
        probs.append(dict(kw, dw=outs[1][0].data_ptr(), dbias=(outs[1][1].data_ptr() if bias else 0)))
        refs.append((init.cpu() + dy.cpu().t() @ x.cpu(), binit.cpu() + dy.cpu().sum(0), outs, bias))
        keep += [x, dy, ws]
    arr = (_lib.CSTRUCT['SdmiWgradArgs'] * len(probs))()
    for a, kw in zip(arr, probs):
        for k, v in kw.items():
            setattr(a, k, v)
    _lib.call('sdmi_wgrad_group', st, problems=ctypes.addressof(arr), n=len(probs))
    torch.cuda.synchronize()
    for (rw, rb, outs, bias), sh in zip(refs, shapes):
        assert torch.equal(outs[0][0], outs[1][0]), sh
        assert float((outs[1][0].cpu() - rw).norm() / rw.norm()) < 1e-5, sh
        if bias:
            assert torch.equal(outs[0][1], outs[1][1]), sh
            assert float((outs[1][1].cpu() - rb).norm() / rb.norm()) < 1e-5, sh
    mixed = (_lib.CSTRUCT['SdmiWgradArgs'] * 2)()
    for a, kw in zip(mixed, [probs[0], dict(probs[1], dtype=_lib.BF16)]):
        for k, v in kw.items():
            setattr(a, k, v)
    with pytest.raises(_lib.SdmiError):
        _lib.call('sdmi_wgrad_group', st, problems=ctypes.addressof(mixed), n=2)


@pytest.mark.parametrize('case', [
    # (B, H, Cin, Cout, k, bias, residual, splits)   H = 0: linear with B rows
    (1024, 0, 512, 512, 1, True, False, 1), (1000, 0, 384, 1536, 1, False, True, 3),
    (16384, 0, 256, 256, 1, True, True, 16), (448, 0, 192, 576, 1, True, False, 1),
    (4, 16, 128, 256, 3, True, True, 2), (2, 32, 128, 128, 3, True, False, 4),
    (8, 8, 384, 384, 3, False, True, 1), (3, 16, 256, 192, 1, True, False, 2),
    (64, 4, 512, 512, 3, True, False, 1),
    # the geometry B = 64 selects in the timed step: 128 x 128 data-gradient tiles, MODE 2 (3x3), 128-wide dW tiles
    (64, 16, 256, 256, 3, True, True, 4), (32, 32, 128, 128, 3, True, False, 8),
    (3, 28, 128, 192, 3, True, True, 3), (5, 14, 256, 256, 3, False, False, 2)])     # 28^2 / 14^2: not powers of two
def test_bwd_pair_matches_separate_launches(case):
    """sdmi_bwd_pair: data gradient + weight gradient (+ the fold of an earlier layer's M-split partials)
    in ONE launch equal sdmi_igemm + sdmi_wgrad (+ its fold) bit for bit, and torch's fp32 gradients
    within bf16 tolerance.  Linear and 3x3 / 1x1 convolutions, ragged M, bias gradient, residual in the
    data gradient's epilogue, splits 1 ... 16, chained folds, accumulation into existing gradients."""
    import ctypes
    from slotdiffusion_amd import _lib
    B, H, Cin, Cout, k, bias, res, splits = case
    g = torch.Generator().manual_seed(sum(case[:5]) + splits)
    st = torch.cuda.current_stream().cuda_stream
    conv = H > 0
    M = B * H * H if conv else B
    K = k * k * Cin
    pad = (k - 1) // 2
    xf = torch.randn(M, Cin, generator=g).bfloat16()
    wf = (torch.randn(Cout, K, generator=g) / math.sqrt(K)).bfloat16()           # [Cout][kh][kw][Cin]
    dyf = (torch.randn(M, Cout, generator=g) / 8).bfloat16()
    rf = torch.randn(M, Cin, generator=g).bfloat16() if res else None
    x, w, dy = xf.cuda(), wf.cuda(), dyf.cuda()
    r = rf.cuda() if res else None
    wd = torch.zeros(Cin * k * k, Cout, dtype=torch.bfloat16, device='cuda')
    _lib.call('sdmi_pack_dgrad', st, src=w.data_ptr(), dst=wd.data_ptr(), dtype=_lib.BF16, Cout=Cout, KH=k, KW=k,
              Cin=Cin, CoutPad=Cout)
    Hh = H if conv else 1
    Bb = B if conv else M
    init = torch.randn(Cout, K, generator=g).cuda()
    binit = torch.randn(Cout, generator=g).cuda()

    def dkw(out):
        return dict(a=dy.data_ptr(), w=wd.data_ptr(), out=out.data_ptr(), dtype=_lib.BF16, out_dtype=_lib.BF16,
                    M=M, N=Cin, K=k * k * Cout, lda=Cout, ldw=k * k * Cout, ldc=Cin, B=Bb, H=Hh, W=Hh, Cin=Cout,
                    Ho=Hh, Wo=Hh, KH=k, KW=k, stride=1, pad_t=k - 1 - pad, pad_l=k - 1 - pad, ups=0, act=0,
                    alpha=1.0, split_k=1, batch=1, residual=(r.data_ptr() if res else 0), ldr=Cin)

    def wkw(dw, db, ws, sp):
        return dict(a=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(), dbias=(db.data_ptr() if bias else 0),
                    workspace=ws.data_ptr(), dtype=_lib.BF16, M=M, N=Cout, K=K, lda=Cin, ldy=Cout, B=Bb, H=Hh,
                    W=Hh, Cin=Cin, Ho=Hh, Wo=Hh, KH=k, KW=k, stride=1, pad_t=pad, pad_l=pad, ups=0, splits=sp,
                    accumulate=1)
    # ---- separate launches
    dx0 = torch.empty(M, Cin, dtype=torch.bfloat16, device='cuda')
    dw0, db0 = init.clone(), binit.clone()
    ws0 = torch.empty(splits * (Cout * K + Cout), device='cuda')
    _lib.call('sdmi_igemm', st, **dkw(dx0))
    _lib.call('sdmi_wgrad', st, **wkw(dw0, db0, ws0, splits))
    # ---- one launch (+ a second pair launch that folds the first one's partials, + the final fold)
    S = _lib.CSTRUCT

    def pair(dx, dw, db, ws, sp, fold, wt=0):
        d, wg = S['SdmiGemmArgs'](), S['SdmiWgradArgs']()
        for kk, v in dkw(dx).items():
            setattr(d, kk, v)
        for kk, v in wkw(dw, db, ws, sp).items():
            setattr(wg, kk, v)
        wg.defer_fold = 1
        f = None
        if fold is not None:
            f = S['SdmiWgradArgs']()
            for kk, v in fold.items():
                setattr(f, kk, v)
        _lib.call('sdmi_bwd_pair', st, dgrad=ctypes.addressof(d), wgrad=ctypes.addressof(wg),
                  fold=(ctypes.addressof(f) if f is not None else 0), dgrad_cap=0, wgrad_tile=wt)
    dx1 = torch.empty_like(dx0)
    dw1, db1 = init.clone(), binit.clone()
    ws1 = torch.empty_like(ws0)
    pair(dx1, dw1, db1, ws1, splits, None)
    fold1 = dict(dw=dw1.data_ptr(), dbias=(db1.data_ptr() if bias else 0), workspace=ws1.data_ptr(), N=Cout, K=K,
                 splits=splits, accumulate=1) if splits > 1 else None
    # second "layer" (same operands, other destinations) whose launch folds the first one's partials; its
    # own weight gradient on 64 x 64 tiles (every dW element is the same accumulation chain: still bit-exact)
    dx2 = torch.empty_like(dx0)
    dw2, db2 = init.clone(), binit.clone()
    ws2 = torch.empty_like(ws0)
    big = ((M + 127) // 128) * ((Cin + 127) // 128) >= 192      # 128 x 128 data-gradient tiles: dW tiles stay 128 wide
    pair(dx2, dw2, db2, ws2, splits, fold1, wt=(0 if big else 64))
    if splits > 1:
        arr = (S['SdmiWgradArgs'] * 1)()
        for kk, v in dict(dw=dw2.data_ptr(), dbias=(db2.data_ptr() if bias else 0), workspace=ws2.data_ptr(), N=Cout,
                          K=K, splits=splits, accumulate=1).items():
            setattr(arr[0], kk, v)
        _lib.call('sdmi_wgrad_fold_group', st, problems=ctypes.addressof(arr), n=1)
    torch.cuda.synchronize()
    for dxp, dwp, dbp, exact_bias in ((dx1, dw1, db1, True), (dx2, dw2, db2, big)):
        assert torch.equal(dxp, dx0), case
        assert torch.equal(dwp, dw0), case
        if bias and exact_bias:
            assert torch.equal(dbp, db0), case
        elif bias:       # 64-wide bias tiles sum the rows in a different thread partition: fp32 reordering only
            assert float((dbp - db0).abs().max()) <= 1e-5 * float(db0.abs().max()), case
    # ---- and the numbers themselves (torch fp32 on the bf16-rounded operands)
    if conv:
        x4 = xf.float().view(B, H, H, Cin).permute(0, 3, 1, 2).requires_grad_(True)
        w4 = wf.float().view(Cout, k, k, Cin).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        y = F.conv2d(x4, w4, None, padding=pad)
        y.backward(dyf.float().view(B, H, H, Cout).permute(0, 3, 1, 2))
        ref_dx = x4.grad.permute(0, 2, 3, 1).reshape(M, Cin)
        ref_dw = w4.grad.permute(0, 2, 3, 1).reshape(Cout, K)
    else:
        ref_dx = dyf.float() @ wf.float()
        ref_dw = dyf.float().t() @ xf.float()
    if res:
        ref_dx = ref_dx + rf.float()
    assert float((dx1.float().cpu() - ref_dx).norm() / ref_dx.norm()) < 1e-2, case
    assert float((dw1.cpu() - init.cpu() - ref_dw).norm() / ref_dw.norm()) < 3e-3, case
    if bias:
        rb = dyf.float().sum(0)
        assert float((db1.cpu() - binit.cpu() - rb).norm() / rb.norm()) < 3e-3, case
    # argument validation: narrow outputs / mismatched loader classes are refused without a launch
    d, wg = S['SdmiGemmArgs'](), S['SdmiWgradArgs']()
    for kk, v in dict(dkw(dx1), N=32).items():
        setattr(d, kk, v)
    for kk, v in wkw(dw1, db1, ws1, 1).items():
        setattr(wg, kk, v)
    with pytest.raises(_lib.SdmiError):
        _lib.call('sdmi_bwd_pair', st, dgrad=ctypes.addressof(d), wgrad=ctypes.addressof(wg), fold=0, dgrad_cap=0)


def test_pair_backward_equals_side_stream_backward():
    """The train step's backward with the fused data / weight gradient launches (sdmi_bwd_pair, the
    default) against the same step with separate launches on side streams (pair_bwd = False): the
    gradients of the bf16 model agree to bf16 rounding noise -- the tile bodies are the same, but the
    separate path runs skinny data gradients with split-K, so a dX element may round to the neighbouring
    bf16 value and the difference travels down the chain (measured 1.8e-3 relative; each path alone is
    pinned against the reference by the gradient tests) -- and the pair path replaces most igemm + wgrad
    launch pairs."""
    from slotdiffusion_amd._lib import KernelTimer
    G = C.load_golden()
    img = C.make_inputs(2)[0].cuda()
    arenas, launches = [], []
    m = _model(torch.bfloat16)
    for pair in (True, False):
        m.bank().pair_bwd = pair
        _train_backward(m, G, img)                 # warm-up (lazy operands)
        with KernelTimer() as kt:
            _train_backward(m, G, img)
        summ = kt.summary()
        torch.cuda.synchronize()
        arenas.append(m.grad_arena().clone())
        launches.append({k: v['calls'] for k, v in summ.items() if k in ('sdmi_igemm', 'sdmi_wgrad', 'sdmi_bwd_pair')})
    m.bank().pair_bwd = True
    REPORT['pair_bwd_launches'] = launches
    a, b = arenas
    rel = float((a - b).norm() / b.norm())
    REPORT['pair_vs_side_grad_rel_l2'] = rel
    _dump()
    assert launches[0].get('sdmi_bwd_pair', 0) > 100 and launches[1].get('sdmi_bwd_pair', 0) == 0, launches
    assert rel < 1e-2, rel


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_layernorm_and_gemm_fanout(dtype):
    """LayerNormFn / GemmFn alias outputs: the residual branch's gradient is summed inside the
    LayerNorm backward kernel / the dgrad epilogue (stride 1, stride 2 full and partial parity cover,
    linear)."""
    from slotdiffusion_amd.kern import LayerNormFn, GemmFn
    g = torch.Generator().manual_seed(3)
    q = lambda t: t.to(dtype).float()
    tol = 3e-5 if dtype == torch.float32 else 2e-2
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    rows, C = 300, 256
    x0 = q(torch.randn(rows, C, generator=g) * 2 + 0.5)
    gam, bet = 1 + 0.3 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    dy, d1 = q(torch.randn(rows, C, generator=g)), q(torch.randn(rows, C, generator=g))

    class WB:
        class model:
            @staticmethod
            def grad_arena():
                return WB.g
            _offsets = {'n.weight': (0, C), 'n.bias': (C, C)}
        t = {'n.weight': gam.cuda(), 'n.bias': bet.cuda()}
        g = torch.zeros(2 * C, device='cuda')
        defer_colsum = False

        @staticmethod
        def f(k):
            return WB.t[k]
    xr = x0.clone().requires_grad_(True)
    F.layer_norm(xr, (C,), gam, bet, 1e-5).backward(dy)
    xd = x0.to(dtype).cuda().requires_grad_(True)
    anchor = torch.zeros(1, device='cuda', requires_grad=True)
    y, a = LayerNormFn.apply(xd, anchor, WB, 'n', 1)
    torch.autograd.backward([y, a], [dy.to(dtype).cuda(), d1.to(dtype).cuda()])
    assert rel(xd.grad, xr.grad + d1) <= tol
    # convolution fan-out through the model-level provider (real WeightBank)
    m = _model(dtype)
    m.train()
    K = m.KG()
    for name, stride, kh, pad in (('encoder.layer1.0.conv1.weight', 1, 3, (1, 1, 1, 1)),
                                  ('encoder.layer2.0.conv1.weight', 2, 3, (1, 1, 1, 1)),
                                  ('encoder.layer2.0.downsample.0.weight', 2, 1, (0, 0, 0, 0))):
        w = m.state_dict()[name].float().cpu()
        cin = w.shape[1]
        xi = q(torch.randn(2, cin, 16, 16, generator=g))
        xr = xi.clone().requires_grad_(True)
        yr = F.conv2d(xr, q(w), None, stride, pad[0])
        dyc = q(torch.randn(yr.shape, generator=g))
        dal = q(torch.randn(xi.shape, generator=g))
        yr.backward(dyc)
        xd = xi.permute(0, 2, 3, 1).contiguous().to(dtype).cuda().requires_grad_(True)
        m.grad_arena().zero_()
        o, al = K.conv_fan(xd, name, kh=kh, kw=kh, stride=stride, pad=pad)
        torch.autograd.backward([o, al], [dyc.permute(0, 2, 3, 1).contiguous().to(dtype).cuda(),
                                          dal.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()])
        assert rel(xd.grad.permute(0, 3, 1, 2), xr.grad + dal) <= (tol if dtype == torch.float32 else 3e-2), name


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('cfg', [(2, 8, 256, 256), (2, 12, 64, 64), (3, 16, 16, 16), (2, 8, 256, 7),
                                 (2, 4, 50, 15), (1, 2, 130, 100), (2, 5, 784, 784), (1, 2, 1000, 600),
                                 (1, 3, 520, 1024), (2, 6, 784, 7), (1, 2, 1030, 11),
                                 (1, 2, 513, 512), (1, 2, 512, 513), (1, 1, 129, 33)])
def test_attention_bwd_kernel(cfg, dtype):
    """dq/dk/dv of the attention backward (matrix-core kernels for bf16, VALU kernels for fp32)
    against torch autograd; inputs are quantised to the compute dtype first.  The cases beyond 400
    keys (bf16 only) cross the 512-row LDS chunks of the matrix-core kernels on both sides."""
    from slotdiffusion_amd import _lib, ops
    from slotdiffusion_amd.kern import _DT
    B, heads, Sq, Skv = cfg
    if Skv > ops.ATTN_LDS_MAX_KV and dtype != torch.bfloat16:
        pytest.skip('fp32 keeps K/V whole in LDS: at most 400 keys')
    C = heads * 32
    g = torch.Generator().manual_seed(Sq * 3 + Skv)
    qz = lambda t: t.to(dtype).float()
    qq = qz(torch.randn(B, Sq, C, generator=g)).requires_grad_(True)
    kk = qz(torch.randn(B, Skv, C, generator=g)).requires_grad_(True)
    vv = qz(torch.randn(B, Skv, C, generator=g)).requires_grad_(True)
    dout = qz(torch.randn(B, Sq, C, generator=g))
    sp = lambda t, S: t.view(B, S, heads, 32).permute(0, 2, 1, 3)
    sim = torch.einsum('bhid,bhjd->bhij', sp(qq, Sq), sp(kk, Skv)) * 32 ** -0.5
    ref = torch.einsum('bhij,bhjd->bhid', sim.softmax(-1), sp(vv, Skv))
    ref.permute(0, 2, 1, 3).reshape(B, Sq, C).backward(dout)
    qd, kd, vd = (t.detach().to(dtype).cuda() for t in (qq, kk, vv))
    lse = torch.empty(B, heads, Sq, device='cuda')
    out = ops.attention(qd, kd, vd, heads, lse=lse)
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    dod = dout.to(dtype).cuda()
    _lib.call('sdmi_attention_bwd', torch.cuda.current_stream().cuda_stream, q=qd.data_ptr(),
              k=kd.data_ptr(), v=vd.data_ptr(), out=out.data_ptr(), dout=dod.data_ptr(),
              lse=lse.data_ptr(), dq=dq.data_ptr(), dk=dk.data_ptr(), dv=dv.data_ptr(),
              dtype=_DT[dtype], B=B, heads=heads, Sq=Sq, Skv=Skv, ldq=C, ldk=C, ldv=C, ldo=C,
              scale=32 ** -0.5, head_dim=32)
    tol = 3e-5 if dtype == torch.float32 else 2.5e-2
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    assert rel(dq, qq.grad) <= tol, ('dq', rel(dq, qq.grad))
    assert rel(dk, kk.grad) <= tol, ('dk', rel(dk, kk.grad))
    assert rel(dv, vv.grad) <= tol, ('dv', rel(dv, vv.grad))


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('groups,rows,N,pitch', [(16, 784, 384, 384), (3, 130, 70, 96), (2, 50, 40, 40),
                                                  (1, 16, 3000, 3000), (4, 256, 62, 64)])
def test_rowgroup_sum(groups, rows, N, pitch, dtype):
    """Per-group column sums (time-embedding / position gradients): the column-per-thread kernel and the
    row-parallel one for tall groups against an fp64 sum of the same (dtype-rounded) values."""
    from slotdiffusion_amd import _lib
    from slotdiffusion_amd.kern import _DT
    g = torch.Generator().manual_seed(rows + N)
    x = torch.randn(groups * rows, pitch, generator=g).to(dtype)
    ref = x[:, :N].double().view(groups, rows, N).sum(1)
    xd = x.cuda()
    out = torch.full((groups, N + 3), -7.0, device=DEV)
    _lib.call('sdmi_rowgroup_sum', torch.cuda.current_stream().cuda_stream, x=xd.data_ptr(),
              out=out.data_ptr(), dtype=_DT[dtype], groups=groups, rows_per=rows, N=N, ldx=pitch, ldo=N + 3)
    assert float((out[:, :N].cpu().double() - ref).abs().max()) <= 2e-5 * math.sqrt(rows) * 4
    assert bool((out[:, N:] == -7.0).all())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,heads,S,hd', [(2, 4, 784, 32), (1, 6, 785, 64), (2, 2, 408, 32)])
def test_long_sequence_attention(B, heads, S, hd, dtype):
    """Key sequences beyond the LDS-resident kernels (28 x 28 UNet tokens of the 224^2 configs, the
    785 DINO tokens): per-head batched-GEMM attention, forward and (for 16-byte-aligned S) the
    autograd backward, against torch."""
    from slotdiffusion_amd import ops
    from slotdiffusion_amd.kern import LongAttnFn
    C = heads * hd
    g = torch.Generator().manual_seed(S + hd)
    qkv = (torch.randn(B, S, 3 * C, generator=g) * 0.7).to(dtype).float().requires_grad_(True)
    dout = torch.randn(B, S, C, generator=g).to(dtype).float()
    sp = lambda t: t.view(B, S, heads, hd).permute(0, 2, 1, 3)
    q, k, v = (sp(qkv[..., j * C:(j + 1) * C]) for j in range(3))
    ref = (torch.einsum('bhid,bhjd->bhij', q, k) * hd ** -0.5).softmax(-1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(B, S, C)
    ref.backward(dout)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    xd = qkv.detach().to(dtype).cuda()
    out = ops.attention_long(xd[..., :C], xd[..., C:2 * C], xd[..., 2 * C:], heads, hd)
    assert rel(out, ref.detach()) <= tol
    if S % ops.vec_of(dtype) == 0:
        xd.requires_grad_(True)
        o = LongAttnFn.apply(xd, heads, hd)
        assert rel(o.detach(), ref.detach()) <= tol
        o.backward(dout.to(dtype).cuda())
        assert rel(xd.grad, qkv.grad) <= (tol if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,M,N,D', [(2, 1024, 7, 192), (3, 200, 5, 192), (2, 130, 8, 128),
                                     (2, 1024, 15, 192), (2, 300, 11, 192), (1, 784, 16, 256), (2, 784, 7, 256)])
def test_sa_attend_tiled_matches_reference(B, M, N, D, dtype):
    """Token-tiled Slot-Attention pass (forward and backward) against torch autograd of the
    reference formula (sa_diffusion.py:40-58) and against the one-workgroup-per-image kernels."""
    from slotdiffusion_amd import _lib
    from slotdiffusion_amd.kern import _DT
    g = torch.Generator().manual_seed(M + N)
    qz = lambda t: t.to(dtype).float()
    kv = qz(torch.randn(B, M, 2 * D, generator=g)).requires_grad_(True)
    q = torch.randn(B, N, D, generator=g).requires_grad_(True)
    dupd = torch.randn(B, N, D, generator=g)
    eps, scale = 1e-6, D ** -0.5
    k, v = kv[..., :D], kv[..., D:]
    attn = (torch.einsum('bnd,bmd->bmn', q * scale, k)).softmax(-1)
    w = attn + eps
    den = w.sum(1)
    upd = torch.einsum('bmn,bmd->bnd', w, v) / den[..., None]
    upd.backward(dupd)
    kvd, qd, dud = kv.detach().to(dtype).cuda(), q.detach().cuda(), dupd.cuda()
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for tiled in (True, False):
        a_o = torch.empty(B, M, N, device='cuda')
        u_o, d_o = torch.empty(B, N, D, device='cuda'), torch.empty(B, N, device='cuda')
        ws = torch.empty(B * ((M + 63) // 64) * N * (D + 1), device='cuda')
        common = dict(k=kvd.data_ptr(), v=kvd[..., D:].data_ptr(), q=qd.data_ptr(), dtype=_DT[dtype],
                      B=B, M=M, N=N, D=D, ldkv=2 * D, eps=eps, scale=scale,
                      workspace=(ws.data_ptr() if tiled else 0))
        _lib.call('sdmi_sa_attend_fwd', st, attn=a_o.data_ptr(), upd=u_o.data_ptr(),
                  den=d_o.data_ptr(), **common)
        dq = torch.empty_like(qd)
        dkv = torch.empty_like(kvd)
        _lib.call('sdmi_sa_attend_bwd', st, attn=a_o.data_ptr(), upd=u_o.data_ptr(),
                  den=d_o.data_ptr(), dupd=dud.data_ptr(), dq=dq.data_ptr(), dk=dkv.data_ptr(),
                  dv=dkv[..., D:].data_ptr(), **common)
        res[tiled] = (a_o, u_o, d_o, dq, dkv)
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    a_o, u_o, d_o, dq, dkv = res[True]
    assert rel(a_o, attn.detach()) <= 1e-5 and rel(u_o, upd.detach()) <= 1e-5
    assert rel(d_o, den.detach()) <= 1e-5
    assert rel(dq, q.grad) <= tol and rel(dkv, kv.grad) <= tol
    for x, y in zip(res[True], res[False]):
        assert rel(x, y.float().cpu()) <= (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_pack_dgrad_batch_matches_single(dtype):
    """One-launch repack of a table of dgrad operands == the per-operand kernel (bit exact)."""
    from slotdiffusion_amd import _lib, ops
    from slotdiffusion_amd.kern import _DT
    vec = ops.vec_of(dtype)
    shapes = [(3, 3, 3, 128), (64, 3, 3, 64), (200, 1, 1, 136), (384, 3, 3, 640 if dtype == torch.bfloat16 else 72),
              (8, 1, 1, 8)]
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device='cuda').manual_seed(3)
    Desc = _lib.CSTRUCT['SdmiPackDesc']
    arr = (Desc * len(shapes))()
    keep, blk = [], 0
    for d, (co, kh, kw, ci) in zip(arr, shapes):
        npad = (co + vec - 1) // vec * vec
        src = torch.randn(co, kh * kw * ci, device='cuda', generator=g).to(dtype)
        ref = torch.zeros(ci * kh * kw, npad, dtype=dtype, device='cuda')
        out = torch.zeros_like(ref)
        _lib.call('sdmi_pack_dgrad', st, src=src.data_ptr(), dst=ref.data_ptr(), dtype=_DT[dtype],
                  Cout=co, KH=kh, KW=kw, Cin=ci, CoutPad=npad)
        d.src, d.dst = src.data_ptr(), out.data_ptr()
        d.Cout, d.KH, d.KW, d.Cin, d.CoutPad, d.block_begin = co, kh, kw, ci, npad, blk
        blk += kh * kw * ((co + 63) // 64) * ((ci + 63) // 64)
        keep.append((src, ref, out))
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    _lib.call('sdmi_pack_dgrad_batch', st, descs=tab.data_ptr(), n_desc=len(shapes), dtype=_DT[dtype],
              total_blocks=blk)
    for (src, ref, out), shp in zip(keep, shapes):
        assert torch.equal(ref, out), shp


def test_ema_hook_matches_reference_formula():
    """Row a18: LitEma semantics (ddpm/ema.py:29-85) on the flat arena -- decay warm-up
    min(decay, (1+n)/(10+n)), in-place shadow update, ema_scope store / copy_to / restore."""
    from oracle import slotdiff_oracle as O
    m = _model(torch.float32)
    dm = m.dm_decoder
    assert dm.use_ema is False
    with dm.ema_scope():          # disabled: a no-op context
        pass
    dm.enable_ema(decay=0.9999)
    lo, hi = dm._ema_range()
    names = [n for n, _ in m.named_parameters() if n.startswith('dm_decoder.model.')]
    assert sum(dict(m.named_parameters())[n].numel() for n in names) <= hi - lo
    shadow_ref = [m.arena()[lo:hi].detach().cpu().clone()]
    nupd = 0
    g = torch.Generator(device='cuda').manual_seed(11)
    for it in range(3):
        with torch.no_grad():
            m.arena()[lo:hi].add_(0.01 * torch.randn(hi - lo, device='cuda', generator=g))
        m._training_step_end()
        nupd = O.ema_update(shadow_ref, [m.arena()[lo:hi].detach().cpu()], nupd, 0.9999)
    assert dm.ema_num_updates == nupd == 3
    err = float((dm._ema_shadow.cpu() - shadow_ref[0]).abs().max())
    assert err <= 1e-6, err
    before = m.arena().detach().clone()
    with dm.ema_scope('eval'):
        assert torch.equal(m.arena()[lo:hi], dm._ema_shadow)
        assert torch.equal(m.arena()[:lo], before[:lo])
    assert torch.equal(m.arena(), before)


def test_ema_kernel_matches_reference_litema_fixture():
    """Row a18 against the reference itself: sdmi_ema_update with the host-side decay arithmetic of
    LDM._training_step_end on the parameter sequence of tests/golden/ema_lit.npz (the reference's
    LitEma run, tools/gen_golden.py ema): shadows after each of 4 updates, with the num_updates
    warm-up and with a fixed decay."""
    import numpy as np
    from slotdiffusion_amd import _lib
    G = C.load_golden('ema_lit.npz')
    st = torch.cuda.current_stream().cuda_stream
    worst, exact = 0.0, True
    for tag, decay, n in (('warm', 0.9999, 0), ('flat', 0.95, -1)):
        shadow = torch.cat([G['init_a'].flatten(), G['init_w'].flatten()]).cuda()
        for it in range(4):
            p = torch.cat([G[f'{tag}_p_a_{it}'].flatten(), G[f'{tag}_p_w_{it}'].flatten()]).cuda()
            d = np.float32(decay)
            if n >= 0:
                n += 1
                d = min(d, np.float32(1 + n) / np.float32(10 + n))
            _lib.call('sdmi_ema_update', st, shadow=shadow.data_ptr(), p=p.data_ptr(), n=shadow.numel(),
                      one_minus_decay=float(np.float32(1.0) - d))
            ref = torch.cat([G[f'{tag}_s_a_{it}'].flatten(), G[f'{tag}_s_w_{it}'].flatten()])
            worst = max(worst, float((shadow.cpu() - ref).abs().max()))
            exact = exact and torch.equal(shadow.cpu(), ref)
        assert n == int(G[f'{tag}_num_updates'])
    REPORT['ema_vs_litema_maxerr'] = worst
    REPORT['ema_vs_litema_bit_exact'] = bool(exact)
    _dump()
    assert worst <= 2.4e-7, worst


@pytest.mark.parametrize('name', ['SADiffusion', 'SA', 'VQVAE'])
def test_method_fit_through_the_registry(name):
    """build_model -> build_dataset -> build_method -> fit(): the plugin path of scripts/train.py,
    3 optimiser steps on synthetic data (graph replay on), losses finite, weights move."""
    from slotdiffusion_amd import img_based as task
    P = C.make_params(name)
    model = task.build_model(P)
    det_fill_(model.state_dict().items(), skip=is_buffer_name)
    model = model.cuda()
    model.set_compute_dtype('fp32')
    before = model.arena().detach().clone()
    dm = task.build_dataset(P)
    dm.steps_per_epoch = 5
    method = task.build_method(model=model, datamodule=dm, params=P, ckp_path=None, local_rank=0,
                               use_ddp=False, use_fp16=False)
    method.fit(resume_from='', san_check_val_step=0, max_steps=5)
    losses = [float(l) for l in method.history]
    n_logged = 5
    assert len(losses) == n_logged and all(math.isfinite(l) for l in losses), losses
    # (no monotonicity check: 5 steps at the config's lr with a 0.25-step warm-up, a new random batch
    # every step and update 1 at lr 0 -- gradient parity of stage-1 training is pinned by
    # test_vqvae_stage1_training_gradients)
    assert method.optimizer.step_count == 5
    assert float((model.arena() - before).abs().max()) > 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_vae_train_kernels(dtype):
    """VQ-VAE stage-1 backward helpers: batched transpose, softmax backward, the single-head
    AttnBlock core (VaeAttnFn) against torch autograd, the straight-through quantizer backward and
    the L1 loss."""
    from slotdiffusion_amd import ops
    from slotdiffusion_amd.kern import MseFn, VaeAttnFn, VqFn
    g = torch.Generator().manual_seed(5)
    q = lambda t: t.to(dtype).float()
    # transpose (ragged sizes, strided source)
    x = q(torch.randn(3, 70, 200, generator=g))
    xd = x.to(dtype).cuda()
    assert torch.equal(ops.transpose2d(xd[..., 8:136]).cpu().float(), x[..., 8:136].transpose(1, 2))
    # attention core
    B, S, C = 2, 192, 64
    qkv = q(torch.randn(B, S, 3 * C, generator=g)).requires_grad_(True)
    do = q(torch.randn(B, S, C, generator=g))
    qq, kk, vv = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    p = torch.softmax(qq @ kk.transpose(1, 2) * C ** -0.5, -1)
    (p @ vv).backward(do)
    qd = qkv.detach().to(dtype).cuda().requires_grad_(True)
    o = VaeAttnFn.apply(qd)
    o.backward(do.to(dtype).cuda())
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    rel = lambda a, b: float((a.float().cpu() - b).norm() / b.norm())
    assert rel(o.detach(), (p @ vv).detach()) <= tol
    assert rel(qd.grad, qkv.grad) <= tol, rel(qd.grad, qkv.grad)
    if dtype == torch.bfloat16:
        return
    # straight-through quantizer: gradients of  w1 * sum(zq * a) + w2 * quant_loss
    R, n_codes, beta = 300, 64, 0.25
    code = torch.randn(n_codes, 3, generator=g).requires_grad_(True)
    z = F.pad(torch.randn(R, 3, generator=g), (0, 1)).requires_grad_(True)
    a = torch.randn(R, 4, generator=g)
    d = (z[:, :3] ** 2).sum(1, keepdim=True) + (code ** 2).sum(1) - 2 * z[:, :3] @ code.t()
    idx = d.argmin(1)
    zq = code[idx]
    ql = ((zq.detach() - z[:, :3]) ** 2).mean() + beta * ((zq - z[:, :3].detach()) ** 2).mean()
    st = z[:, :3] + (zq - z[:, :3]).detach()
    ((st * a[:, :3]).sum() * 0.7 + 1.3 * ql).backward()
    zd = z.detach().cuda().requires_grad_(True)
    dcode = torch.zeros(n_codes, 3, device='cuda')
    anchor = torch.zeros(1, device='cuda', requires_grad=True)
    zq_d, ql_d, idx_d = VqFn.apply(zd, anchor, code.detach().cuda(), dcode, beta)
    assert torch.equal(idx_d.cpu(), idx) and abs(float(ql_d) - float(ql)) <= 1e-6
    ((zq_d * a.cuda()).sum() * 0.7 + 1.3 * ql_d).backward()
    assert float((zd.grad.cpu() - z.grad).abs().max()) <= 1e-6
    assert float((dcode.cpu() - code.grad).abs().max()) <= 1e-6
    # L1 loss
    pr = torch.randn(2, 8, 8, 4, generator=g)
    tg = torch.randn(2, 8, 8, 4, generator=g)
    pd = pr.cuda().requires_grad_(True)
    l = MseFn.apply(pd, tg.cuda(), 1.0, True)
    l.backward()
    pr.requires_grad_(True)
    lr = (pr - tg).abs().mean()
    lr.backward()
    assert abs(float(l) - float(lr)) <= 1e-6 and float((pd.grad.cpu() - pr.grad).abs().max()) <= 1e-7


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('case', [(2, 32, 48, 16, 3, 2, (1, 1, 1, 1)),      # UNet / ResNet downsample
                                  (2, 32, 40, 16, 3, 2, (0, 1, 0, 1)),      # VQ-VAE asymmetric pad
                                  (2, 64, 128, 16, 1, 2, (0, 0, 0, 0)),     # 1x1 stride 2: empty parities
                                  (1, 16, 32, 15, 3, 2, (1, 1, 1, 1)),      # odd image
                                  (2, 16, 24, 18, 5, 3, (2, 2, 2, 2))])     # 5x5 stride 3
def test_strided_dgrad_by_parity(case, dtype):
    """Data gradient of stride-s convolutions as s*s stride-1 convolutions written interleaved
    (igemm's sub-sampled output placement) against torch autograd."""
    from slotdiffusion_amd import _lib, ops
    from slotdiffusion_amd.kern import _DT, GemmFn
    B, Cin, Cout, H, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    q = lambda t: t.to(dtype).float()
    x = q(torch.randn(B, Cin, H, H, generator=g)).requires_grad_(True)
    w = q(torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k))
    y = F.conv2d(F.pad(x, (pad[2], pad[3], pad[0], pad[1])), w, None, stride=stride)
    dy = q(torch.randn(y.shape, generator=g))
    y.backward(dy)
    vec = ops.vec_of(dtype)
    npad = (Cout + vec - 1) // vec * vec
    Ho = y.shape[2]
    K = k * k * Cin
    dyd = F.pad(dy.permute(0, 2, 3, 1), (0, npad - Cout)).contiguous().to(dtype).cuda()
    wp = w.permute(0, 2, 3, 1).reshape(Cout, K).contiguous().to(dtype).cuda()
    wd = torch.zeros(Cin * k * k, npad, dtype=dtype, device='cuda')
    _lib.call('sdmi_pack_dgrad', torch.cuda.current_stream().cuda_stream, src=wp.data_ptr(),
              dst=wd.data_ptr(), dtype=_DT[dtype], Cout=Cout, KH=k, KW=k, Cin=Cin, CoutPad=npad)
    dx, _ = GemmFn._dgrad_strided(dyd, wd.view(Cin, k * k * npad), B, H, H, Ho, Ho, Cin, npad, k, k,
                                  stride, pad, dtype)
    ref = x.grad.permute(0, 2, 3, 1)
    e = float((dx.float().cpu() - ref).norm() / ref.norm())
    assert e <= (2e-5 if dtype == torch.float32 else 2e-2), e
    # with the gradient of x's other consumer riding in the epilogue (full parity cover) or handed
    # back to the caller (partial cover)
    extra = q(torch.randn(B, H, H, Cin, generator=g)).to(dtype).cuda()
    dx2, left = GemmFn._dgrad_strided(dyd, wd.view(Cin, k * k * npad), B, H, H, Ho, Ho, Cin, npad, k, k,
                                      stride, pad, dtype, extra)
    if left is not None:
        dx2 = dx2.float() + left.float()
    e2 = float((dx2.float().cpu() - (ref + extra.float().cpu())).norm() / ref.norm())
    assert e2 <= (2e-5 if dtype == torch.float32 else 3e-2), e2


def test_fp8_train_step_deviation_and_device_scales():
    """BASELINE configs[4] names an "fp8 MFMA UNet"; the train step of that mode (models.set_compute_dtype('fp8'))
    multiplies e4m3fn operands in the forward 3x3 convolutions of the denoiser -- weights re-quantised on the device
    every step (sdmi_fp8_quant_group: scale = 448 / amax, no read-back, part of the captured step), activations at
    the inference path's fixed scale -- and keeps the bf16 backward.  Against the bf16 step on the same inputs:
    loss within 5 %, eps within 10 % rel-L2 (VERDICT r4 item 7's bars), gradients close; the graphed step replays
    with weights that change, so the quantised bytes and the scales must follow."""
    from slotdiffusion_amd import _lib, ops, optim
    G = C.load_golden()
    img = C.make_inputs(2)[0].cuda()
    R = {}
    grads, eps = {}, {}
    for mode in ('bf16', 'fp8'):
        m = _model(torch.bfloat16)
        m.set_compute_dtype(mode)
        assert m.fp8_unet == (mode == 'fp8') and m.compute_dtype == torch.bfloat16
        with _lib.KernelTimer() as kt:
            loss = _train_backward(m, G, img)
        torch.cuda.synchronize()
        R[mode + '_loss'] = float(loss)
        grads[mode] = m.grad_arena().float().clone()
        n8 = sum(1 for r in kt.records if r[0] == 'sdmi_igemm' and r[3].get('fp8'))
        nq = sum(1 for r in kt.records if r[0] == 'sdmi_fp8_quant_group')
        if mode == 'fp8':
            R['fp8_igemm_launches'], R['fp8_quant_group_calls'] = n8, nq
            R['fp8_quant_launches'] = sum(1 for r in kt.records if r[0] == 'sdmi_quant_fp8')
            assert n8 >= 30 and nq >= 30         # first step: one registration call per operand
            assert R['fp8_quant_launches'] == 0  # the GroupNorms in front write the e4m3fn operands themselves
        else:
            assert n8 == 0 and nq == 0
        # eps of the training provider on the fixture's x_t / t / slots
        xt = m._latent_nhwc(G['x_t'].cuda())
        with torch.enable_grad():
            e = m._unet_eps(xt, G['t'].float().cuda(), G['slots'].cuda().requires_grad_(True), m.KG())
        eps[mode] = ops.nhwc_to_nchw(e.detach(), 3).float().cpu()
        m.bank().join()
        if mode == 'fp8':
            wb = m.bank()
            names = sorted(wb._w8, key=lambda n: wb._w8[n][4])
            assert len(names) >= 30
            # every operand: bytes = e4m3fn(w * 448 / amax) of the fp32 master, inv = amax / 448
            for n in names[:4] + names[-2:]:
                b8, inv = wb.w8_dev(n)
                w = wb._flat(n, torch.float32)
                am = w.abs().max()
                assert abs(float(inv) - float(am / 448.0)) <= 1e-6 * float(am / 448.0)
                ref = ops.quant_fp8(w.contiguous(), float(448.0 / am))
                assert float((ref != b8).float().mean()) <= 1e-4
            # an optimiser step changes the weights: ONE grouped call re-quantises all of them
            before = {n: (wb._w8[n][0].clone(), float(wb._w8[n][1])) for n in names[:3]}
            with torch.no_grad():
                m.arena().mul_(1.25)
            m.weights_updated()
            with _lib.KernelTimer() as kt:
                wb.w8_dev(names[0])
                wb.w8_dev(names[1])
            assert sum(1 for r in kt.records if r[0] == 'sdmi_fp8_quant_group') == 1
            for n in names[:3]:
                assert abs(float(wb._w8[n][1]) - 1.25 * before[n][1]) <= 1e-5 * before[n][1]
                assert float((wb._w8[n][0] != before[n][0]).float().mean()) <= 1e-3     # same bytes at 1.25x the scale
    R['fp8_eps_rel_l2_vs_bf16'] = float((eps['fp8'] - eps['bf16']).norm() / eps['bf16'].norm())
    R['fp8_eps_rel_l2_vs_ref'] = float((eps['fp8'] - G['eps_pred']).norm() / G['eps_pred'].norm())
    R['bf16_eps_rel_l2_vs_ref'] = float((eps['bf16'] - G['eps_pred']).norm() / G['eps_pred'].norm())
    R['fp8_loss_rel'] = abs(R['fp8_loss'] - R['bf16_loss']) / abs(R['bf16_loss'])
    R['fp8_grad_rel_l2_vs_bf16'] = float((grads['fp8'] - grads['bf16']).norm() / grads['bf16'].norm())
    R['fp8_grad_cosine_vs_bf16'] = float(F.cosine_similarity(grads['fp8'], grads['bf16'], dim=0))
    REPORT['fp8_train'] = R
    _dump()
    assert R['fp8_loss_rel'] < 0.05 and abs(R['fp8_loss'] - float(G['train_loss'])) / float(G['train_loss']) < 0.05
    assert R['fp8_eps_rel_l2_vs_ref'] < 0.10
    assert R['fp8_grad_cosine_vs_bf16'] > 0.95

    # the captured step: graph replays re-quantise the weights they multiply with
    m = _model(torch.bfloat16)
    m.set_compute_dtype('fp8')
    m.train()
    opt = optim.FusedAdam(m, lr=1e-3, dec_lr=1e-3, clip_grad=1.0)
    step = optim.GraphedTrainStep(m, opt, dict(img=img), loss_key='denoise_loss')
    wb = m.bank()
    name = sorted(wb._w8, key=lambda n: wb._w8[n][4])[0]
    losses, invs = [], []
    for _ in range(4):
        losses.append(float(step(dict(img=img))))
        invs.append(float(wb._w8[name][1]))
    torch.cuda.synchronize()
    assert all(math.isfinite(v) for v in losses)
    w = wb._flat(name, torch.float32)
    # after the last replay the slot holds the scale of the weights that step's FORWARD used (before its update):
    # it moved between replays, and the current master is one Adam step (lr 1e-3) away from it
    assert len(set(invs)) > 1
    assert abs(invs[-1] - float(w.abs().max() / 448.0)) <= 0.05 * invs[-1]
