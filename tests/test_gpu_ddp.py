"""Data-parallel train step on a real GPU: two processes (both on cuda:0, gloo transport -- RCCL
refuses two ranks on one device) run the HIP-graph train step with the gradient all-reduce between
its two graphs, each on its half of a batch; the result must equal a single process on the whole
batch (mean loss => averaged half-batch gradients = full-batch gradient)."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import common as C
from tests.detfill import det_fill_, is_buffer_name

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    from slotdiffusion_amd.models import SADiffusion
    cfg = C.clevrtex_cfg()
    m = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                    cfg['loss_dict'], compute_dtype=torch.float32)
    det_fill_(m.state_dict().items(), skip=is_buffer_name)
    m.train_dropout = 0.0
    return m.cuda().train()


def _batch(lo, hi):
    img, t, noise, _ = C.make_inputs(4)
    return dict(img=img[lo:hi].cuda(), t=t[lo:hi].cuda(), noise=noise[lo:hi].cuda())


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from slotdiffusion_amd import parallel
        from slotdiffusion_amd.optim import FusedAdam, GraphedTrainStep
        torch.cuda.set_device(0)
        m = _model()
        parallel.broadcast_parameters(m.arena())
        opt = FusedAdam(m, lr=1e-4, dec_lr=2e-4, clip_grad=1.0)
        batch = _batch(2 * rank, 2 * rank + 2)
        # gradients of the first step at the common initial weights, through the same split
        # backward + range-wise all-reduce the graphed step uses (zero learning rate: weights stay)
        opt0 = FusedAdam(m, lr=0.0, dec_lr=0.0, clip_grad=1.0)
        s0 = GraphedTrainStep(m, opt0, batch, allreduce=True, world=world)
        s0(batch)
        torch.cuda.synchronize()
        # (the exchange leaves the SUM over ranks where the optimiser reads it; 1 / world rides in the Adam kernels)
        torch.save((s0.reducer.grad_src.float() * s0.reducer.grad_scale).detach().cpu(), os.path.join(out_dir, f'grad{rank}.pt'))
        step = GraphedTrainStep(m, opt, batch, allreduce=True, world=world)   # split backward,
        assert step.overlap                                                  # overlapped all-reduce
        step(batch)                                   # 2 warm-up steps inside + 1 replay
        torch.cuda.synchronize()
        torch.save(m.arena().detach().cpu(), os.path.join(out_dir, f'arena{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_two_rank_graphed_step_equals_full_batch():
    from slotdiffusion_amd.optim import FusedAdam
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        a0 = torch.load(os.path.join(d, 'arena0.pt'))
        a1 = torch.load(os.path.join(d, 'arena1.pt'))
        g0 = torch.load(os.path.join(d, 'grad0.pt'))
        g1 = torch.load(os.path.join(d, 'grad1.pt'))
    assert torch.equal(a0, a1)                        # ranks stay in lock-step through the graphs
    assert torch.equal(g0, g1)
    # averaged half-batch gradients == full-batch gradient (compared on gradients: the first Adam
    # steps are sign-like, so parameters are not a well-conditioned comparison)
    m = _model()
    FusedAdam(m, lr=1e-4, dec_lr=2e-4, clip_grad=1.0).zero_grad()
    batch = _batch(0, 4)
    m.calc_train_loss(batch, m(batch))['denoise_loss'].backward()
    ref = m.grad_arena().detach().cpu()
    rel = float((g0 - ref).norm() / ref.norm())
    assert rel <= 2e-5, rel
    moved = float((a0 - m.arena().detach().cpu()).abs().max())
    assert 0 < moved < 1e-2


def test_bench_rccl_path_single_rank():
    """bench.py's multi-GPU code path over the real RCCL backend with ONE rank
    (SDMI_BENCH_FORCE_DIST=1): process group init, parameter broadcast, HIP-graph capture while the
    RCCL watchdog thread is alive, split backward with the overlapped all-reduce, barriers,
    MAX-over-ranks timing -- and the result line must be the LAST line of stdout (RCCL prints its
    version banner to stdout)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDMI_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--only-train', '--no-cpu-baseline',
                        '--no-roofline', '--steps', '2', '--warmup', '1', '--batch', '8'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    out = json.loads(lines[-1])
    assert out['n_gpus'] == 1 and out['unit'] == 'images/s' and out['value'] > 0
    assert out['config']['parallelism'] == 'dp1' and out['steps'] == 2
