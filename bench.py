#!/usr/bin/env python
"""Hot-path benchmark (see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W [--batch B] [--dtype bf16|fp32] [--mode sample|train]

Workload = BASELINE.json configs[1]: img_based SlotDiffusion (Slot Attention + LDM UNet),
CLEVRTex 128x128, 7 slots, bf16 storage / fp32 accumulate, synthetic data, seeded random weights
(zero-initialised layers replaced by small random values so no work is optimised away).

A "step" (mode=sample, default this round) is one 20-NFE DPM-Solver++ sampling pass over a batch
of B images' slots: 20 x (UNet eps evaluation + x0 conversion + VQ quantise) + solver updates,
inputs resident in HBM.  value = B * 20 * K / seconds  [image-denoise-steps/s], whole job.
For N > 1 each rank samples its own B images (data parallel, no collective on this path: weak
scaling); time is max over ranks between barriers.
Extra JSON objects: roofline (MFMA, dominant kernel sdmi_igemm, measured live with HIP events on
the launch stream) and cpu_baseline (the CPU oracle on this box's host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md)


def build_model(dtype, seed=1234):
    from slotdiffusion_amd.models import SADiffusion
    from tests.common import clevrtex_cfg
    cfg = clevrtex_cfg(num_slots=7)
    m = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                    cfg['loss_dict'], compute_dtype=dtype, seed=seed)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        init = {s.name: s.init for s in m._spec}
        for n, p in m.named_parameters():
            if init[n] == 'zlin' and p.dim() > 1:         # zero-init convs -> N(0, 0.02)
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return m, cfg


def synth_batch(B, rank, device):
    g = torch.Generator().manual_seed(1234 + rank)
    img = (torch.randn(B, 3, 128, 128, generator=g) * 0.5).clamp(-1, 1)
    return img.to(device)


def cpu_baseline(cfg, mode, quick=False):
    """Oracle (CPU port of the reference path) timed on this box's host cores, B=4."""
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd import spec
    from tests.common import oracle_weights
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    torch.set_num_threads(threads)
    W = oracle_weights(cfg)
    B = 4
    g = torch.Generator().manual_seed(0)
    img = (torch.randn(B, 3, 128, 128, generator=g) * 0.5).clamp(-1, 1)
    rplan = spec.resnet18_plan(False)
    uplan = spec.unet_plan(cfg['dec_dict']['unet_dict'])
    with torch.no_grad():
        slots, _ = O.sa_encode(W, img, rplan, 3, training=False)
        x_T = torch.randn(B, 3, 32, 32, generator=g)
        nfe = 5 if quick else 20
        betas = W['dm_decoder.betas']
        eps_fn = lambda xc, t_in: O.unet_forward(W, uplan, xc, t_in, slots)
        q_fn = lambda x0: O.vq_quantize(W, x0)[0]
        t0 = time.perf_counter()
        O.dpm_solver_sample(eps_fn, q_fn, betas, x_T, steps=nfe, order=3)
        dt = time.perf_counter() - t0
    return dict(value=B * nfe / dt, unit='image-denoise-steps/s', cores=threads, kind='port',
                sample=f'oracle DPM-Solver++ {nfe} NFE, B={B}, fp32, torch-CPU {threads} threads '
                       f'of {cores} cores, {dt:.1f}s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--mode', default='sample', choices=['sample'])
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32

    model, cfg = build_model(dtype)
    model = model.to(dev).eval()
    model.use_graph = not args.no_graph
    B = args.batch
    img = synth_batch(B, rank, dev)
    from slotdiffusion_amd import ops
    with torch.no_grad():
        slots, _ = model.encode(img)                      # conditioning (not in the timed region)
        g = torch.Generator(device='cpu').manual_seed(77 + rank)
        x_T = ops.nchw_to_nhwc(torch.randn(B, 3, 32, 32, generator=g).to(dev), torch.float32, 4)

        def step():
            return model._dpm_sample(x_T, slots)[0]

        for _ in range(args.warmup):
            step()

        def barrier():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()

        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        nfe = 20
        value = world * B * nfe * args.steps / dt

        out = {
            'metric': 'DPM-Solver denoise-steps/sec, 128^2 7-slot (train-step images/sec: not yet '
                      'native, see DESIGN.md)',
            'value': value, 'unit': 'image-denoise-steps/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'img_based SlotDiffusion CLEVRTex 128x128 7 slots: 20-NFE '
                                   'DPM-Solver++ sampling (UNet eps + VQ per NFE)',
                       'batch_per_gpu': B, 'nfe_per_step': nfe, 'hip_graph': model.use_graph,
                       'parallelism': f'dp{world}'},
        }
        if rank == 0 and not args.no_roofline:
            # live per-kernel timing of ONE sampling pass, eager (events around every launch)
            from slotdiffusion_amd._lib import KernelTimer
            model.use_graph = False
            step()
            with KernelTimer() as kt:
                step()
            summ = kt.summary()
            model.use_graph = not args.no_graph
            ig = summ['sdmi_igemm']
            total_ms = sum(v['ms'] for v in summ.values())
            ach = ig['flops'] / (ig['ms'] * 1e-3) / 1e12
            peak = PEAK_TFLOPS[args.dtype]
            out['roofline'] = {'bound': 'mfma', 'kernel': 'sdmi_igemm (implicit-GEMM conv/linear)',
                               'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                               'traffic': None, 'launches_per_step': ig['calls'],
                               'avg_launch_us': 1e3 * ig['ms'] / ig['calls'],
                               'igemm_ms_per_step': ig['ms'],
                               'all_kernels_ms_per_step_eager': total_ms,
                               'algorithmic_gflop_per_step': ig['flops'] / 1e9}
            out['kernel_breakdown_ms'] = {k: round(v['ms'], 3) for k, v in
                                          sorted(summ.items(), key=lambda kv: -kv[1]['ms'])}
        if rank == 0 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, args.mode)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
