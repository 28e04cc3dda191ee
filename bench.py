#!/usr/bin/env python
"""Hot-path benchmark (see DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W [--config C] [--batch B] [--dtype bf16|fp32|fp8]
                    [--mode train|sample]

Workloads (`--config`, values restated in slotdiffusion_amd/configs.py), bf16 storage / fp32 accumulate,
synthetic data, seeded random weights (zero-initialised layers replaced by small random values so no
work is optimised away):
  clevrtex128 (default) BASELINE.json configs[1], the configuration the metric is quoted on: img_based
               SlotDiffusion (Slot Attention + LDM UNet), CLEVRTex 128x128, 7 slots, 64 images per GPU
  movid11x6    configs[2]: video_based SAVi+LDM, MOVi-D 128x128 x 6-frame clips, 11 slots, 16 clips per GPU
  movie15x6    configs[3]: video_based SlotDiffusion, MOVi-E 128x128 x 6 frames, 15 slots, 16 clips per GPU
               (the configuration north_star states its 2- / 8-GPU targets on)
  coco224      configs[4]: DINO ViT-S/8 encoder, COCO 224x224, 7 slots, 16 images per GPU

A "step" (mode=train, the default) is one optimiser step on a batch: zero-grad, slot encoder forward
(per-frame recurrence + predictor on clips), frozen VQ-VAE encode, q-sample, UNet forward, MSE, full
backward, gradient all-reduce (N > 1), global-norm clip, Adam.  value = N * images * K / seconds
[images/s; a clip counts its frames], whole job.  mode=sample: one 20-NFE DPM-Solver++ sampling pass over
the batch's slots (20 x (UNet eps + x0 conversion + VQ quantise) + solver updates); value = N * images *
20 * K / seconds.  Inputs are resident in HBM; for N > 1 every rank works on its own batch (weak
scaling), time is the max over ranks between barriers.
Extra JSON objects: `roofline` (MFMA, dominant kernel family sdmi_igemm) and `roofline_hbm` (GroupNorm
family) for the timed step, `denoise.roofline` for the sampling leg: algorithmic flops / bytes per
launch counted from the launch arguments, durations from a rocprofv3 --kernel-trace of THIS command's
graph-replayed timed region (child runs on the same box, see replayed_trace(); falls back to a live eager
HIP-event pass, labelled, if rocprofv3 is not usable); `traffic` / `mfma_util_pmc` from rocprofv3 --pmc
passes of this command taken by this run (pmc_pass()); `cpu_baseline` (the CPU oracle on this box's host
cores, bounded sample, pinned thread count and affinity).
"""
import argparse
import json
import os

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC (RCCL across processes)
os.environ.setdefault('DEBUG_HIP_FORCE_GRAPH_QUEUES', '2')  # = slotdiffusion_amd.configure_runtime(), before HIP starts
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3, 'fp8': 5000.0}      # MI355X dense MFMA peaks (MI355X_MICROARCH.md)
CPU_THREADS = 64       # cpu_baseline: fixed thread count, pinned to the first cores (repeatable across boxes)


def build_model(dtype, seed=1234, config='clevrtex128'):
    """-> (model, cfg dict, frames per clip or None)."""
    from slotdiffusion_amd import configs
    from slotdiffusion_amd.models import SADiffusion, SAViDiffusion
    bc = configs.BENCH_CONFIGS[config]
    cfg = bc['cfg']()
    if bc['frames']:
        m = SAViDiffusion(cfg['resolution'], bc['frames'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                          cfg['pred_dict'], cfg['loss_dict'], compute_dtype=dtype, seed=seed)
    else:
        m = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                        cfg['loss_dict'], compute_dtype=dtype, seed=seed)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        init = {s.name: s.init for s in m._spec}
        for n, p in m.named_parameters():
            if init[n] == 'zlin' and p.dim() > 1:         # zero-init convs -> N(0, 0.02)
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    return m, cfg, bc['frames']


def synth_batch(B, rank, device, res=128, frames=None):
    g = torch.Generator().manual_seed(1234 + rank)
    shape = (B, 3, res, res) if not frames else (B, frames, 3, res, res)
    return (torch.randn(shape, generator=g) * 0.5).clamp(-1, 1).to(device)


def cpu_baseline(model, cfg, mode, quick=False):
    """Oracle (CPU port of the reference path) timed on this box's host cores, B=4 (bounded sample).
    The weights are the bench model's own (a CPU copy of its state dict: checkpoint keys are the
    reference's).  Threads: min(CPU_THREADS, cores), pinned to the first cores, so that the figure does
    not depend on how many cores the box happens to have beyond that."""
    from oracle import slotdiff_oracle as O
    from slotdiffusion_amd import spec
    cores = os.cpu_count() or 1
    threads = min(cores, CPU_THREADS)
    pinned = False
    old_aff = None
    try:
        old_aff = os.sched_getaffinity(0)
        os.sched_setaffinity(0, set(sorted(old_aff)[:threads]))
        pinned = True
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(threads)
    W = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    B = 4
    g = torch.Generator().manual_seed(0)
    img = (torch.randn(B, 3, 128, 128, generator=g) * 0.5).clamp(-1, 1)
    rplan = spec.resnet18_plan(False)
    uplan = spec.unet_plan(cfg['dec_dict']['unet_dict'])
    ed = cfg['dec_dict']['vae_dict']['enc_dec_dict']
    aff = f'pinned to {threads} cores' if pinned else 'no affinity control'
    try:
        if mode == 'train':
            P = [W[k].requires_grad_(True) for k in trainable]
            M = [torch.zeros_like(p) for p in P]
            V = [torch.zeros_like(p) for p in P]
            lrs = [2e-4 if 'dm_decoder' in n else 1e-4 for n in trainable]
            t = torch.randint(0, 1000, (B,), generator=g)
            noise = torch.randn(B, 3, 32, 32, generator=g)
            nsteps = 1 if quick else 3
            times = []
            for it in range(nsteps + 1):
                t0 = time.perf_counter()
                for p in P:
                    p.grad = None
                slots, _ = O.sa_encode(W, img, rplan, 3, training=True)
                loss, _, _ = O.ldm_loss(W, uplan, ed, img, slots, t, noise)
                loss.backward()
                with torch.no_grad():
                    O.clip_and_adam(P, [p.grad for p in P], M, V, it + 1, lrs, clip=1.0)
                times.append(time.perf_counter() - t0)
            dt = sorted(times[1:])[len(times[1:]) // 2]
            return dict(value=B / dt, unit='images/s', cores=threads, kind='port',
                        sample=f'oracle train step (fwd+bwd+clip+Adam, no dropout), B={B}, fp32, torch-CPU '
                               f'{threads} threads of {cores} cores ({aff}), median of {nsteps} steps, '
                               f'{dt:.2f}s/step')
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
                           f'of {cores} cores ({aff}), {dt:.1f}s')
    finally:
        if pinned:
            os.sched_setaffinity(0, old_aff)


# kernels behind the C-ABI entry points the roofline legs price (csrc/igemm.hip, wgrad.hip, norm.hip, norm_bwd.hip)
FAMILIES = {
    'sdmi_igemm': ('igemm_kernel', 'igemm_kernel_tall', 'igemm_dma_kernel', 'igemm_sym_kernel',
                   'igemm_halo_kernel', 'conv3x3_c64_kernel', 'splitk_epilogue_kernel', 'bwd_pair_kernel', 'st_block_a_kernel', 'st_block_b_kernel',
                   'st_train_a_kernel', 'st_train_b_kernel', 'st_train_bwd_b1_kernel', 'st_train_bwd_b2_kernel', 'st_train_bwd_a_kernel'),
    'sdmi_wgrad': ('wgrad_kernel', 'wgrad_tr_kernel', 'wgrad3x3_c64_kernel', 'wgrad3x3_halo_kernel', 'wgrad_group_kernel',
                   'wgrad_group_reduce_kernel', 'wgrad_reduce_kernel'),
    'sdmi_groupnorm': ('gn_fused_kernel', 'gn_fused2_kernel', 'gn_stats_kernel', 'gn_apply_kernel'),
    'sdmi_groupnorm_bwd': ('gn_bwd_fused_kernel', 'gn_bwd_fused2_kernel', 'gn_bwd_stats_kernel', 'gn_bwd_apply_kernel'),
}
MARK = 'sqerr_rows_kernel'      # a kernel no train / sampling step launches: brackets the timed region


def kernel_base_name(k):
    import re
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    return re.sub(r'[<(].*', '', k).strip()


def _rocprof_child(extra_prof_args, argv, child_flags, timeout=900, extra_env=None):
    """Run this same command under rocprofv3 (same box) -> (output dir, note) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    d = tempfile.mkdtemp(prefix='sdmi_trace_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    env.update({k: v.replace('%d', d) for k, v in (extra_env or {}).items()})
    cmd = [exe, '--kernel-trace'] + extra_prof_args + ['--output-format', 'csv', '-d', d, '-o', 't', '--',
                                                     sys.executable, os.path.abspath(__file__)] + argv + child_flags
    try:
        r = subprocess.run(cmd, cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout)
    except Exception as e:              # noqa: BLE001 -- instrumentation must not take the bench down
        shutil.rmtree(d, ignore_errors=True)
        return None, f'{type(e).__name__}: {e}'
    if r.returncode != 0:
        shutil.rmtree(d, ignore_errors=True)
        return None, f'rocprofv3 child failed (rc {r.returncode}): {(r.stderr or "")[-300:]}'
    return d, 'ok'


def _marked_segment(rows):
    marks = [i for i, q in enumerate(rows) if kernel_base_name(q[2]) == MARK]
    if len(marks) < 2:
        return None
    return rows[marks[-2] + 1:marks[-1]]


def replayed_trace(argv, steps, mode):
    """Per-kernel durations of the graph-replayed timed region of `mode`'s leg: runs this same command
    (same box, same flags, --mark) under `rocprofv3 --kernel-trace` and keeps the kernels between the
    two marker launches = exactly the K timed steps.  -> ({kernel base name: [calls/step, ms/step]},
    note) or (None, reason)."""
    import csv
    import glob
    import shutil
    flags = ['--mark', '--no-roofline', '--no-cpu-baseline', '--big-batch', '0', '--mode', mode,
             '--steps', str(steps)]          # (the last occurrence of a flag wins)
    if mode == 'train':
        flags.append('--only-train')
    d, note = _rocprof_child([], argv, flags)
    if d is None:
        return None, note
    try:
        files = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)
        if not files:
            return None, 'no kernel trace written'
        rows = []
        with open(files[0]) as f:
            for q in csv.DictReader(f):
                rows.append((int(q['Start_Timestamp']), int(q['End_Timestamp']), q['Kernel_Name']))
        rows.sort()
        seg = _marked_segment(rows)
        if seg is None:
            return None, 'markers not found in the trace'
        agg = {}
        for a, b, k in seg:
            v = agg.setdefault(kernel_base_name(k), [0.0, 0.0])
            v[0] += 1.0 / steps
            v[1] += (b - a) / 1e6 / steps
        span = (seg[-1][1] - seg[0][0]) / 1e6 / steps if seg else 0.0
        agg['__span_ms_per_step__'] = [len(seg) / steps, span]
        return agg, 'rocprofv3 --kernel-trace of the graph-replayed timed region (child run of this command, same box)'
    except Exception as e:              # noqa: BLE001
        return None, f'{type(e).__name__}: {e}'
    finally:
        shutil.rmtree(d, ignore_errors=True)


# entry point -> the kernels of which each of its calls launches exactly ONE (second stages -- split-K epilogues, folds,
# the statistics pass of a two-pass GroupNorm -- are extra dispatches of the same call)
PRIMARY = {
    'sdmi_igemm': ('igemm_kernel', 'igemm_kernel_tall', 'igemm_dma_kernel', 'igemm_sym_kernel',
                   'igemm_halo_kernel', 'conv3x3_c64_kernel'),
    'sdmi_wgrad': ('wgrad_kernel', 'wgrad_tr_kernel', 'wgrad3x3_c64_kernel', 'wgrad3x3_halo_kernel'),
    'sdmi_groupnorm': ('gn_fused_kernel', 'gn_fused2_kernel', 'gn_apply_kernel'),
    'sdmi_groupnorm_bwd': ('gn_bwd_fused_kernel', 'gn_bwd_fused2_kernel', 'gn_bwd_apply_kernel'),
}


def _attribute_calls(log_path, rows, lo, hi, per_kernel):
    """Algorithmic bytes per kernel: the child's call log (slotdiffusion_amd/_lib.py: SDMI_CALL_LOG, launch order) against
    the dispatch order of the same pass -- the k-th call of an entry point is the k-th dispatch among that entry's
    primary kernels.  Adds `alg_bytes` / `alg_n` to per_kernel; returns a note when a sequence does not line up."""
    try:
        calls = [ln.rstrip('\n').split('\t') for ln in open(log_path)]
    except OSError as e:
        return f'call log: {e}'
    marks = [i for i, c in enumerate(calls) if c[0] == 'sdmi_sqerr_rows']
    if len(marks) < 2:
        return 'call log: markers not found'
    seg = calls[marks[-2] + 1:marks[-1]]
    disp = sorted({(did, kernel_base_name(k)) for did, k, _, _ in rows if lo < did < hi})
    bad = []
    for entry, prim in PRIMARY.items():
        ds = [k for _, k in disp if k in prim]
        cs = [float(c[1]) for c in seg if c[0] == entry]
        if len(ds) != len(cs):
            if ds or cs:
                bad.append(f'{entry}: {len(cs)} calls vs {len(ds)} primary dispatches')
            continue
        for k, b in zip(ds, cs):
            e = per_kernel.setdefault(k, {})
            e['alg_bytes'] = e.get('alg_bytes', 0.0) + b
            e['alg_n'] = e.get('alg_n', 0) + 1
    return ('call attribution skipped for ' + '; '.join(bad)) if bad else None


def write_pmc_table(path, pk, fs, wsc):
    """Per-kernel counters of the step between the markers as CSV: MFMA-busy fraction, observed HBM-side bytes per launch
    (calibrated FETCH_SIZE + WRITE_SIZE) against the algorithmic bytes of the launches that produced them."""
    names = sorted(pk, key=lambda k: -pk[k].get('GRBM_GUI_ACTIVE', 0.0))
    with open(path, 'w') as f:
        f.write('kernel,dispatches,GRBM_GUI_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES,mfma_busy_frac,FETCH_SIZE_KiB,WRITE_SIZE_KiB,'
                'observed_bytes_per_launch,algorithmic_bytes_per_launch,observed_over_algorithmic\n')
        for k in names:
            e = pk[k]
            n = max([len(e.get('_ids_' + c, ())) for c in ('FETCH_SIZE', 'WRITE_SIZE', 'GRBM_GUI_ACTIVE')] + [1])
            act = e.get('GRBM_GUI_ACTIVE', 0.0)
            busy = e.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
            obs = (fs * e.get('FETCH_SIZE', 0.0) + wsc * e.get('WRITE_SIZE', 0.0)) * 1024.0 / n
            alg = e['alg_bytes'] / e['alg_n'] if e.get('alg_n') else None
            f.write(f'"{k}",{n},{act:.6g},{busy:.6g},{(busy / (act / 8.0 * 1024.0)) if act else 0.0:.4f},'
                    f'{e.get("FETCH_SIZE", 0.0):.6g},{e.get("WRITE_SIZE", 0.0):.6g},{obs:.6g},'
                    f'{"" if alg is None else f"{alg:.6g}"},{"" if not alg else f"{obs / alg:.3f}"}\n')


def pmc_pass(argv, mode):
    """HBM traffic and MFMA busy fraction of the igemm family OBSERVED BY THIS RUN: two `rocprofv3
    --kernel-trace --pmc` child passes of this command (eager launches so that every dispatch is
    attributable, one timed step between the markers; counters in their own runs, MI355X_MICROARCH.md
    "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE cannot share a pass).
    Calibration, as the guide prescribes ("calibrate on a known byte count in your own access pattern"):
    the same passes see two kernels of known traffic in the step -- the gradient arena's zero fill
    (`fillBufferAligned`: n * 4 bytes written, nothing read) and `sqsum_kernel` (n * 4 bytes read) -- and
    the byte counters are scaled by known / counted for each of them (the guide's gfx950 note: FETCH_SIZE
    reads exactly 1/2 of a wide coalesced stream; WRITE_SIZE uncalibrated)."""
    import csv
    import glob
    import shutil
    flags = ['--mark', '--no-roofline', '--no-cpu-baseline', '--big-batch', '0', '--mode', mode, '--no-graph',
             '--steps', '1', '--warmup', '1']
    if mode == 'train':
        flags.append('--only-train')
    argv = [a for i, a in enumerate(argv) if a not in ('--steps', '--warmup') and
            (i == 0 or argv[i - 1] not in ('--steps', '--warmup'))]
    out = {'source': 'rocprofv3 --pmc child passes of this command on this box (this run; eager launches, 1 step)'}
    per_kernel = {}
    notes = []
    for counters in (['SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'FETCH_SIZE'], ['WRITE_SIZE']):
        first = counters[0] != 'WRITE_SIZE'
        d, note = _rocprof_child(['--pmc'] + counters, argv, flags,
                                 extra_env=({'SDMI_CALL_LOG': '%d/calls.tsv'} if first else None))
        if d is None:
            notes.append(f'{"+".join(counters)}: {note}')
            continue
        try:
            files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
            rows = []
            for path in files:
                with open(path) as f:
                    for q in csv.DictReader(f):
                        rows.append((int(q['Dispatch_Id']), q['Kernel_Name'], q['Counter_Name'],
                                     float(q['Counter_Value'])))
            rows.sort()
            ids = sorted({r[0] for r in rows if kernel_base_name(r[1]) == MARK})
            if len(ids) < 2:
                notes.append(f'{"+".join(counters)}: markers not found')
                continue
            lo, hi = ids[-2], ids[-1]
            for did, k, c, v in rows:
                if lo < did < hi:
                    e = per_kernel.setdefault(kernel_base_name(k), {})
                    e[c] = e.get(c, 0.0) + v
                    e['_max_' + c] = max(e.get('_max_' + c, 0.0), v)      # largest single dispatch
                    e.setdefault('_ids_' + c, set()).add(did)
            if first:
                note = _attribute_calls(os.path.join(d, 'calls.tsv'), rows, lo, hi, per_kernel)
                if note:
                    notes.append(note)
        except Exception as e:          # noqa: BLE001
            notes.append(f'{"+".join(counters)}: {type(e).__name__}: {e}')
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if notes:
        out['notes'] = notes
    out['per_kernel'] = per_kernel
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=0,
                    help='images (clips for the video configs) per GPU; default 64 / 16 clips / coco224: 16')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32', 'fp8'],
                    help="fp8 = bf16 storage + e4m3fn operands on the denoiser's 3x3 convolutions "
                         '(train step: the forward GEMMs of those layers, scales derived on the device each step)')
    ap.add_argument('--config', default='clevrtex128',
                    choices=['clevrtex128', 'coco224', 'movid11x6', 'movie15x6'],
                    help='clevrtex128 = BASELINE configs[1] (the metric); movid11x6 / movie15x6 = configs[2] / '
                         '[3] (video, 6-frame clips); coco224 = configs[4]')
    ap.add_argument('--mode', default='train', choices=['train', 'sample'])
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 --pmc child passes (traffic = null)')
    ap.add_argument('--pmc-table', default='', help='write the per-kernel counters of the PMC passes here as CSV '
                    '(%%m = train / sample): MFMA busy, observed vs algorithmic bytes per launch')
    ap.add_argument('--big-batch', type=int, default=256,
                    help='extra sampling measurement at this batch (0 = off): the chip is far from '
                         'full at the configured B = 64')
    ap.add_argument('--only-train', action='store_true',
                    help='profiling aid: skip the sampling leg of --mode train')
    ap.add_argument('--mark', action='store_true',
                    help='internal (replayed_trace): bracket the timed region with marker kernels')
    args = ap.parse_args()

    # --gpus N is the number of ranks (one per GPU).  Started without a launcher it re-executes
    # itself under `python -m torch.distributed.run --nproc-per-node N` (the reference's launch:
    # torch.distributed.launch --nproc_per_node=$GPUS, scripts/sbatch_run.sh:36-39); a mismatch
    # between --gpus and the world the launcher set up, or fewer visible devices than ranks, is an
    # error -- never a silent single-GPU measurement.
    n_vis = torch.cuda.device_count()
    if args.gpus > n_vis:
        sys.exit(f'bench.py: --gpus {args.gpus} requested but only {n_vis} GPU(s) are visible on this box')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        from slotdiffusion_amd import parallel as _par
        _par.respawn_under_launcher(args.gpus, os.path.abspath(__file__), sys.argv[1:])
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    dist = None
    if rank != 0:
        # only rank 0 owns stdout (the result line); native banners of the other ranks (RCCL prints
        # its version block to stdout) must not trail it
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    # SDMI_BENCH_FORCE_DIST=1: run the multi-GPU code path (RCCL communicator, split backward with
    # overlapped all-reduce, barriers, MAX-over-ranks timing) with a single rank -- a self-test of
    # that path on a 1-GPU box
    if world > 1 or os.environ.get('SDMI_BENCH_FORCE_DIST'):
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    dtype = torch.float32 if args.dtype == 'fp32' else torch.bfloat16

    from slotdiffusion_amd import configs, ops, parallel
    bc = configs.BENCH_CONFIGS[args.config]
    model, cfg, frames = build_model(dtype, config=args.config)
    model = model.to(dev)
    if args.dtype == 'fp8':
        model.set_compute_dtype('fp8')
    model.use_graph = not args.no_graph
    res = cfg['resolution'][0]
    lat = res // 4
    B = args.batch if args.batch > 0 else bc['batch']
    n_img = B * (frames or 1)                 # images per step and GPU (a clip counts its frames)
    img = synth_batch(B, rank, dev, res, frames)
    # gradient buckets on a bf16 wire when the model computes in bf16: an explicit opt-in of this benchmark (the library
    # default is fp32 like the reference's DDP), SDMI_GRAD_BF16=0 / 1 overrides
    wire = parallel.resolve_wire('bf16' if dtype == torch.bfloat16 else 'fp32')
    if dist is not None:                      # every rank starts from rank 0's parameters
        parallel.broadcast_parameters(model.arena())
        model.weights_updated()
    from slotdiffusion_amd.optim import FusedAdam
    nfe = 20

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def mark():
        if args.mark:           # (outside the timed region: before t0 / after the closing barrier)
            from slotdiffusion_amd import metrics
            metrics._sqerr(torch.zeros(1, 8, device=dev), torch.ones(1, 8, device=dev))

    def timed(fn, steps, warmup, marked=False):
        for _ in range(warmup):
            fn()
        if marked:
            mark()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        if marked:
            mark()
        if dist is not None:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt)
        return dt

    # ---- sampling leg: 20-NFE DPM-Solver++ over the batch's slots ------------------------
    model.eval()
    with torch.no_grad():
        slots, _ = model.encode(img)
        if frames:
            slots = slots.flatten(0, 1).contiguous()          # every frame's slots condition one latent
        g = torch.Generator(device='cpu').manual_seed(77 + rank)
        x_T = ops.nchw_to_nhwc(torch.randn(n_img, 3, lat, lat, generator=g).to(dev), torch.float32, 4)

        def sample_step():
            return model._dpm_sample(x_T, slots)[0]
        skip_sample = args.only_train and args.mode == 'train'
        n_s = args.steps if args.mode == 'sample' else max(2, args.steps // 2)
        dt_s = timed((lambda: None) if skip_sample else sample_step, n_s,
                     args.warmup if args.mode == 'sample' else 1, marked=args.mode == 'sample')
        denoise_rate = world * n_img * nfe * n_s / dt_s
        # the timed configuration's result must at least be finite: final latents of one more pass
        sample_finite = None
        if not skip_sample:
            z_fin = sample_step()
            sample_finite = bool(torch.isfinite(z_fin.float()).all())
            sample_absmax = float(z_fin.float().abs().max())
        big_rate = None
        if args.big_batch and args.big_batch != n_img and world == 1 and not args.only_train and \
                args.config == 'clevrtex128' and args.mode == 'train':
            # informational: the same sampler at a batch that fills the chip better
            Bb = args.big_batch
            slots_b = slots[:1].expand(Bb, -1, -1).contiguous() + 0.01 * torch.randn(
                Bb, slots.shape[1], slots.shape[2], device=dev)
            xT_b = ops.nchw_to_nhwc(torch.randn(Bb, 3, lat, lat, device=dev), torch.float32, 4)
            big = lambda: model._dpm_sample(xT_b, slots_b)[0]
            dt_b = timed(big, 2, 1)
            big_rate = Bb * nfe * 2 / dt_b
            model._graph_cache.clear()
            del slots_b, xT_b

    # ---- training leg: forward + loss + backward (+ DDP all-reduce) + clip + Adam --------
    model.train()
    opt = FusedAdam(model, lr=1e-4, dec_lr=2e-4, clip_grad=bc['clip_grad'], total_steps=100000)
    garena = model.grad_arena()
    reducer = parallel.GradReducer(garena, world, wire) if dist is not None else None

    def train_step(reduce=True):
        opt.zero_grad()
        out = model(dict(img=img))
        loss = model.calc_train_loss(dict(img=img), out)['denoise_loss']
        loss.backward()
        if dist is not None and reduce:          # gradients only: the flat arena in a few large buckets
            r = reducer.reduce_all(4)
            opt.step(grad_src=r.grad_src, grad_scale=r.grad_scale)
        else:
            opt.step()
        return loss

    train_rate = None
    dt_t = None
    train_loss = None
    params_finite = None
    comm = None
    if args.mode == 'train':
        run_step = train_step
        if not args.no_graph:
            from slotdiffusion_amd.optim import GraphedTrainStep
            # world > 1: backward split at the slots, denoiser gradients all-reduced under the
            # encoder's backward (optim.GraphedTrainStep)
            graphed = GraphedTrainStep(model, opt, dict(img=img),
                                       allreduce=(True if dist is not None else None), world=world, wire=wire)
            run_step = lambda: graphed(dict(img=img))
        dt_t = timed(run_step, args.steps, args.warmup, marked=True)
        train_rate = world * n_img * args.steps / dt_t
        # loss of the last timed step (the graphed step keeps it in a static tensor) + a finite check of the
        # updated parameters: a fast step that produces NaN is not a measurement
        last = run_step()
        train_loss = float(graphed.loss if not args.no_graph else last)
        params_finite = bool(torch.isfinite(model.arena()).all())
        if dist is not None:
            # the exchange by itself (all ranks idle otherwise) and what of it the step exposes:
            # the same graphed step without the collective, timed the same way
            def only_reduce():
                reducer.reduce_all(4)
            dt_ar = timed(only_reduce, 3, 1)
            dt_nored = None
            if not args.no_graph:
                nored = GraphedTrainStep(model, opt, dict(img=img), allreduce=None, world=world)
                dt_nored = timed(lambda: nored(dict(img=img)), args.steps, 1)
            comm = {'rccl_ranks': dist.get_world_size(), 'backend': dist.get_backend(),
                    'devices': [f'cuda:{i}' for i in range(world)],
                    'grad_dtype_on_wire': wire,
                    'grad_bytes_per_step': garena.numel() * (2 if wire == 'bf16' else 4),
                    'all_reduce_ms': 1e3 * dt_ar / 3,
                    'exposed_comm_ms': (1e3 * (dt_t - dt_nored) / args.steps) if dt_nored else None}

    what = bc['baseline']
    shape = f'{res}x{res}' + (f' x {frames}-frame clips' if frames else '')
    slots_n = cfg['slot_dict']['num_slots']
    if args.mode == 'train':
        value, unit, ms = train_rate, 'images/s', 1e3 * dt_t / args.steps
        metric = f'train-step images/sec, {shape} {slots_n}-slot (DPM-Solver denoise-steps/sec in `denoise`)'
        work = (f'BASELINE {what}: train step = slot encoder + frozen VQ-VAE encode + q-sample + UNet eps + '
                f'MSE, backward, clip, Adam (dropout 0.1)')
    else:
        value, unit, ms = denoise_rate, 'image-denoise-steps/s', 1e3 * dt_s / n_s
        metric = f'DPM-Solver denoise-steps/sec, {shape} {slots_n}-slot'
        work = f'BASELINE {what}: 20-NFE DPM-Solver++ sampling (UNet eps + VQ per NFE)'
    if args.dtype == 'fp8' and args.mode == 'train':
        work += ("; e4m3fn operands in the FORWARD 3x3 convolutions of the UNet (fp8 MFMA; weights re-quantised on the "
                 "device every step at 448 / amax, activations at a fixed scale), bf16 backward and elsewhere")
    elif args.dtype == 'fp8':
        work += "; e4m3fn operands on the UNet's 3x3 convolutions (fp8 MFMA), bf16 elsewhere"
    out = {
        'metric': metric, 'value': value, 'unit': unit, 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': work, 'name': args.config, 'batch_per_gpu': B,
                   'images_per_step_per_gpu': n_img, 'frames_per_clip': frames,
                   'hip_graph': not args.no_graph, 'parallelism': f'dp{world}'},
        'denoise_big_batch': ({'value': big_rate, 'unit': 'image-denoise-steps/s',
                               'batch': args.big_batch} if big_rate else None),
        'denoise': {'value': denoise_rate, 'unit': 'image-denoise-steps/s',
                    'ms_per_20nfe_pass': 1e3 * dt_s / n_s, 'nfe': nfe, 'images': n_img},
        'comm': comm,
    }
    if skip_sample:
        out['denoise'] = None
    else:
        out['denoise']['finite'] = sample_finite
        out['denoise']['latent_absmax'] = sample_absmax
    if args.mode == 'train':
        out['loss'] = train_loss
        out['finite'] = bool(params_finite and train_loss == train_loss and abs(train_loss) != float('inf'))
    else:
        out['finite'] = sample_finite

    # ---- rank-0-only instrumentation: no collective inside -------------------------------
    argv0 = [a for a in sys.argv[1:] if a != '--mark']
    argv0 = [a for i, a in enumerate(argv0) if a != '--mode' and (i == 0 or argv0[i - 1] != '--mode')]

    def roofline_of(mode, step_fn, steps, ms_step, want_hbm):
        """Roofline objects of one leg (see the module docstring)."""
        from slotdiffusion_amd._lib import KernelTimer
        model.use_graph = False
        overlap, model.bank().overlap_wgrad = model.bank().overlap_wgrad, False  # clean timings
        with torch.set_grad_enabled(mode == 'train'):
            step_fn()
            with KernelTimer() as kt:
                step_fn()
            summ = kt.summary()       # launch counts + algorithmic flops / bytes per entry point
        model.bank().overlap_wgrad = overlap
        model.use_graph = not args.no_graph
        # (fp8 train step: only the forward 3x3 convolutions of the denoiser multiply e4m3fn operands -- the backward
        #  pass and every other GEMM are bf16 -- so the leg is priced against the bf16 peak)
        peak = PEAK_TFLOPS['bf16' if (args.dtype == 'fp8' and mode == 'train') else args.dtype]
        # durations: the graph-replayed timed region itself (rocprofv3 child run on this box);
        # eager HIP events only as the labelled fallback
        trace, tnote = (None, 'not taken (--no-graph or N > 1)') if (args.no_graph or world > 1) else \
            replayed_trace(argv0, steps, mode)

        def fam_ms(entry):
            if trace is None:
                return summ.get(entry, {}).get('ms', 0.0), summ.get(entry, {}).get('calls', 0)
            t_ms = sum(trace[k][1] for k in FAMILIES[entry] if k in trace)
            n = sum(trace[k][0] for k in FAMILIES[entry] if k in trace)
            return t_ms, n
        # the igemm family = sdmi_igemm launches + the fused data / weight gradient launches (sdmi_bwd_pair:
        # igemm's tile body + the weight-gradient body in one kernel; its flops count both gradients)
        pr = summ.get('sdmi_bwd_pair', {})
        # fused SpatialTransformer blocks (inference: sdmi_st_block; training: sdmi_st_train_fwd / _bwd): their GEMMs + the
        # attention contractions inside them
        stb = [summ.get(e, {}) for e in ('sdmi_st_block', 'sdmi_st_train_fwd', 'sdmi_st_train_bwd')]
        ig = {k: summ['sdmi_igemm'].get(k, 0.0) + pr.get(k, 0.0) + sum(x.get(k, 0.0) for x in stb)
              for k in ('calls', 'ms', 'flops', 'bytes')}
        ig_flops = ig['flops']
        ig_ms, ig_kernels = fam_ms('sdmi_igemm')
        if trace is None:
            ig_ms = ig['ms']
        wg = {k: summ.get('sdmi_wgrad', {}).get(k, 0.0) + summ.get('sdmi_wgrad_group', {}).get(k, 0.0)
              for k in ('calls', 'ms', 'flops', 'bytes')}      # stand-alone and grouped weight gradients
        wg = wg if wg['calls'] else {}
        wg_ms, wg_kernels = fam_ms('sdmi_wgrad') if wg else (0.0, 0)
        if trace is None and wg:
            wg_ms = wg['ms']
        fam_flops = ig_flops + wg.get('flops', 0.0)
        fam_time = ig_ms + wg_ms
        # Round 5: the family is every GEMM-shaped launch of the convolution / linear layers -- forward, data gradient AND
        # weight gradient.  Rounds 1 - 4 quoted the igemm family alone, which held the weight gradients of the layers
        # that ran them inside sdmi_bwd_pair; the 3x3 layers that now leave the pair launch (igemm_halo / wgrad3x3_halo
        # kernels) would otherwise move their weight-gradient flops out of the family while their stand-alone launches
        # still overlap (and stretch) the family's kernels.  `igemm_only` keeps the old definition for comparison.
        ach = fam_flops / (fam_time * 1e-3) / 1e12
        rf = {
            'bound': 'mfma', 'kernel': 'GEMM family (sdmi_igemm: implicit-GEMM conv/linear fwd + dgrad; sdmi_bwd_pair: '
                                       'dgrad + wgrad of a layer in one launch; sdmi_st_block / sdmi_st_train_fwd / _bwd: fused SpatialTransformer blocks; '
                                       'sdmi_wgrad / sdmi_wgrad_group: stand-alone and grouped weight gradients): '
                                       + ', '.join(FAMILIES['sdmi_igemm'] + FAMILIES['sdmi_wgrad']),
            'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
            'igemm_only': {'achieved': ig_flops / (ig_ms * 1e-3) / 1e12, 'frac': ig_flops / (ig_ms * 1e-3) / 1e12 / peak,
                           'family_ms_per_step': ig_ms, 'algorithmic_gflop_per_step': ig_flops / 1e9},
            'timing_source': tnote if trace is not None else f'eager HIP-event pass (fallback: {tnote})',
            'driver_observed': trace is not None,
            'launches_per_step': ig['calls'] + wg.get('calls', 0), 'kernels_per_step': ig_kernels + wg_kernels,
            'family_ms_per_step': fam_time, 'avg_launch_us': 1e3 * fam_time / (ig['calls'] + wg.get('calls', 0)),
            'algorithmic_gflop_per_launch': fam_flops / (ig['calls'] + wg.get('calls', 0)) / 1e9,
            'algorithmic_gflop_per_step': fam_flops / 1e9,
            'algorithmic_bytes_per_launch': (ig['bytes'] + wg.get('bytes', 0.0)) / (ig['calls'] + wg.get('calls', 0)),
            'traffic': None, 'traffic_unit': 'HBM bytes per launch', 'traffic_source': None,
            'mfma_util_pmc': None,
            'gemm_family_tflops_incl_wgrad': fam_flops / (fam_time * 1e-3) / 1e12,
            'gemm_family_gflop_per_step_incl_wgrad': fam_flops / 1e9,
            'whole_step_tflops': fam_flops / (ms_step * 1e-3) / 1e12,
            'whole_step_frac': fam_flops / (ms_step * 1e-3) / 1e12 / peak,
            'eager_event_pass': {'igemm_ms': ig['ms'], 'tflops': ig_flops / (ig['ms'] * 1e-3) / 1e12,
                                 'all_kernels_ms': sum(v['ms'] for v in summ.values())}}
        if trace is not None:
            rf['kernels_in_timed_region_per_step'] = trace['__span_ms_per_step__'][0]
            rf['kernel_span_ms_per_step'] = trace['__span_ms_per_step__'][1]
        hbm = None
        if want_hbm:
            # the HBM-bound family that costs the most time: GroupNorm (+SiLU/residual/dropout) passes
            hb = {}
            for entry in ('sdmi_groupnorm', 'sdmi_groupnorm_bwd'):
                if entry in summ and summ[entry]['bytes']:
                    t_ms, n_k = fam_ms(entry)
                    if not t_ms:        # no kernel of the family matched the trace (renamed kernel): say nothing
                        continue
                    gbs = summ[entry]['bytes'] / (t_ms * 1e-3) / 1e9
                    hb[entry] = {'achieved': gbs, 'frac': gbs / 8000.0, 'launches_per_step': summ[entry]['calls'],
                                 'family_ms_per_step': t_ms,
                                 'algorithmic_bytes_per_launch': summ[entry]['bytes'] / summ[entry]['calls'],
                                 'avg_launch_us': 1e3 * t_ms / summ[entry]['calls']}
            if hb:
                worst = min(hb, key=lambda k: hb[k]['frac'])
                hbm = dict(hb[worst], bound='hbm', peak=8000.0, unit='GB/s',
                           kernel=f'{worst}: ' + ', '.join(FAMILIES[worst]),
                           timing_source=rf['timing_source'], traffic=None, families=hb)
        return rf, hbm, summ

    def apply_pmc(rf, mode, n_known):
        """Fill traffic / mfma_util_pmc of a roofline object from this run's PMC child passes."""
        pm = pmc_pass(argv0, mode)
        pk = pm.get('per_kernel', {})
        fam = [pk[k] for k in FAMILIES['sdmi_igemm'] + FAMILIES['sdmi_wgrad'] if k in pk]
        if not fam:
            rf['traffic_source'] = 'PMC passes unusable on this box: ' + '; '.join(pm.get('notes', ['no igemm dispatches seen']))
            return
        tot = lambda c: sum(e.get(c, 0.0) for e in fam)
        n_disp = max(sum(len(e.get('_ids_' + c, ())) for e in fam) for c in ('FETCH_SIZE', 'WRITE_SIZE', 'GRBM_GUI_ACTIVE'))
        cal = {}
        # known-bytes kernels of the same step (units of the counters: KiB)
        fill = pk.get('__amd_rocclr_fillBufferAligned')
        sq = pk.get('sqsum_kernel')
        # (the step has several fills / one sqsum: the known-size one is the largest dispatch)
        if fill and fill.get('_max_WRITE_SIZE') and n_known:
            cal['write_scale'] = (n_known * 4.0 / 1024.0) / fill['_max_WRITE_SIZE']
        if sq and sq.get('_max_FETCH_SIZE') and n_known:
            cal['fetch_scale'] = (n_known * 4.0 / 1024.0) / sq['_max_FETCH_SIZE']
        fs = cal.get('fetch_scale') or 2.0       # guide: FETCH_SIZE reads 1/2 of a wide coalesced stream
        wsc = cal.get('write_scale') or 1.0
        if tot('FETCH_SIZE') or tot('WRITE_SIZE'):
            rf['traffic'] = (fs * tot('FETCH_SIZE') + wsc * tot('WRITE_SIZE')) * 1024.0 / max(1, n_disp)
        if tot('GRBM_GUI_ACTIVE'):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs
            rf['mfma_util_pmc'] = tot('SQ_VALU_MFMA_BUSY_CYCLES') / (tot('GRBM_GUI_ACTIVE') / 8.0 * 1024.0)
        rf['traffic_source'] = pm['source'] + (
            f"; FETCH_SIZE x {fs:.3f} ({'calibrated on sqsum_kernel, ' + str(n_known * 4) + ' bytes read' if 'fetch_scale' in cal else 'guide default: 128-byte requests tallied at 64 B'})"
            f", WRITE_SIZE x {wsc:.3f} ({'calibrated on the arena zero fill, ' + str(n_known * 4) + ' bytes written' if 'write_scale' in cal else 'uncalibrated'})"
            + (' -- ' + '; '.join(pm['notes']) if pm.get('notes') else ''))
        if args.pmc_table:
            try:
                write_pmc_table(args.pmc_table.replace('%m', mode), pk, fs, wsc)
            except OSError as e:
                rf['traffic_source'] += f' -- table not written: {e}'
        rf['traffic_dispatches'] = n_disp
        rf['traffic_ratio_to_algorithmic'] = (rf['traffic'] / rf['algorithmic_bytes_per_launch']) if rf['traffic'] else None

    if rank == 0 and not args.no_roofline:
        if args.mode == 'train':
            rf, hbm, summ = roofline_of('train', lambda: train_step(reduce=False), args.steps, ms, True)
            if world == 1 and not args.no_graph and not args.no_pmc:
                apply_pmc(rf, 'train', garena.numel())
            out['roofline'] = rf
            if hbm:
                out['roofline_hbm'] = hbm
            out['kernel_breakdown_ms'] = {k: round(v['ms'], 3) for k, v in
                                          sorted(summ.items(), key=lambda kv: -kv[1]['ms'])}
            if not skip_sample:
                model.eval()
                rfs, _, _ = roofline_of('sample', sample_step, n_s, 1e3 * dt_s / n_s, False)
                if world == 1 and not args.no_graph and not args.no_pmc:
                    apply_pmc(rfs, 'sample', 0)         # (the sampling leg's counters, this run)
                model.train()
                out['denoise']['roofline'] = rfs
        else:
            model.eval()
            rf, hbm, summ = roofline_of('sample', sample_step, n_s, ms, True)
            if world == 1 and not args.no_graph and not args.no_pmc:
                apply_pmc(rf, 'sample', 0)
            out['roofline'] = rf
            if hbm:
                out['roofline_hbm'] = hbm
            out['kernel_breakdown_ms'] = {k: round(v['ms'], 3) for k, v in
                                          sorted(summ.items(), key=lambda kv: -kv[1]['ms'])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(model, cfg, args.mode) if args.config == 'clevrtex128' else None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: flush whatever native libraries (the RCCL
        # version banner) still hold in C stdio buffers first
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)       # (normal interpreter exit: profilers finalise at exit)
        if out.get('finite') is False or (out.get('denoise') or {}).get('finite') is False:
            sys.stderr.write('bench.py: non-finite loss / parameters / latents in the timed configuration\n')
            sys.exit(3)


if __name__ == '__main__':
    main()
