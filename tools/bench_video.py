"""Train-step and sampling throughput of the VIDEO model (BASELINE.json configs 3 / 4:
video_based SAViDiffusion, MOVi 128x128x6-frame clips, 11 / 15 slots, bf16) on one MI355X.
Not the driver's bench line (bench.py measures config 1); same method: HIP-graph replayed step,
synthetic clips resident in HBM, K timed steps after W warm-up steps.

    python tools/bench_video.py [--clips 16] [--frames 6] [--slots 15] [--steps 5] [--warmup 2]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from slotdiffusion_amd.models import SAViDiffusion          # noqa: E402
from slotdiffusion_amd.optim import FusedAdam, GraphedTrainStep   # noqa: E402
from tests.common import movie_cfg                          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--clips', type=int, default=16)
    ap.add_argument('--frames', type=int, default=6)
    ap.add_argument('--slots', type=int, default=15)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    a = ap.parse_args()
    cfg = movie_cfg()
    cfg['slot_dict']['num_slots'] = a.slots
    m = SAViDiffusion(cfg['resolution'], a.frames, cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'],
                      cfg['pred_dict'], cfg['loss_dict'], compute_dtype=torch.bfloat16, seed=1234)
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        init = {s.name: s.init for s in m._spec}
        for n, p in m.named_parameters():
            if init[n] == 'zlin' and p.dim() > 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
    m = m.cuda().train()
    img = (torch.randn(a.clips, a.frames, 3, 128, 128, generator=g) * 0.5).clamp(-1, 1).cuda()
    opt = FusedAdam(m, lr=1e-4, dec_lr=2e-4, clip_grad=0.05, total_steps=100000)
    step = GraphedTrainStep(m, opt, dict(img=img))

    def timed(fn, k, w):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    dt = timed(lambda: step(dict(img=img)), a.steps, a.warmup)
    n_img = a.clips * a.frames
    out = {'workload': f'video_based SAViDiffusion train step, {a.clips} clips x {a.frames} frames, '
                       f'{a.slots} slots, 128x128, bf16, HIP graph',
           'images_per_s': n_img * a.steps / dt, 'ms_per_step': 1e3 * dt / a.steps,
           'images_per_step': n_img}
    # sampling: slots of the clip's frames -> 20-NFE DPM-Solver++ over clips*frames latents
    m.eval()
    with torch.no_grad():
        slots, _ = m.encode(img)
        cond = slots.flatten(0, 1)
        fn = lambda: m.dm_decoder.generate_imgs(cond, batch_size=n_img, same_noise=True)
        dts = timed(fn, max(1, a.steps // 2), 1)
    out['denoise_steps_per_s'] = n_img * 20 * max(1, a.steps // 2) / dts
    print(json.dumps(out))


if __name__ == '__main__':
    main()
