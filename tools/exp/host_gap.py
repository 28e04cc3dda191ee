"""Where a replayed train step loses 0.43 ms of GPU time at its head (profiles/r06_train_step_launch_list.txt: idle gaps of
44 us / 382 us behind the two eager copies in front of the graph): step period with / without the eager copies."""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from slotdiffusion_amd.optim import FusedAdam, GraphedTrainStep

B = 64
m, cfg, _ = bench.build_model(torch.bfloat16)
m = m.cuda()
m.train()
img = bench.synth_batch(B, 0, 'cuda')
opt = FusedAdam(m, lr=1e-4, dec_lr=2e-4, clip_grad=1.0, total_steps=100000)
step = GraphedTrainStep(m, opt, dict(img=img))
batch = dict(img=img)


def run(copy, lr, n=40):
    for _ in range(3):
        step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if copy:
            for k, v in batch.items():
                step.static[k].copy_(v, non_blocking=True)
        if lr:
            opt.set_lr_for_next_step()
        opt.step_count += 1
        step.g_fb.replay()
        step.g_up.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for rep in range(3):
    print('copy+lr %.3f   lr only %.3f   copy only %.3f   neither %.3f ms' %
          (run(True, True), run(False, True), run(True, False), run(False, False)))
