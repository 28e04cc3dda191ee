import os, sys, torch
sys.path.insert(0, '/root/repo')
from tests import common as C
from tests.detfill import det_fill_, is_buffer_name
from slotdiffusion_amd.models import SADiffusion
G = C.load_golden()
img = C.make_inputs(2)[0].cuda()
res = {}
for defer in ('0', '1'):
    os.environ['SDMI_DEFER_COLSUM'] = defer
    cfg = C.clevrtex_cfg()
    m = SADiffusion(cfg['resolution'], cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'], cfg['loss_dict'], compute_dtype=torch.float32)
    det_fill_(m.state_dict().items(), skip=is_buffer_name)
    m.train_dropout = 0.0
    m = m.cuda().train()
    m.grad_arena().zero_()
    out = m(dict(img=img))
    loss = m.calc_train_loss(dict(img=img, t=G['t'].cuda(), noise=G['noise'].cuda()), out)['denoise_loss']
    loss.backward()
    torch.cuda.synchronize()
    res[defer] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
bad = []
for n in res['0']:
    a, b = res['0'][n], res['1'][n]
    d = float((a - b).norm() / (a.norm() + 1e-20))
    if d > 1e-4:
        bad.append((d, n, float(a.norm()), float(b.norm())))
print(len(bad), 'tensors differ of', len(res['0']))
for d, n, na, nb in sorted(bad, reverse=True)[:40]:
    print(f'{d:.3e} {n} {na:.4e} {nb:.4e}')
