import sys, torch
sys.path.insert(0, '/root/repo')
from tests import common as C
from tests.detfill import det_fill_, is_buffer_name
from slotdiffusion_amd.models import SAViDiffusion
cfg = C.movie_cfg(); T=3; seed=11
G = C.load_golden('savidiff_b1t3.npz')
m = SAViDiffusion(cfg['resolution'], T, cfg['slot_dict'], cfg['enc_dict'], cfg['dec_dict'], cfg['pred_dict'], cfg['loss_dict'], compute_dtype=torch.float32)
det_fill_(m.state_dict().items(), skip=is_buffer_name)
m.train_dropout = m.pred_dropout = 0.0
m = m.cuda()
img = C.make_inputs(T, seed=seed)[0].view(1, T, 3, 128, 128).cuda()
m.train(); m.grad_arena().zero_()
out = m(dict(img=img))
loss = m.calc_train_loss(dict(img=img, t=G['t'].cuda(), noise=G['noise'].cuda()), out)['denoise_loss']
loss.backward()
named = dict(m.named_parameters())
names = [str(n) for n in G['grad_norms_names']]
mine = torch.tensor([float(named[n].grad.norm()) for n in names])
ref = G['grad_norms']
rel = ((mine - ref).abs() / (ref.abs() + 1e-12))
idx = torch.argsort(rel, descending=True)[:25]
for i in idx: print(names[i], float(mine[i]), float(ref[i]), float(rel[i]))
