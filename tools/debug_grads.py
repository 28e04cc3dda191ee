import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from tests import common as C
from tests.test_gpu_train import _model, _train_backward
G = C.load_golden()
img = C.make_inputs(2)[0].cuda()
m = _model(torch.float32)
loss = _train_backward(m, G, img)
named = dict(m.named_parameters())
names = [str(n) for n in G['grad_norms_names']]
ref = G['grad_norms']
bad = []
for n, r in zip(names, ref):
    g = float(named[n].grad.norm())
    rel = abs(g - float(r)) / (float(r) + 1e-12)
    if rel > 1e-3:
        bad.append((n, g, float(r)))
print(len(bad), 'bad of', len(names))
import collections
cat = collections.Counter()
for n, g, r in bad:
    key = n.split('.')[-2] + '.' + n.split('.')[-1]
    cat[(key, 'zero' if g == 0 else 'wrong')] += 1
for k, v in sorted(cat.items()): print(k, v)
for n, g, r in bad[:60]: print(f'{n:90s} {g:.5e} {r:.5e} {abs(g-r)/r:.2e}')
for k in G:
    if k.startswith('grad:'):
        gg = named[k[5:]].grad.float().cpu(); rr = G[k]
        print(k, float((gg-rr).abs().max()/rr.abs().max()), float(torch.nn.functional.cosine_similarity(gg.flatten(), rr.flatten(), dim=0)))
