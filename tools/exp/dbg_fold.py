"""Run-to-run determinism of the train backward (GPU box): two identical passes must give bit-identical arenas."""
import sys, os, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..')
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import common as C
from test_gpu_train import _model, _train_backward
G = C.load_golden()
img = C.make_inputs(2)[0].cuda()
arenas = []
for defer in (False, False, True):
    m = _model(torch.bfloat16)
    m.bank().defer_fold = defer
    _train_backward(m, G, img)
    torch.cuda.synchronize()
    arenas.append(m.grad_arena().clone())
for a, b in ((0, 1), (0, 2)):
    d = (arenas[a] - arenas[b]).abs()
    print(a, b, 'max diff', float(d.max()), 'n diff', int((d > 0).sum()), 'of', d.numel())
