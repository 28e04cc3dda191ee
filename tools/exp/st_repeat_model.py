"""Run-to-run repeatability of the fused SpatialTransformer training kernels at the benchmark's shapes (B = 64): forward
(every stored tensor) and backward (input / slot / parameter gradients) N times on fixed inputs."""
import sys
import torch
sys.path.insert(0, '.')
from tests.test_gpu_st_train import _model, _inputs
from slotdiffusion_amd import kern

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
m = _model(seed=3)
wb = m.KG().wb
ga = m.grad_arena()
for name, hw, B, slots, rows in (('input_blocks.4.1', 16, 64, 7, 64), ('input_blocks.7.1', 8, 64, 7, 32),
                                 ('output_blocks.8.1', 16, 64, 15, 64), ('output_blocks.5.1', 8, 64, 11, 32)):
    u, n, heads, Cc, x, kv = _inputs(m, name, hw, B, slots)
    dout = torch.randn(x.shape, generator=torch.Generator().manual_seed(1)).bfloat16().cuda()
    first, bad_f, bad_b = None, 0, 0
    for i in range(N):
        with torch.no_grad():
            out, sv = kern.StBlockFn.run_forward(wb, x, kv, n, heads, rows)
            ga.zero_()
            wb._join_queued = True
            dx, dkv = kern.StBlockFn.backward_fused(wb, n, heads, x, kv, sv, dout, rows)
            wb.join()
        torch.cuda.synchronize()
        cur_f = dict(sv, out=out)
        cur_b = dict(dx=dx, dkv=dkv, params=ga.clone())
        if first is None:
            first = ({k: v.clone() for k, v in cur_f.items()}, {k: v.clone() for k, v in cur_b.items()})
            continue
        bf = [k for k in cur_f if not torch.equal(cur_f[k], first[0][k])]
        bb = [k for k in cur_b if not torch.equal(cur_b[k], first[1][k])]
        bad_f += bool(bf)
        bad_b += bool(bb)
        if bf or bb:
            print('   differs:', bf, bb, flush=True)
    print(f'{name} C={Cc} S={hw * hw} B={B} slots={slots} rows={rows}: forward {bad_f}/{N - 1}, backward {bad_b}/{N - 1} runs differ',
          flush=True)
