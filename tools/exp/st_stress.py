"""Failure rate of the fused block: many runs, count (image, 64-row block, wave slice) tiles that deviate."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
src = open(os.path.join(os.path.dirname(__file__), 'st_check.py')).read().split("gen = torch.Generator")[0]
sys.argv = [sys.argv[0], str(B)]
exec(compile(src, 'st_check_head', 'exec'))
gen = torch.Generator(device=dev).manual_seed(5)
for name, hw in (('input_blocks.4.1', 16), ('input_blocks.7.1', 8)):
    n = P + name
    heads = u.heads_of[name]
    C = heads * 32
    x = torch.randn(B, hw, hw, C, device=dev, generator=gen).bfloat16()
    slots = torch.randn(B, 7, 192, device=dev, generator=gen)
    with torch.no_grad():
        ctx = model._ctx(slots)
        t = n + '.transformer_blocks.0'
        kv = K.linear_multi(ctx, [(t + '.attn2.to_k.weight', t + '.attn2.to_v.weight')])[0]
        fold = K.cross_prepare(kv, t, heads)
        kvp = {'kv': kv, 'fold': fold}
        ref = ref_block(n, x, ctx, heads)
        S = hw * hw
        bad = tot = 0
        outs = []
        for rep in range(REPS):
            out = K.st_fused(x, n, heads, kvp)
            torch.cuda.synchronize()
            d = (out.float() - ref).reshape(B, S // 64, 64, 8, C // 8)
            r = ref.reshape(B, S // 64, 64, 8, C // 8)
            e = (d.pow(2).sum((2, 4)) / r.pow(2).sum((2, 4))).sqrt()      # [B, blocks, 8 waves]
            bad += int((e > 6e-3).sum())
            tot += e.numel()
            outs.append(out)
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        print(f'{name} C={C} S={S} B={B}: bad (image, block, wave) tiles {bad}/{tot} over {REPS} runs; bitwise repeatable {same}', flush=True)
