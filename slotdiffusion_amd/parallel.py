"""Data-parallel glue: gradient-only all-reduce of the flat gradient arena (RCCL over xGMI).

The path is data parallel over images / clips: every rank holds the full
model, processes its own shard of the batch and exchanges ONE thing per step -- the sum of the
fp32 gradient arena (138 M floats for SADiffusion).  Because all gradients live in one contiguous
buffer, the exchange is a handful of large bucket all-reduces (default 4 x ~138 MB) rather than
hundreds of per-tensor collectives: on MI355X's point-to-point xGMI fabric large messages are what
keeps every link busy.  Buckets are launched asynchronously on the collective stream in reverse
arena order (the order backward finishes them) and the 1/world scaling is folded into the fused
clip+Adam kernel's inputs by scaling the arena once.
"""
import os
import sys

import torch
import torch.distributed as dist


def respawn_under_launcher(n_ranks, script, argv, port=None):
    """Replace this process by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    script argv...` (one rank per GPU over RCCL; the reference's launch is
    `torch.distributed.launch --nproc_per_node=$GPUS`, scripts/sbatch_run.sh:36-39).  Used by
    `bench.py --gpus N` when it was started without a launcher.  Does not return."""
    port = port or os.environ.get('MASTER_PORT', '29517')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n_ranks}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), script] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC (RCCL across processes)
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


_BF16_WIRE_DEFAULT = False


def use_bf16_wire(flag):
    """Default wire format of the gradient buckets for this process (GraphedTrainStep / Method / bench.py set it
    from the model's compute dtype); SDMI_GRAD_BF16=0 / 1 overrides."""
    global _BF16_WIRE_DEFAULT
    _BF16_WIRE_DEFAULT = bool(flag)


def grad_bf16_enabled():
    """Gradients cross xGMI as bf16 (277 MB instead of 554 MB per step for SADiffusion) when the model computes
    in bf16 -- the default since round 4 -- or SDMI_GRAD_BF16=1; SDMI_GRAD_BF16=0 keeps fp32 buckets.
    Bound: a rank rounds its fp32 partial to bf16 (<= 2^-9 relative), the ring adds w - 1 times in
    bf16 (<= 2^-9 of the running sum each), so an element of the sum is off by <= w 2^-9 of sum |g_r| in the worst
    case (1.6 % at w = 8) and by ~sqrt(w / 3) 2^-9 = 0.32 % rms; tests/test_parallel_cpu.py measures 0.35 % rel-L2
    for an 8-rank ring on CPU -- below the 0.47 % by which the bf16 compute path's gradients themselves differ
    from the fp32 path's (tests/test_gpu_bench_path.py).  The average and everything behind it stay fp32."""
    e = os.environ.get('SDMI_GRAD_BF16')
    if e in ('0', '1'):
        return e == '1'
    return _BF16_WIRE_DEFAULT


def _to_bf16(x):
    if x.is_cuda:
        from . import ops
        return ops.act(x, None, torch.bfloat16)
    return x.to(torch.bfloat16)


def _from_bf16_(dst, src):
    if dst.is_cuda:
        from . import _lib
        _lib.call('sdmi_act', torch.cuda.current_stream().cuda_stream, x=src.data_ptr(),
                  y=dst.data_ptr(), src_dtype=_lib.BF16, dst_dtype=_lib.F32, act=0, n=dst.numel())
    else:
        dst.copy_(src)
    return dst


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def bucket_bounds(n, n_buckets, align=1024):
    per = (n + n_buckets - 1) // n_buckets
    per = (per + align - 1) // align * align
    return [(lo, min(n, lo + per)) for lo in range(0, n, per)]


def allreduce_gradients(arena, world=None, n_buckets=4, group=None):
    """Sum-all-reduce `arena` (flat fp32 gradient buffer) across ranks and average. In place."""
    if world is None:
        world = dist.get_world_size(group)
    if world == 1:
        return arena
    for w in allreduce_range_async(arena, 0, arena.numel(), n_buckets, group):
        w.wait()
    arena.mul_(1.0 / world)
    return arena


class _Bf16Work:
    """Work handle of a bf16 bucket: wait() finishes the collective and widens the sum back into
    the fp32 gradient range."""

    def __init__(self, work, dst, buf):
        self.work, self.dst, self.buf = work, dst, buf

    def wait(self):
        self.work.wait()
        _from_bf16_(self.dst, self.buf)


def allreduce_range_async(arena, lo, hi, n_buckets=2, group=None):
    """Start the sum-all-reduce of arena[lo:hi] (a few large buckets); returns the work handles.
    The caller overlaps other GPU work, then calls finish_allreduce()."""
    works = []
    if hi <= lo:
        return works
    view = arena[lo:hi]
    bf16 = grad_bf16_enabled()
    for a, b in reversed(bucket_bounds(hi - lo, n_buckets)):
        if bf16:
            buf = _to_bf16(view[a:b])
            works.append(_Bf16Work(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True),
                                   view[a:b], buf))
        else:
            works.append(dist.all_reduce(view[a:b], op=dist.ReduceOp.SUM, group=group, async_op=True))
    return works


def finish_allreduce(arena, works, world):
    for w in works:
        w.wait()
    arena.mul_(1.0 / world)
    return arena


def broadcast_parameters(arena, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (one flat broadcast)."""
    dist.broadcast(arena, src=src, group=group)
    return arena
