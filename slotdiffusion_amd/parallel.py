"""Data-parallel glue: gradient-only all-reduce of the flat gradient arena (RCCL over xGMI).

The path is data parallel over images / clips: every rank holds the full
model, processes its own shard of the batch and exchanges ONE thing per step -- the sum of the
gradient arena (138 M values for SADiffusion).  Because all gradients live in one contiguous
buffer, the exchange is a handful of large bucket all-reduces (default 4 x ~138 MB) rather than
hundreds of per-tensor collectives: on MI355X's point-to-point xGMI fabric large messages are what
keeps every link busy.  Buckets are launched asynchronously on the collective stream in reverse
arena order (the order backward finishes them).

`GradReducer` is the exchange of one model: the buckets are all-reduced IN PLACE -- fp32 wire: slices
of the gradient arena; bf16 wire: slices of one persistent bf16 buffer with the arena's offsets (no
per-step allocation) -- and nothing runs behind the collective: the optimiser kernels read the summed
gradients where the all-reduce left them (`grad_src`, bf16 or fp32) and apply the 1 / world factor
as a scalar (`grad_scale`; sdmi.h: SdmiAdamArgs.gscale / g_dtype, SdmiSqSumArgs.g_dtype).  The wire
format is an explicit argument (default fp32, like the reference's DDP); SDMI_GRAD_BF16=0 / 1 overrides.
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


def resolve_wire(wire=None):
    """Wire format of the gradient buckets: 'fp32' (library default -- what the reference's DDP / amp path
    reduces) or 'bf16' (an explicit opt-in of the caller: bench.py for the bf16 model, `ddp_grad_dtype` of a params
    file); SDMI_GRAD_BF16=0 / 1 overrides either.
    Bound of the bf16 wire: a rank rounds its fp32 partial to bf16 (<= 2^-9 relative), the ring adds
    w - 1 times in bf16 (<= 2^-9 of the running sum each), so an element of the sum is off by <= w 2^-9 of
    sum |g_r| in the worst case (1.6 % at w = 8) and by ~sqrt(w / 3) 2^-9 = 0.32 % rms; tests/test_parallel_cpu.py
    measures 0.35 % rel-L2 for an 8-rank ring on CPU -- below the 0.47 % by which the bf16 compute path's gradients
    themselves differ from the fp32 path's (tests/test_gpu_bench_path.py).  The average and everything behind it
    stay fp32; there is no convergence study behind it, which is why it is not the default."""
    e = os.environ.get('SDMI_GRAD_BF16')
    if e in ('0', '1'):
        return 'bf16' if e == '1' else 'fp32'
    if wire in (None, 'fp32', torch.float32):
        return 'fp32'
    if wire in ('bf16', torch.bfloat16):
        return 'bf16'
    raise ValueError(f'gradient wire format {wire!r}: fp32 or bf16')


def _cast_into(dst, src):
    """dst[:] = src (dtype conversion, one launch on the current stream; no allocation)."""
    if dst.is_cuda:
        from . import _lib
        code = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}
        _lib.call('sdmi_act', torch.cuda.current_stream().cuda_stream, x=src.data_ptr(), y=dst.data_ptr(),
                  src_dtype=code[src.dtype], dst_dtype=code[dst.dtype], act=0, n=dst.numel())
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


class GradReducer:
    """Gradient exchange of one model: sum-all-reduce of ranges of its flat fp32 gradient arena.

    start(lo, hi) launches the buckets of arena[lo:hi] asynchronously and returns their work handles,
    finish(works) waits for them.  Afterwards `grad_src` holds the SUM over ranks (the arena itself on a
    fp32 wire, the persistent bf16 buffer on a bf16 wire) and `grad_scale` = 1 / world is what the
    optimiser multiplies it by -- no averaging or widening pass runs over the arena, and on a bf16 wire
    the arena (and `p.grad`) keeps this rank's LOCAL gradients."""

    def __init__(self, arena, world=None, wire=None, group=None):
        self.arena, self.group = arena, group
        self.world = dist.get_world_size(group) if world is None else int(world)
        self.wire = resolve_wire(wire)
        # zeros, once: elements outside the reduced ranges (alignment gaps between parameter runs) stay 0 for the
        # optimiser's norm pass over the whole buffer
        self.wire_buf = torch.zeros(arena.numel(), dtype=torch.bfloat16, device=arena.device) \
            if self.wire == 'bf16' and self.world > 1 else None

    @property
    def grad_src(self):
        return self.arena if self.wire_buf is None else self.wire_buf

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def start(self, lo, hi, n_buckets=2):
        works = []
        if hi <= lo or self.world == 1:
            return works
        for a, b in reversed(bucket_bounds(hi - lo, n_buckets)):
            if self.wire_buf is not None:
                buf = _cast_into(self.wire_buf[lo + a:lo + b], self.arena[lo + a:lo + b])
            else:
                buf = self.arena[lo + a:lo + b]
            works.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works

    @staticmethod
    def finish(works):
        for w in works:
            w.wait()

    def reduce_all(self, n_buckets=4):
        self.finish(self.start(0, self.arena.numel(), n_buckets))
        return self


def allreduce_gradients(arena, world=None, n_buckets=4, group=None, wire=None):
    """Sum-all-reduce `arena` (flat fp32 gradient buffer) across ranks and AVERAGE it in place: the
    self-contained form (tests, callers with their own optimiser).  The training step uses GradReducer and lets the
    optimiser kernels apply 1 / world instead of this extra pass over the arena."""
    if world is None:
        world = dist.get_world_size(group)
    if world == 1:
        return arena
    r = GradReducer(arena, world, wire, group).reduce_all(n_buckets)
    if r.wire_buf is not None:
        _cast_into(arena, r.wire_buf)
    arena.mul_(r.grad_scale)
    return arena


def broadcast_parameters(arena, src=0, group=None):
    """Make every rank start from rank `src`'s parameters (one flat broadcast)."""
    dist.broadcast(arena, src=src, group=group)
    return arena
