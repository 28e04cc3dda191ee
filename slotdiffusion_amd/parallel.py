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
import torch
import torch.distributed as dist


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
    works = []
    for lo, hi in reversed(bucket_bounds(arena.numel(), n_buckets)):
        works.append(dist.all_reduce(arena[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    arena.mul_(1.0 / world)
    return arena


def allreduce_range_async(arena, lo, hi, n_buckets=2, group=None):
    """Start the sum-all-reduce of arena[lo:hi] (a few large buckets); returns the work handles.
    The caller overlaps other GPU work, then calls finish_allreduce()."""
    works = []
    if hi <= lo:
        return works
    view = arena[lo:hi]
    for a, b in reversed(bucket_bounds(hi - lo, n_buckets)):
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
