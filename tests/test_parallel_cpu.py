"""world_size-2 gloo tests (CPU) of the data-parallel glue: shard arithmetic, bucketed
gradient all-reduce + averaging, parameter broadcast -- the only collective on the hot path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from slotdiffusion_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        grads = torch.randn(n, generator=g)
        ref = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r))
                  for r in range(world)) / world
        parallel.allreduce_gradients(grads, world, n_buckets=4)
        ok1 = torch.allclose(grads, ref, atol=1e-6)
        params = torch.full((1000,), float(rank))
        parallel.broadcast_parameters(params, src=0)
        ok2 = bool((params == 0).all())
        lo, hi = parallel.shard_range(13, rank, world)
        ret[rank] = (ok1, ok2, lo, hi)
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo():
    world, n = 2, 100003
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, n, ret), nprocs=world, join=True)
        r = dict(ret)
    assert r[0][:2] == (True, True) and r[1][:2] == (True, True)
    assert (r[0][2], r[0][3]) == (0, 7) and (r[1][2], r[1][3]) == (7, 13)


def test_bucket_bounds_cover_everything():
    for n in (1, 1023, 1024, 138_480_000):
        b = parallel.bucket_bounds(n, 4)
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
