"""GPU test of the sharded search with the REAL HIP local search and HIP merge: two ranks share cuda:0 (the test box has
one GPU; RCCL refuses two ranks on one device, so the collectives run on gloo, which accepts CUDA tensors).  The
nccl/RCCL all-to-all branch is exercised with world size 1 through bench.py on the same box and by the driver's multi-GPU
runs."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lightningdot_amd.sharded import ShardedFlatIndexer
    rng = np.random.default_rng(77)
    n, d, nq, k = 6000, 128, 90, 50
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    bounds = [0, 2500, n]
    qb = [0, 40, nq]
    lo, hi = bounds[rank], bounds[rank + 1]
    sh = ShardedFlatIndexer(d)
    sh.index_local_shard([f'id{i}' for i in range(lo, hi)], torch.from_numpy(x[lo:hi]).cuda())
    s, l = sh.search(torch.from_numpy(q[qb[rank]:qb[rank + 1]]).cuda(), k)
    np.savez(os.path.join(out_dir, f'r{rank}.npz'), s=s.cpu().numpy(), l=l.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_two_ranks_one_gpu(tmp_path):
    import torch
    import torch.multiprocessing as mp
    from lightningdot_amd import _lib
    _lib.require_gpu()
    from tests.util import assert_topk_matches
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(77)
    n, d, nq, k = 6000, 128, 90, 50
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    qb = [0, 40, nq]
    for r in range(world):
        a = np.load(os.path.join(str(tmp_path), f'r{r}.npz'))
        assert_topk_matches(q[qb[r]:qb[r + 1]], x, a['s'], a['l'], k)
