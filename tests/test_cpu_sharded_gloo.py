"""world_size-2 `gloo` test (CPU) of the multi-GPU collective logic in lightningdot_amd.sharded: query all-gather,
local->global label offsets, partial top-k exchange, merge.  The local searcher and the merge are injected (oracle
based stand-ins — test infrastructure); on GPUs the defaults are the HIP implementations."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import oracle_np as O
    from lightningdot_amd.sharded import ShardedFlatIndexer
    rng = np.random.default_rng(2024)
    n, d, nq, k = 1500, 32, 37, 20
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    bounds = [0, 610, n]                         # uneven shards
    qb = [0, 11, nq]                             # uneven query ownership
    lo, hi = bounds[rank], bounds[rank + 1]
    shard = O.FlatIP(d)
    shard.add(x[lo:hi])

    def local_search(q_all, kk):                 # oracle stand-in for the HIP search of the local shard
        s, l = shard.search(q_all.numpy(), kk)
        return torch.from_numpy(s), torch.from_numpy(l)

    def merge(ps, pl, kk):                       # oracle stand-in for ldot_merge_topk
        parts = [(ps[i].numpy(), pl[i].numpy()) for i in range(ps.shape[0])]
        s, l = O.merge_topk(parts, kk)
        return torch.from_numpy(s), torch.from_numpy(l)

    sh = ShardedFlatIndexer(d, local_search=local_search, merge=merge)
    sh.index_local_shard([f'id{i}' for i in range(lo, hi)], None, n_rows=hi - lo)
    assert sh.ntotal == n and sh.offsets == bounds
    assert sh.index_id_to_db_id == [f'id{i}' for i in range(n)]
    s, l = sh.search(torch.from_numpy(q[qb[rank]:qb[rank + 1]]), k)
    res = sh.search_knn(torch.from_numpy(q[qb[rank]:qb[rank + 1]]), k)
    np.savez(os.path.join(out_dir, f'r{rank}.npz'), s=s.numpy(), l=l.numpy(),
             ids=np.array([r[0] for r in res], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_two_ranks_gloo(tmp_path):
    from oracle import oracle_np as O
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(2024)
    n, d, nq, k = 1500, 32, 37, 20
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    whole = O.FlatIP(d)
    whole.add(x)
    es, el = whole.search(q, k)
    qb = [0, 11, nq]
    for r in range(world):
        a = np.load(os.path.join(str(tmp_path), f'r{r}.npz'), allow_pickle=True)
        np.testing.assert_array_equal(a['l'], el[qb[r]:qb[r + 1]])
        np.testing.assert_allclose(a['s'], es[qb[r]:qb[r + 1]], rtol=1e-6, atol=1e-6)
        assert [list(row) for row in a['ids']] == [[f'id{i}' for i in row] for row in el[qb[r]:qb[r + 1]]]


def _loss_worker(rank, world, port, out_dir):
    """Cross-rank in-batch negatives: autograd-aware embedding all-gather (lightningdot_amd.loss._AllGatherCat)."""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lightningdot_amd.loss import _AllGatherCat
    g = torch.Generator().manual_seed(7)
    full = torch.randn(10, 8, generator=g)
    sizes = [4, 6]
    start = sum(sizes[:rank])
    local = full[start:start + sizes[rank]].clone().requires_grad_()
    gathered = _AllGatherCat.apply(local)
    assert torch.equal(gathered.detach(), full)
    w = torch.arange(80, dtype=torch.float32).reshape(10, 8) * (rank + 1)
    (gathered * w).sum().backward()              # rank-dependent loss
    torch.save(local.grad, os.path.join(out_dir, f'g{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_allgather_autograd_two_ranks_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_loss_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    w = torch.arange(80, dtype=torch.float32).reshape(10, 8)
    total = w * 1 + w * 2                        # d(sum over ranks of loss_r)/d(full)
    sizes = [4, 6]
    for r in range(world):
        g = torch.load(os.path.join(str(tmp_path), f'g{r}.pt'))
        start = sum(sizes[:r])
        assert torch.equal(g, total[start:start + sizes[r]])


def _dp_worker(rank, world, port, out_dir):
    """Config-5 data-parallel plumbing: rank-0 parameter broadcast + flat-bucket averaged gradient all-reduce."""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lightningdot_amd.train import allreduce_gradients, broadcast_parameters
    torch.manual_seed(100 + rank)                      # ranks start DIFFERENT
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.LayerNorm(5), torch.nn.Linear(5, 3))
    m[2].bias.requires_grad_(False)                    # a frozen parameter takes no part
    broadcast_parameters(m, 0)
    x = torch.full((4, 6), float(rank + 1))
    m(x).sum().backward()
    local = [p.grad.clone() if p.grad is not None else None for p in m.parameters()]
    allreduce_gradients(m.parameters(), bucket_bytes=64)        # tiny buckets: several collectives
    torch.save(dict(params=[p.data.clone() for p in m.parameters()], local=local,
                    reduced=[p.grad.clone() if p.grad is not None else None for p in m.parameters()]),
               os.path.join(out_dir, f'dp{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_dp_broadcast_and_gradient_allreduce_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_dp_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(os.path.join(str(tmp_path), 'dp0.pt'))
    b = torch.load(os.path.join(str(tmp_path), 'dp1.pt'))
    for pa, pb in zip(a['params'], b['params']):
        assert torch.equal(pa, pb)                     # broadcast made the replicas identical
    for la, lb, ra, rb in zip(a['local'], b['local'], a['reduced'], b['reduced']):
        if la is None:
            assert ra is None and rb is None
            continue
        assert torch.allclose(ra, (la + lb) / 2) and torch.equal(ra, rb)
