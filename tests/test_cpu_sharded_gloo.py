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


# uneven shards and uneven query ownership (one rank of the 3-rank case owns NO query), per world size
CASES = {2: dict(bounds=[0, 610, 1500], qb=[0, 11, 37]),
         3: dict(bounds=[0, 410, 1130, 1500], qb=[0, 11, 11, 37])}


def _mk_id(kind, i):
    if kind == 'mixed':          # integers on the first rank of the 2-rank case, strings on the second: the ranks must fall back TOGETHER
        return (7 * i + 3) if i < 610 else f'id{i}'
    if kind == 'npint':          # numpy integers travel as int64 like Python ints (and come back as Python ints)
        return np.int64(7 * i + 3)
    return f'id{i}' if kind == 'str' else (7 * i + 3) if kind == 'int' else ('id', i)


def _expect_repr(kind, i):
    return repr(7 * i + 3) if kind == 'npint' else repr(_mk_id(kind, i))


def _worker(rank, world, port, out_dir, exchange, gather_ids, id_kind='str'):
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
    bounds, qb = CASES[world]['bounds'], CASES[world]['qb']
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

    sh = ShardedFlatIndexer(d, local_search=local_search, merge=merge, exchange=exchange)
    # the shard is added in TWO calls (a second call used to drop the first call's ids from the global map)
    mid = lo + (hi - lo) // 3
    sh.index_local_shard([_mk_id(id_kind, i) for i in range(lo, mid)], None, n_rows=mid - lo, gather_ids=gather_ids)
    sh.index_local_shard([_mk_id(id_kind, i) for i in range(mid, hi)], None, n_rows=hi - mid, gather_ids=gather_ids)
    assert sh.ntotal == n and sh.offsets == bounds
    assert sh.local_ids == [_mk_id(id_kind, i) for i in range(lo, hi)]
    assert sh.index_id_to_db_id == ([_mk_id(id_kind, i) for i in range(n)] if gather_ids else [])
    s, l = sh.search(torch.from_numpy(q[qb[rank]:qb[rank + 1]]), k)
    res = sh.search_knn(torch.from_numpy(q[qb[rank]:qb[rank + 1]]), k)
    np.savez(os.path.join(out_dir, f'r{rank}.npz'), s=s.numpy(), l=l.numpy(),
             ids=np.array([[repr(i) for i in r[0]] for r in res], dtype=object).reshape(len(res), k), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,exchange,gather_ids,id_kind', [(2, 'all_to_all', False, 'str'), (3, 'all_to_all', False, 'str'),
                                                                (2, 'all_gather', True, 'str'), (3, 'all_gather', False, 'int'),
                                                                (3, 'all_to_all', False, 'tuple'), (2, 'all_to_all', False, 'mixed'),
                                                                (2, 'all_to_all', False, 'npint')])
def test_sharded_search_gloo(tmp_path, world, exchange, gather_ids, id_kind):
    # (resolve_ids: string ids travel as UTF-8 bytes, integer ids as int64 — tensor all-to-alls —, other id types are pickled)
    from oracle import oracle_np as O
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), exchange, gather_ids, id_kind), nprocs=world, join=True)
    rng = np.random.default_rng(2024)
    n, d, nq, k = 1500, 32, 37, 20
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    whole = O.FlatIP(d)
    whole.add(x)
    es, el = whole.search(q, k)
    qb = CASES[world]['qb']
    for r in range(world):
        a = np.load(os.path.join(str(tmp_path), f'r{r}.npz'), allow_pickle=True)
        assert a['l'].shape == (qb[r + 1] - qb[r], k)
        np.testing.assert_array_equal(a['l'], el[qb[r]:qb[r + 1]])
        np.testing.assert_allclose(a['s'], es[qb[r]:qb[r + 1]], rtol=1e-6, atol=1e-6)
        assert [list(row) for row in a['ids']] == [[_expect_repr(id_kind, int(i)) for i in row] for row in el[qb[r]:qb[r + 1]]]
