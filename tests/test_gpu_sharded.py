"""GPU test of the sharded search with the REAL HIP local search and HIP merge: two ranks share cuda:0 (the test box has
one GPU; RCCL refuses two ranks on one device, so the collectives run on gloo, which accepts CUDA tensors).  The
nccl/RCCL all-to-all branch is exercised with world size 1 through bench.py on the same box and by the driver's multi-GPU
runs."""
import os
import socket
import sys

import numpy as np
import pytest

from tests.util import assert_topk_matches

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


# ---- round 4: four ranks on one GPU through the FUSED path at the BASELINE shard shapes ------------------------------------------
# case A: the COCO-5k index of BASELINE configs[2] (25 000 x 768 rows) in 4 UNEQUAL shards (every shard below the fused threshold: the
#         one-pass paths, neutral warm-up statistics);  case B: 125 000 rows per rank, the shard size of configs[3], where the fused
#         filter runs and the ranks agree on thresholds after their warm-ups.  Unequal query counts, one rank without a query.
_CASES = {
    'coco': dict(bounds=[0, 4000, 11000, 19000, 25000], counts=[300, 0, 500, 224], d=768, k=10),
    'shard125k': dict(bounds=[0, 125000, 250000, 375000, 500000], counts=[0, 700, 804, 800], d=768, k=100),
    # equal query slices (the bench's shape): the blocked exchange — re-score kernel -> send buffer -> ONE all-to-all -> merge kernel —
    # with the per-search query-count exchange switched off
    'equal': dict(bounds=[0, 40000, 80000, 120000, 160000], counts=[320, 320, 320, 320], d=256, k=100, equal=True),
    # rows that are NOT exchangeable between the shards: the first 30 000 rows of shard 1 are three times as long as all others, so
    # every query's best rows sit there and that shard's pooled statistics (taken from its first 4096 rows) aim above the global k'-th
    # best.  The ranks notice together (ldot_index_shard_floor: fewer than k' rows at or above the largest level), repeat the search on their own thresholds and back off
    'skewed': dict(bounds=[0, 40000, 80000, 120000, 160000], counts=[320, 320, 320, 320], d=128, k=100, equal=True,
                   scale=(40000, 70000, 3.0)),
}


def _case_rows(c, lo, hi, seed):
    rows = _gen_rows(lo, hi, c['d'], seed)
    if 'scale' in c:
        a, b, f = c['scale']
        a, b = max(a, lo), min(b, hi)
        if a < b:
            rows[a - lo:b - lo] *= f
    return rows


def _gen_rows(lo, hi, d, seed):
    """rows [lo, hi) of a clustered synthetic index, generated on the GPU in blocks of 25 000 rows so that any shard layout sees the
    same data"""
    import torch
    out, blk = [], 25000
    for b in range(lo // blk, (hi + blk - 1) // blk):
        g = torch.Generator(device='cuda').manual_seed(seed * 1000 + b)
        rows = torch.randn(blk, d, device='cuda', generator=g)
        out.append(rows[max(lo, b * blk) - b * blk:min(hi, (b + 1) * blk) - b * blk])
    return torch.cat(out, 0)


def _gen_queries(n, d, seed, rows_total):
    import torch
    g = torch.Generator(device='cuda').manual_seed(seed)
    return torch.randn(n, d, device='cuda', generator=g)


def _worker4(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from lightningdot_amd.sharded import ShardedFlatIndexer
    for ci, (name, c) in enumerate(_CASES.items()):
        lo, hi = c['bounds'][rank], c['bounds'][rank + 1]
        qs = np.cumsum([0] + c['counts'])
        sh = ShardedFlatIndexer(c['d'], equal_query_counts=bool(c.get('equal')))
        sh.index_local_shard(list(range(lo, hi)), _case_rows(c, lo, hi, 11 + ci))
        q = _gen_queries(int(qs[-1]), c['d'], 500 + ci, c['bounds'][-1])[qs[rank]:qs[rank + 1]].contiguous()
        from lightningdot_amd import _lib as L
        sh.local.index.set_option(L.OPT_PROFILE, 1)
        # default: every shard scans on statistics pooled over the whole index, ONE exchange (three numbers per query) after the pass
        assert sh.exchange_warmup is False and sh.pooled_statistics is True
        s, l = sh.search(q, c['k'])
        st, n1 = sh.local.index.last_stats(), sh.local.index.last_profile()['launches']
        info = dict(sh.last_search)
        s9, l9 = sh.search(q, c['k'])         # (after a repeated search the pooled statistics are skipped for a while)
        info9 = dict(sh.last_search)
        sh.pooled_statistics = False          # option: every shard on its own (optimistic) thresholds, the same exchange
        s1, l1 = sh.search(q, c['k'])
        st1, n1b = sh.local.index.last_stats(), sh.local.index.last_profile()['launches']
        sh.exchange_warmup = True             # option: thresholds agreed after the warm-ups (ldot_index_search_warmup / _scan)
        s0, l0 = sh.search(q, c['k'])
        st0, n0 = sh.local.index.last_stats(), sh.local.index.last_profile()['launches']
        sh.exchange_warmup = False
        assert torch.equal(s, s0) and torch.equal(l, l0)
        assert torch.equal(s, s1) and torch.equal(l, l1)
        assert torch.equal(s, s9) and torch.equal(l, l9)
        np.savez(os.path.join(out_dir, f'{name}_r{rank}.npz'), s=s.cpu().numpy(), l=l.cpu().numpy(),
                 fused_pairs=st['fused_pairs'], fused_candidates=st['fused_candidates'], overflowed=st['overflowed_queries'],
                 fused_candidates_own=st1['fused_candidates'], launches_own=n1b,
                 fused_candidates_agreed=st0['fused_candidates'], launches=n1, launches_agreed=n0,
                 pooled=int(info['pooled']), repeated=int(info['repeated']), pooled_next=int(info9['pooled']))
        del sh
        torch.cuda.empty_cache()
        dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_four_ranks_fused_path_baseline_shapes(tmp_path):
    import torch
    import torch.multiprocessing as mp
    from lightningdot_amd import _lib
    from lightningdot_amd.indexer import FlatIPIndex
    _lib.require_gpu()
    world, port = 4, _free_port()
    mp.spawn(_worker4, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for ci, (name, c) in enumerate(_CASES.items()):
        n = c['bounds'][-1]
        ix = FlatIPIndex(c['d'])
        ix.add(_case_rows(c, 0, n, 11 + ci))
        qs = np.cumsum([0] + c['counts'])
        q = _gen_queries(int(qs[-1]), c['d'], 500 + ci, n)
        es, el = ix.search_tensors(q, c['k'])
        es, el = es.cpu().numpy(), el.cpu().numpy()
        # ... and the checker itself, not only the unsharded HIP search: the merged lists of the four ranks against an fp64 brute-force
        # scan of the whole index, for the first queries of every rank's slice
        x_all = _case_rows(c, 0, n, 11 + ci).cpu().numpy()
        for r in range(world):
            nchk = min(8, c['counts'][r])
            if nchk:
                a = np.load(os.path.join(str(tmp_path), f'{name}_r{r}.npz'))
                assert_topk_matches(q[qs[r]:qs[r] + nchk].cpu().numpy(), x_all, a['s'][:nchk], a['l'][:nchk], c['k'])
        del x_all
        cand = []
        for r in range(world):
            a = np.load(os.path.join(str(tmp_path), f'{name}_r{r}.npz'))
            assert a['s'].shape == (c['counts'][r], c['k'])
            # the same rows in the same order with bit-identical fp32 scores as the unsharded search of the whole index
            np.testing.assert_array_equal(a['l'], el[qs[r]:qs[r + 1]])
            np.testing.assert_array_equal(a['s'], es[qs[r]:qs[r + 1]])
            assert int(a['pooled']) == 1
            if name == 'skewed':
                # the pooled statistics of shard 1 aimed too high: noticed by all ranks, searched again, skipped in the next search
                assert int(a['repeated']) == 1 and int(a['pooled_next']) == 0
                continue
            assert int(a['repeated']) == 0 and int(a['pooled_next']) == 1
            assert int(a['overflowed']) == 0
            if name == 'shard125k':   # every rank scanned its shard with the fused filter (all queries x its rows beyond the warm-up)
                assert int(a['fused_pairs']) > 0.9 * int(qs[-1]) * 125000
                cand.append((int(a['fused_candidates']), int(a['fused_candidates_own']), int(a['fused_candidates_agreed']),
                             int(a['launches']), int(a['launches_own']), int(a['launches_agreed'])))
            elif name == 'equal':
                assert int(a['fused_pairs']) > 0.8 * int(qs[-1]) * 40000
            else:
                assert int(a['fused_pairs']) == 0
        if name == 'shard125k':
            # all three schedules give the same lists (asserted in the workers); pooled statistics admit the fewest records in no
            # more launches, a shard's own optimistic thresholds fewer than the agreed-threshold option (tools/shard_floor.py)
            assert all(a < b < c_ and la <= lb for a, b, c_, la, lb, lc in cand), cand
        del ix
        torch.cuda.empty_cache()


def test_shard_entry_points_contract_in_one_process():
    """ldot_index_search_begin_shard / ldot_index_shard_floor without collectives: one shard of one (its statistics are the search's own:
    level -inf, floor = its k'-th best, every count = k', _finish(floor) = the plain search), two shards played in turn with pooled
    statistics, and the argument checks of the C entry points."""
    import ctypes
    import torch
    from lightningdot_amd import _lib as L
    from lightningdot_amd.indexer import FlatIPIndex
    L.require_gpu()
    g = torch.Generator(device='cuda').manual_seed(4)
    x = torch.randn(90000, 128, device='cuda', generator=g)
    q = torch.randn(700, 128, device='cuda', generator=g)
    k = 20
    whole = FlatIPIndex(128)
    whole.add(x)
    es, el = whole.search_tensors(q, k)
    # one shard of one
    stat = whole.search_begin_shard(q, k, 1, 0)
    assert stat.shape == (3, 700) and bool((stat[2] == float('-inf')).all()) and bool((stat[1] == -stat[0]).all())
    floor, count, kp = whole.shard_floor(stat)
    assert kp >= k and bool((floor == stat[0]).all()) and bool((count == kp).all())
    s, l = whole.search_finish(floor)
    assert torch.equal(s, es) and torch.equal(l, el)
    # two shards, pooled statistics, played in turn; shares in proportion to the rows
    a, b = FlatIPIndex(128), FlatIPIndex(128)
    a.add(x[:50000])
    b.add(x[50000:])
    st = torch.maximum(a.search_begin_shard(q, k, 2, 90000, share=5 / 9), b.search_begin_shard(q, k, 2, 90000, share=4 / 9))
    assert bool(torch.isfinite(st[2]).all())                      # both shards scanned on pooled statistics and published their levels
    fa, ca, kp = a.shard_floor(st)
    fb, cb, _ = b.shard_floor(st)
    assert torch.equal(fa, fb) and bool((ca + cb >= kp).all())     # every query proven
    sa, la = a.search_finish(fa)
    sb, lb = b.search_finish(fb)
    S = torch.cat([sa, sb], 1)
    Lb = torch.cat([la, torch.where(lb >= 0, lb + 50000, lb)], 1)
    S = torch.where(Lb >= 0, S, torch.full_like(S, float('-inf')))
    o = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
    assert torch.equal(torch.gather(S, 1, o), es) and torch.equal(torch.gather(Lb, 1, o), el)
    # argument checks
    lib = L.load_library()
    buf = torch.empty(3 * 700, dtype=torch.float32, device='cuda')
    qp, bp = ctypes.c_void_p(q.data_ptr()), ctypes.c_void_p(buf.data_ptr())
    for parts, share, total in ((0, 0.5, 0), (2, 1.5, 0), (2, -0.1, 0), (2, 0.5, -1)):
        rc = lib.ldot_index_search_begin_shard(whole._h, qp, 700, L.F32, L.DEVICE, 0, k, parts, share, total, bp, None)
        assert rc == -1, (parts, share, total, rc)
    assert lib.ldot_index_search_begin_shard(whole._h, qp, 700, L.F32, L.DEVICE, 0, k, 2, 0.5, 0, None, None) == -1   # NULL statistics buffer
    assert lib.ldot_index_shard_floor(None, bp, bp, bp, None, None) == -1
    fresh = FlatIPIndex(128)
    with pytest.raises(L.LdotError):
        fresh.shard_floor(buf.view(3, 700))                       # no pending search_begin_shard
