"""GPU parity tests of the retrieval path: HIP library (through the C ABI / DenseFlatIndexer) vs the CPU oracle."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_np as O
from tests.util import assert_topk_matches, planted_queries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L():
    from lightningdot_amd import _lib
    _lib.require_gpu()
    return _lib


def _index(x, **opts):
    from lightningdot_amd import _lib as LL
    from lightningdot_amd.indexer import FlatIPIndex
    ix = FlatIPIndex(x.shape[1], normalize=opts.pop('normalize', False))
    for k, v in opts.items():
        ix.set_option(getattr(LL, 'OPT_' + k.upper()), v)
    if x.shape[0]:
        ix.add(x)
    return ix


@pytest.mark.parametrize('nq,n,d,k', [
    (37, 1000, 768, 100),      # Flickr text->image shape (reduced nq)
    (300, 5000, 768, 100),     # Flickr image->text shape
    (5, 300, 48, 10),          # D not a multiple of 64
    (1, 257, 100, 1),          # single query, k = 1, ragged rows
    (513, 70, 64, 100),        # k > ntotal -> padding
    (260, 33000, 128, 50),     # more than one dense chunk (32768) and more than one query tile
    (70, 5000, 64, 2048),      # largest k: the margin is kept (k' = 2560)
])
def test_dense_matches_oracle(L, nq, n, d, k):
    rng = np.random.default_rng(nq * 7 + n)
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ix = _index(x, mode=L.MODE_DENSE)
    assert ix.ntotal == n
    s, l = ix.search(q, k)
    assert_topk_matches(q, x, s, l, k)
    # oracle (fp32 blocked sgemm + selection) agrees as well
    os_, ol = O.FlatIP(d), None
    os_.add(x)
    so, lo = os_.search(q, k)
    np.testing.assert_allclose(s, so, rtol=0, atol=1e-3)
    assert (l[:, 0] == lo[:, 0]).mean() > 0.999


def test_empty_and_errors(L):
    from lightningdot_amd.indexer import FlatIPIndex
    ix = FlatIPIndex(32)
    s, l = ix.search(np.zeros((3, 32), np.float32), 5)
    assert (l == -1).all() and (s == np.float32(O.NEG_FLT_MAX)).all()
    with pytest.raises(L.LdotError):
        ix.search(np.zeros((3, 32), np.float32), 0)
    with pytest.raises(L.LdotError):
        ix.search(np.zeros((3, 32), np.float32), L.MAX_K + 1)
    with pytest.raises(ValueError):
        ix.add(np.zeros((3, 31), np.float32))
    with pytest.raises(L.LdotError):
        FlatIPIndex(0)


def test_incremental_add_and_ties(L):
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3000, 96)).astype(np.float32)
    x[100] = x[50]
    x[2999] = x[50]                      # exact ties -> lower label first
    q = np.concatenate([x[50:51] * 2.0, rng.standard_normal((40, 96)).astype(np.float32)])
    from lightningdot_amd.indexer import FlatIPIndex
    ix = FlatIPIndex(96)
    ix.set_option(L.OPT_MODE, L.MODE_DENSE)
    for a, b in [(0, 1), (1, 700), (700, 701), (701, 3000)]:
        ix.add(x[a:b])
    s, l = ix.search(q, 20)
    assert list(l[0, :3]) == [50, 100, 2999]
    assert s[0, 0] == s[0, 1] == s[0, 2]
    assert_topk_matches(q, x, s, l, 20)
    np.testing.assert_array_equal(ix.get_rows(699, 3), x[699:702])


def test_device_tensors_and_dtypes(L):
    import torch
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2000, 256)).astype(np.float32)
    q = rng.standard_normal((65, 256)).astype(np.float32)
    from lightningdot_amd.indexer import FlatIPIndex
    ix = FlatIPIndex(256)
    ix.add(torch.from_numpy(x).cuda())
    s, l = ix.search_tensors(torch.from_numpy(q).cuda(), 30)
    assert s.is_cuda and l.is_cuda and l.dtype == torch.int64
    assert_topk_matches(q, x, s.cpu().numpy(), l.cpu().numpy(), 30)
    # bf16 / fp16 inputs are widened exactly
    xb = torch.from_numpy(x).cuda().bfloat16()
    qb = torch.from_numpy(q).cuda().half()
    ix2 = FlatIPIndex(256)
    ix2.add(xb)
    s2, l2 = ix2.search(qb, 30)
    assert_topk_matches(qb.float().cpu().numpy(), xb.float().cpu().numpy(), s2, l2, 30)


def test_normalize_option(L):
    rng = np.random.default_rng(8)
    x = (rng.standard_normal((1500, 768)) * rng.uniform(0.1, 5, (1500, 1))).astype(np.float32)
    q = (rng.standard_normal((50, 768)) * 3).astype(np.float32)
    ix = _index(x, normalize=True)
    s, l = ix.search(q, 10)
    xn = x / np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True)
    qn = q / np.linalg.norm(q.astype(np.float64), axis=1, keepdims=True)
    assert_topk_matches(qn, xn, s, l, 10, atol=1e-5)
    assert np.abs(s).max() <= 1.0 + 1e-5


@pytest.mark.parametrize('nq,n,d,k', [(300, 20000, 768, 100), (1100, 70000, 128, 10), (257, 150000, 64, 1000),
                                      (770, 150000, 128, 2048)])   # largest k keeps its margin (k' = 2560)
def test_fused_matches_oracle(L, nq, n, d, k):
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((n, d)).astype(np.float32)
    q, g = planted_queries(x, nq)
    ix = _index(x, mode=L.MODE_FUSED, warm_rows=2048)
    s, l = ix.search(q, k)
    st = ix.last_stats()
    assert st['fused_pairs'] > 0 and st['overflowed_queries'] == 0, st
    assert (l[:, 0] == g).all()
    assert_topk_matches(q, x, s, l, k)
    # fused == dense, bit for bit (both end in the same fp32 re-score and ordering)
    ixd = _index(x, mode=L.MODE_DENSE)
    sd, ld = ixd.search(q, k)
    np.testing.assert_array_equal(l, ld)
    np.testing.assert_array_equal(s, sd)


def test_fused_adversarial_order_falls_back(L):
    """Rows sorted so that every later row beats all earlier ones for every query: the lane-private pools
    overflow, the library detects it and redoes the search densely — results stay exact."""
    rng = np.random.default_rng(11)
    # 8 query blocks -> 32 row slices; a sub-pool (one lane group of one wave row: 48 rows of a 384-row tile) holds 16 records of 8 rows
    # each, so a launch with 3 tiles per slice (the third geometric launch, rows 26 624 .. 63 488) overflows it: 144 winning rows = 18 records
    n, d, nq = 70000, 64, 2048
    base = rng.standard_normal(d).astype(np.float32)
    base /= np.linalg.norm(base)
    x = (rng.standard_normal((n, d)) * 0.01).astype(np.float32) + np.outer(np.linspace(0.0, 300.0, n), base).astype(np.float32)
    q = (np.outer(np.ones(nq), base) + rng.standard_normal((nq, d)) * 0.01).astype(np.float32)
    ix = _index(x, mode=L.MODE_FUSED, warm_rows=2048)
    s, l = ix.search(q, 100)
    st = ix.last_stats()
    assert st['overflowed_queries'] > 0, st
    assert_topk_matches(q, x, s, l, 100)


def test_overflow_recovery_redoes_only_the_flagged_queries(L):
    """Rows stored in cluster order (what the inverted-file index keeps): a contiguous block of 40 000 similar rows overflows the
    lane-private pools of the few queries that match it.  Only those queries are searched again — one fused launch with their
    (valid) thresholds, no dense pass at all — and the results equal the dense path's bit for bit."""
    rng = np.random.default_rng(33)
    n, d, nq, nb = 200000, 64, 2048, 24      # (8 query blocks -> 32 row slices, 256 sub-pools per query)
    x = rng.standard_normal((n, d)).astype(np.float32)
    v = rng.standard_normal(d).astype(np.float32)
    v *= 4.0 / np.linalg.norm(v)
    x[110000:190000] += v                                   # the block: 208 tiles of 384 rows = 6.5 per row slice (6 records per tile: > 16 records per sub-pool)
    q, g = planted_queries(x[:100000], nq)                  # (planted outside the block)
    q[:nb] = v + 0.7 * rng.standard_normal((nb, d)).astype(np.float32)     # queries that match every row of the block
    ix = _index(x, mode=L.MODE_FUSED, warm_rows=4096)
    s, l = ix.search(q, 100)
    st = ix.last_stats()
    assert nb <= st['overflowed_queries'] < nq // 8, st     # the block's queries (+ a few whose direction happens to favour it)
    assert st['dense_pairs'] == 4096 * nq, st               # the warm-up only: nobody took the dense path
    assert st['fused_pairs'] == (n - 4096) * nq + n * st['overflowed_queries'], st
    assert_topk_matches(q, x, s, l, 100)
    ixd = _index(x, mode=L.MODE_DENSE)
    sd, ld = ixd.search(q, 100)
    np.testing.assert_array_equal(l, ld)
    np.testing.assert_array_equal(s, sd)
    # a second search on the same handle (flags and counters were left clean)
    s2, l2 = ix.search(q, 100)
    np.testing.assert_array_equal(l2, l)


def test_optimistic_thresholds_are_verified_and_front_loaded_rows_are_redone(L):
    """LDOT_OPT_OPTIMISTIC (default on, large batches): the fused scan filters with order statistics of the rows seen so far that lie
    below the final threshold — unless the row ORDER front-loads a query's best rows.  Here the sixteen best rows of half the queries
    sit in the warm-up region (the first threshold of this scan is the 14th best warm-up score: k' r / N = 2.6 of the final top k' are
    expected there): their optimistic thresholds end far above their final k'-th best, the end-of-scan check flags them, and the recovery
    searches them again on guaranteed thresholds.  Results equal the dense path's bit for bit either way; with i.i.d. row order nothing
    is flagged and the scan admits a fraction of the candidates the guaranteed schedule does."""
    rng = np.random.default_rng(77)
    n, d, nq = 200000, 64, 512
    q = rng.standard_normal((nq, d)).astype(np.float32)
    x = rng.standard_normal((n, d)).astype(np.float32)
    for j in range(4096):                                   # rows 0 .. 4095: sixteen near-copies of each of the first 256 queries
        x[j] = q[j % 256] * (1.0 + 0.01 * rng.standard_normal())
    ix = _index(x, mode=L.MODE_FUSED, warm_rows=4096)
    s, l = ix.search(q, 100)
    st = ix.last_stats()
    assert 256 <= st['overflowed_queries'] < 300, st        # the front-loaded queries failed the check and were searched again
    # an index whose search fails the check for many queries switches to the scrambled scan order (LDOT_OPT_SCAN_ORDER, auto): the next
    # search sees the front-loaded rows spread over the scan like everybody else's, flags nothing and returns the same lists
    # (an index that keeps failing even then stops trying for a while: 16 searches on guaranteed thresholds, then 32, ...)
    sb, lb = ix.search(q, 100)
    assert ix.last_stats()['overflowed_queries'] == 0
    np.testing.assert_array_equal(lb, l)
    np.testing.assert_array_equal(sb, s)
    ixd = _index(x, mode=L.MODE_DENSE)
    sd, ld = ixd.search(q, 100)
    np.testing.assert_array_equal(l, ld)
    np.testing.assert_array_equal(s, sd)
    assert_topk_matches(q, x, s, l, 100)
    # the same rows in random order: nothing is flagged, and the optimistic schedule admits far fewer records than the guaranteed one
    perm = rng.permutation(n)
    xs = x[perm]
    ix2 = _index(xs, mode=L.MODE_FUSED, warm_rows=4096)
    s2, l2 = ix2.search(q, 100)
    st2 = ix2.last_stats()
    assert st2['overflowed_queries'] == 0, st2
    ix3 = _index(xs, mode=L.MODE_FUSED, warm_rows=4096)
    ix3.set_option(L.OPT_OPTIMISTIC, 0)
    s3, l3 = ix3.search(q, 100)
    st3 = ix3.last_stats()
    np.testing.assert_array_equal(l2, l3)
    np.testing.assert_array_equal(s2, s3)
    assert st2['fused_candidates'] < 0.6 * st3['fused_candidates'], (st2, st3)
    # ... and the same answer as in the front-loaded order: identical fp32 scores; identical rows wherever a score is unique in its list
    # (equal scores are ordered by row number, which the permutation changes)
    np.testing.assert_array_equal(s2, s)
    uniq = np.ones_like(s, dtype=bool)
    uniq[:, 1:] &= s[:, 1:] != s[:, :-1]
    uniq[:, :-1] &= s[:, :-1] != s[:, 1:]
    np.testing.assert_array_equal(perm[l2][uniq], l[uniq])


def test_scrambled_scan_order_on_cluster_sorted_rows(L):
    """LDOT_OPT_SCAN_ORDER.  Rows stored SORTED BY CLUSTER are not a fair sample of the index in storage order: a query's best rows
    arrive together, its optimistic thresholds aim too high or too low and candidate pools overflow.  The scrambled scan (row tiles of the
    whole index in a pseudo-random order + a warm-up on a spread sample) makes the scan order independent of the storage order: nothing
    is flagged, far fewer records are admitted, and the lists equal those of the storage-order scan and of the dense path bit for bit.
    Default (auto): storage order until one query in a thousand of a search fails the check, scrambled from then on."""
    rng = np.random.default_rng(5)
    n, d, nq, k = 160000, 64, 600, 50
    cent = rng.standard_normal((40, d)).astype(np.float32)
    assign = np.sort(rng.integers(0, 40, n))                       # 40 contiguous clusters of ~4000 rows
    x = (cent[assign] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
    q = (x[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, d))).astype(np.float32)
    ref = _index(x, mode=L.MODE_DENSE)
    sd, ld = ref.search(q, k)
    seq = _index(x, mode=L.MODE_FUSED, scan_order=1)
    s1, l1 = seq.search(q, k)
    st1 = seq.last_stats()
    scr = _index(x, mode=L.MODE_FUSED, scan_order=2)
    s2, l2 = scr.search(q, k)
    st2 = scr.last_stats()
    for s_, l_ in ((s1, l1), (s2, l2)):
        np.testing.assert_array_equal(l_, ld)
        np.testing.assert_array_equal(s_, sd)
    assert_topk_matches(q, x, s2, l2, k)
    assert st1['overflowed_queries'] > nq // 64, st1              # storage order: the check fails for many queries
    assert st2['overflowed_queries'] == 0, st2                    # scrambled order: for none
    assert st2['fused_candidates'] < 0.5 * st1['fused_candidates'], (st1, st2)
    assert st2['fused_pairs'] >= nq * n                           # every row went through the filter (the sample's rows again)
    auto = _index(x, mode=L.MODE_FUSED)
    sa, la = auto.search(q, k)
    assert auto.last_stats()['overflowed_queries'] > nq // 64
    sb, lb = auto.search(q, k)                                     # ... which switched the index to the scrambled order
    stb = auto.last_stats()
    assert stb['overflowed_queries'] == 0 and stb['fused_candidates'] == st2['fused_candidates'], stb
    for s_, l_ in ((sa, la), (sb, lb)):
        np.testing.assert_array_equal(l_, ld)
        np.testing.assert_array_equal(s_, sd)
    # rows in random order: both orders flag nothing and return the same lists
    perm = rng.permutation(n)
    a = _index(x[perm], mode=L.MODE_FUSED, scan_order=1)
    b = _index(x[perm], mode=L.MODE_FUSED, scan_order=2)
    sa, la = a.search(q, k)
    sb, lb = b.search(q, k)
    np.testing.assert_array_equal(la, lb)
    np.testing.assert_array_equal(sa, sb)
    assert a.last_stats()['overflowed_queries'] == 0 and b.last_stats()['overflowed_queries'] == 0


def test_rows_sorted_in_short_runs_and_the_row_shuffle_of_the_indexer(L):
    """Rows sorted by a fine clustering (runs of ~250 similar rows: a run IS a 384-row tile) keep a query's whole top k' in one or two
    tiles whatever the tile order: the optimistic check fails in both scan orders and the index ends on guaranteed thresholds — exact,
    but with flagged queries.  DenseFlatIndexer(shuffle_seed=) stores the rows (and their ids) in a pseudo-random order: nothing is
    flagged from the first search on, and search_knn returns the same external ids and scores."""
    from lightningdot_amd.indexer import DenseFlatIndexer
    rng = np.random.default_rng(9)
    n, d, nq, k = 200000, 64, 600, 50
    cent = rng.standard_normal((800, d)).astype(np.float32)
    assign = np.sort(rng.integers(0, 800, n))                      # 800 contiguous clusters of ~250 rows
    x = (cent[assign] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
    q = (x[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, d))).astype(np.float32)
    ids = [f'row{i}' for i in range(n)]
    plain = DenseFlatIndexer(d)
    plain.index.set_option(L.OPT_MODE, L.MODE_FUSED)
    plain.index_tensor(ids, x)
    res_plain = plain.search_knn(q, k)
    assert plain.index.last_stats()['overflowed_queries'] > 0      # storage order: flagged queries (searched again: still exact)
    shuf = DenseFlatIndexer(d, shuffle_seed=3)
    shuf.index.set_option(L.OPT_MODE, L.MODE_FUSED)
    shuf.index_tensor(ids, x)
    assert shuf.index_id_to_db_id != ids and sorted(shuf.index_id_to_db_id) == sorted(ids)
    res_shuf = shuf.search_knn(q, k)
    assert shuf.index.last_stats()['overflowed_queries'] == 0      # a fair order from the first search on
    for (ia, sa), (ib, sb) in zip(res_plain, res_shuf):
        np.testing.assert_array_equal(sa, sb)                      # the same fp32 scores
        uniq = np.ones(k, dtype=bool)
        uniq[1:] &= sa[1:] != sa[:-1]
        uniq[:-1] &= sa[:-1] != sa[1:]
        assert [a for a, u in zip(ia, uniq) if u] == [b for b, u in zip(ib, uniq) if u]   # the same ids wherever a score is unique
    # index_data (host tuples) shuffles too, with the ids
    small = DenseFlatIndexer(d, shuffle_seed=3)
    small.index_data([(ids[i], x[i]) for i in range(3000)])
    ref = DenseFlatIndexer(d)
    ref.index_data([(ids[i], x[i]) for i in range(3000)])
    assert small.index_id_to_db_id != ref.index_id_to_db_id
    assert [r[0][0] for r in small.search_knn(q[:20], 1)] == [r[0][0] for r in ref.search_knn(q[:20], 1)]


def test_no_rescore_reports_bf16_input_scores(L):
    rng = np.random.default_rng(12)
    x = rng.standard_normal((4000, 768)).astype(np.float32)
    q, g = planted_queries(x, 64)
    ix = _index(x, rescore=0)
    s, l = ix.search(q, 10)
    assert (l[:, 0] == g).all()
    full = q.astype(np.float64) @ x.astype(np.float64).T
    err = np.abs(s - np.take_along_axis(full, l, axis=1))
    assert err.max() < 1.0 and err.max() > 1e-4      # bf16-input error is visible, fp32 re-score removes it


def test_g7_golden_indexer_wrapper(L, golden_dir):
    from lightningdot_amd.indexer import DenseFlatIndexer
    g = json.load(open(os.path.join(golden_dir, 'g7_indexer.json')))
    a = np.load(os.path.join(golden_dir, 'g7_indexer_inputs.npz'))
    data = list(zip(g['ids'], a['x']))
    for k, res in g['results'].items():
        ix = DenseFlatIndexer(a['x'].shape[1], buffer_size=20)
        ix.index_data(data)
        out = ix.search_knn(a['q'], int(k))
        assert [r[0] for r in out] == res['ids']          # incl. the k > ntotal -> last-id behaviour
        np.testing.assert_allclose(np.stack([r[1] for r in out]), np.asarray(res['scores'], np.float32),
                                   rtol=0, atol=1e-4)
        assert out[0][1].dtype == np.float32


def test_serialize_roundtrip(L, tmp_path):
    from lightningdot_amd.indexer import DenseFlatIndexer
    rng = np.random.default_rng(13)
    x = rng.standard_normal((777, 40)).astype(np.float32)
    ix = DenseFlatIndexer(40)
    ix.index_data([(f'k{i}', x[i]) for i in range(len(x))])
    f = str(tmp_path / 'idx')
    ix.serialize(f)
    assert os.path.exists(f + '.index.dpr') and os.path.exists(f + '.index_meta.dpr')
    ix2 = DenseFlatIndexer(40)
    ix2.deserialize_from(f)
    assert ix2.index.ntotal == 777 and ix2.index_id_to_db_id == ix.index_id_to_db_id
    q = rng.standard_normal((9, 40)).astype(np.float32)
    r1, r2 = ix.search_knn(q, 7), ix2.search_knn(q, 7)
    assert [a[0] for a in r1] == [a[0] for a in r2]
    np.testing.assert_array_equal(np.stack([a[1] for a in r1]), np.stack([a[1] for a in r2]))
    ix2.index_id_to_db_id.pop()
    with open(f + '.index_meta.dpr', 'wb') as fh:
        import pickle
        pickle.dump(ix2.index_id_to_db_id, fh)
    with pytest.raises(AssertionError):
        DenseFlatIndexer(40).deserialize_from(f)


def test_merge_topk(L):
    import ctypes
    rng = np.random.default_rng(14)
    x = rng.standard_normal((5000, 64)).astype(np.float32)
    q = rng.standard_normal((70, 64)).astype(np.float32)
    k = 40
    parts_s, parts_l = [], []
    bounds = [(0, 1500), (1500, 1530), (1530, 5000)]      # a shard smaller than k -> padded partial list
    for a, b in bounds:
        ix = _index(x[a:b])
        s, l = ix.search(q, k)
        parts_s.append(s)
        parts_l.append(np.where(l >= 0, l + a, -1))
    S = np.ascontiguousarray(np.stack(parts_s)).astype(np.float32)
    Lb = np.ascontiguousarray(np.stack(parts_l)).astype(np.int64)
    out_s = np.empty((70, k), np.float32)
    out_l = np.empty((70, k), np.int64)
    lib = L.load_library()
    L.check(lib.ldot_merge_topk(S.ctypes.data, Lb.ctypes.data, 3, 70, k, k, out_s.ctypes.data, out_l.ctypes.data,
                                L.HOST, None))
    ms, ml = O.merge_topk(list(zip(parts_s, parts_l)), k)
    np.testing.assert_array_equal(out_l, ml)
    np.testing.assert_array_equal(out_s, ms)
    whole_s, whole_l = _index(x).search(q, k)
    np.testing.assert_array_equal(out_l, whole_l)
    np.testing.assert_array_equal(out_s, whole_s)


def test_cls_pool(L):
    import ctypes
    import torch
    lib = L.load_library()
    B, Ls, D = 9, 21, 768
    seq = torch.randn(B, Ls, D, device='cuda')
    for dt, code in [(torch.float32, L.F32), (torch.bfloat16, L.BF16), (torch.float16, L.F16)]:
        s = seq.to(dt).contiguous()
        o32 = torch.empty(B, D, device='cuda')
        o16 = torch.empty(B, D, device='cuda', dtype=torch.bfloat16)
        L.check(lib.ldot_cls_pool(s.data_ptr(), code, B, Ls * D, D, 0, o32.data_ptr(), o16.data_ptr(), None))
        torch.cuda.synchronize()
        ref = O.cls_pool(s.float().cpu().numpy())
        np.testing.assert_array_equal(o32.cpu().numpy(), ref)
        np.testing.assert_array_equal(o16.float().cpu().numpy(), torch.from_numpy(ref).bfloat16().float().numpy())
        L.check(lib.ldot_cls_pool(s.data_ptr(), code, B, Ls * D, D, 1, o32.data_ptr(), None, None))
        torch.cuda.synchronize()
        np.testing.assert_allclose(o32.cpu().numpy(), O.cls_pool(s.float().cpu().numpy(), l2_normalize=True),
                                   rtol=2e-7, atol=1e-8)


def test_s1_reduced_planted(L):
    """SURVEY §8d S1 at 1/5 scale: 200k x 768 index, 2048 planted queries, top-100, fused path."""
    rng = np.random.default_rng(1234)
    n, d, nq, k = 200_000, 768, 2048, 100
    x = rng.standard_normal((n, d), dtype=np.float32)
    q, g = planted_queries(x, nq)
    ix = _index(x)
    s, l = ix.search(q, k)
    st = ix.last_stats()
    assert st['fused_pairs'] > 0 and st['overflowed_queries'] == 0, st
    assert (l[:, 0] == g).all()                                   # Recall@1 == 1 on planted data
    assert (np.diff(s.astype(np.float64), axis=1) <= 0).all()
    sub = slice(0, 96)
    assert_topk_matches(q[sub], x, s[sub], l[sub], k)


@pytest.mark.parametrize('nq,n,d,k', [(1, 150_000, 128, 100), (7, 70_001, 768, 10), (256, 200_000, 64, 100)])
def test_few_queries_wide_path(L, nq, n, d, k):
    """Single-query serving shape (dvl/utils.py:204-211): one query tile against many rows takes the wide dense path
    (one score launch, segmented select, merge)."""
    rng = np.random.default_rng(n + nq)
    x = rng.standard_normal((n, d)).astype(np.float32)
    q, g = planted_queries(x, nq)
    ix = _index(x)
    s, l = ix.search(q, k)
    assert (l[:, 0] == g).all()
    assert_topk_matches(q, x, s, l, k)
    # incremental: rows added after a search are visible to the next one
    x2 = rng.standard_normal((500, d)).astype(np.float32)
    x2[7] = q[0] * 3.0
    ix.add(x2)
    s2, l2 = ix.search(q[:1], k)
    assert l2[0, 0] == n + 7


@pytest.mark.parametrize('nq,n,d,k', [
    (1, 123_287, 768, 100),    # the demo's image index size (dvl/utils.py:204-211), 24 slabs -> 8 blocks in flight
    (16, 40_000, 768, 100),    # the widest narrow batch
    (3, 2_500, 64, 10),        # 2 slabs, fewer groups than waves
    (5, 33_333, 96, 100),      # d padded to 128: 4 slabs; ragged last 16-row group
    (2, 20, 768, 100),         # k > ntotal, one partial group (below 2048 rows: the plain dense path)
    (9, 4_200_000, 64, 50),    # more rows than one wide chunk (4M): partial lists of two chunks are merged
    (17, 50_000, 128, 100),    # two groups of 16 queries per scan
    (40, 123_287, 768, 50),    # four groups (512-thread workgroups, 96 KiB of query operand in LDS)
    (64, 300_000, 96, 10),     # the widest narrow batch; runs of 32 rows
])
def test_narrow_scan_up_to_64_queries(L, nq, n, d, k):
    """<= 64 queries take the HBM-speed narrow scan (score_narrow.hip) in AUTO mode at every index size; the forced fused scan
    must return the same lists."""
    rng = np.random.default_rng(n * 3 + nq)
    x = rng.standard_normal((n, d)).astype(np.float32)
    q, g = planted_queries(x, nq)
    ix = _index(x)
    s, l = ix.search(q, k)
    if n > k:
        assert (l[:, 0] == g).all()
    assert_topk_matches(q, x, s, l, k)
    st = ix.last_stats()
    assert st['fused_pairs'] == 0 and st['dense_pairs'] == n * nq
    if n >= 70_000:
        ix.set_option(L.OPT_MODE, L.MODE_FUSED)
        s2, l2 = ix.search(q, k)
        np.testing.assert_array_equal(l, l2)
        np.testing.assert_array_equal(s, s2)


def test_narrow_search_full_candidate_buffer_falls_back(L):
    """Rows stored as runs of 64 identical copies: a run maximum stands for 64 equal scores, the candidate buffer (16384 keys) of the
    narrow search fills up, the overflow is reported (stats) and the search is redone with the streaming selector — exact."""
    rng = np.random.default_rng(11)
    base = rng.standard_normal((5_000, 64)).astype(np.float32)
    x = np.repeat(base, 64, axis=0)                      # 320 000 rows -> runs of 32 rows, every run constant
    q = (base[[7, 4_001]] + 0.3 * rng.standard_normal((2, 64))).astype(np.float32)
    ix = _index(x)
    s, l = ix.search(q, 200)
    st = ix.last_stats()
    assert st['overflowed_queries'] == 2
    assert_topk_matches(q, x, s, l, 200)
    assert (l[0, :64] == np.arange(7 * 64, 8 * 64)).all()          # ties: ascending row order
    # the index backs off from the narrow search for a while (streaming selector: no overflow to report), results unchanged
    s1, l1 = ix.search(q, 200)
    assert ix.last_stats()['overflowed_queries'] == 0
    np.testing.assert_array_equal(l, l1)
    np.testing.assert_array_equal(s, s1)
    # the same through the route that does not stage the queries (fp32 device queries, kernel-written outputs: the scan converts them
    # itself, the recovery stages them after the fact)
    import torch
    ix3 = _index(x)
    s3, l3 = ix3.search_tensors(torch.from_numpy(q).cuda(), 200)
    assert ix3.last_stats()['overflowed_queries'] == 2
    np.testing.assert_array_equal(l3.cpu().numpy(), l)
    np.testing.assert_array_equal(s3.cpu().numpy(), s)
    # an ordinary search afterwards is clean again (counters were left zero)
    x2 = rng.standard_normal((50_000, 64)).astype(np.float32)
    ix2 = _index(x2)
    q2, g2 = planted_queries(x2, 3)
    for _ in range(2):
        s2, l2 = ix2.search(q2, 10)
        assert (l2[:, 0] == g2).all() and ix2.last_stats()['overflowed_queries'] == 0


def test_narrow_scan_falls_back_when_rows_are_too_long(L):
    """split-bf16 rows of d = 768 are 72 slabs (> 64 KiB of query operand in LDS): the ring engine handles them"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((9_000, 768)).astype(np.float32)
    q, g = planted_queries(x, 4)
    ix = _index(x, precision=1)
    s, l = ix.search(q, 20)
    assert (l[:, 0] == g).all()
    assert_topk_matches(q, x, s, l, 20)


@pytest.mark.gpu
@pytest.mark.parametrize('n,mode', [(5000, 'dense'), (60000, 'fused')])
def test_split_bf16_precision_on_crowded_scores(L, n, mode):
    """Scores that crowd closer than bf16 resolves (rows = one large vector + 1e-3 perturbations): the accuracy contract of
    the default bf16 candidate pass does not cover this; LDOT_OPT_PRECISION = 1 (split-bf16 operands) does."""
    rng = np.random.default_rng(77)
    d, nq, k = 128, 300, 10
    u = rng.standard_normal(d).astype(np.float32)
    x = (u[None, :] + 1e-3 * rng.standard_normal((n, d))).astype(np.float32)
    q = (u[None, :] + 1e-3 * rng.standard_normal((nq, d))).astype(np.float32)
    ix = _index(x, mode=L.MODE_DENSE if mode == 'dense' else L.MODE_FUSED)
    ix.set_option(L.OPT_PRECISION, 1)          # rebuilds the shadow from the fp32 master copy
    s, l = ix.search(q, k)
    if mode == 'fused':
        assert ix.last_stats()['fused_pairs'] > 0
    assert_topk_matches(q, x, s, l, k, eps=2e-4)
    # the default precision is expected to miss true neighbours here (documents the limit the option exists for)
    ix.set_option(L.OPT_PRECISION, 0)
    s0, l0 = ix.search(q, k)
    full = q.astype(np.float64) @ x.astype(np.float64).T
    kth = np.sort(full, axis=1)[:, -k][:, None]
    missed = (np.take_along_axis(full, l0, axis=1) < kth - 2e-4).any(axis=1).mean()
    assert missed > 0.2, missed
    # and switching back restores exact results
    ix.set_option(L.OPT_PRECISION, 1)
    s1, l1 = ix.search(q, k)
    np.testing.assert_array_equal(l1, l)
    np.testing.assert_array_equal(s1, s)


@pytest.mark.gpu
def test_reset_then_refill_and_growth(L):
    """reset() clears all three copies of the rows (fp32 master, row-major and blocked bf16 shadows); capacity growth copies
    them (the blocked shadow in whole 16-row blocks, also when ntotal is not a multiple of 16)."""
    rng = np.random.default_rng(3)
    d, k = 64, 20
    x1 = rng.standard_normal((40003, d)).astype(np.float32)
    x2 = (rng.standard_normal((41001, d)) * 0.5).astype(np.float32)
    q = rng.standard_normal((300, d)).astype(np.float32)
    ix = _index(x1, mode=L.MODE_FUSED)
    ix.search(q, k)
    ix.reset()
    assert ix.ntotal == 0
    s, l = ix.search(q, k)
    assert (l == -1).all()
    ix.add(x2[:7])                     # partly filled first block
    ix.add(x2[7:20011])
    ix.add(x2[20011:])                 # crosses the old capacity: growth with copy
    assert ix.ntotal == x2.shape[0]
    s, l = ix.search(q, k)
    assert ix.last_stats()['fused_pairs'] > 0
    assert_topk_matches(q, x2, s, l, k)
    ix.set_option(L.OPT_RESERVE_ROWS, 200000)   # explicit reservation: the next add does not reallocate
    more = rng.standard_normal((60000, d)).astype(np.float32)
    ix.add(more)
    s, l = ix.search(q, k)
    assert_topk_matches(q, np.concatenate([x2, more]), s, l, k)


@pytest.mark.gpu
def test_search_begin_finish_with_floor(L):
    """The two-halves search of the sharded index: thresholds out, floor in.  With floor = None (or the shard's own
    threshold) the result equals ldot_index_search; a higher floor drops exactly the candidates below it."""
    import torch
    rng = np.random.default_rng(21)
    n, d, nq, k = 50000, 64, 300, 20
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ix = _index(x)
    qt = torch.from_numpy(q).cuda()
    s_ref, l_ref = ix.search_tensors(qt, k)
    tau = ix.search_begin(qt, k)
    assert tau.shape == (nq,) and torch.isfinite(tau).all()
    s0, l0 = ix.search_finish(None)
    assert torch.equal(s0, s_ref) and torch.equal(l0, l_ref)
    ix.search_begin(qt, k)
    s1, l1 = ix.search_finish(tau)                      # own threshold: nothing is dropped
    assert torch.equal(s1, s_ref) and torch.equal(l1, l_ref)
    # a floor between the 5th and 6th exact score keeps (about) the five best: bf16 candidate scores differ from the exact
    # ones by ~1e-2 here, so compare through the exact scores with that slack
    floor = ((s_ref[:, 4] + s_ref[:, 5]) * 0.5).contiguous()
    ix.search_begin(qt, k)
    s2, l2 = ix.search_finish(floor)
    kept = (l2 >= 0).sum(1)
    assert (kept >= 3).all() and (kept <= 8).all(), kept
    for j in range(nq):
        m = int(kept[j])
        assert torch.equal(l2[j, :min(m, 3)], l_ref[j, :min(m, 3)])
        assert (l2[j, m:] == -1).all()
    # dense path (small index) maintains the thresholds too
    ixs = _index(x[:3000])
    t2 = ixs.search_begin(qt, k)
    ss, ll = ixs.search_finish(t2)
    s3, l3 = ixs.search_tensors(qt, k)
    assert torch.isfinite(t2).all() and torch.equal(ss, s3) and torch.equal(ll, l3)


def test_few_query_search_all_output_routes_agree(L):
    """<= 64 queries end in ONE kernel after the index scan (narrow_finish_kernel; <= 16 fp32 device queries are not even staged).  Its three uses — results written directly
    (device outputs / pinned host outputs), list + threshold only followed by the ordinary re-score (pageable host outputs, the
    begin / finish halves of a sharded search) — and the streaming-selector path (MODE_DENSE) give identical results; 70 000 rows
    = runs of 64 rows, 300 000 rows = runs of 256 rows."""
    import torch
    for n, nq, k in ((70000, 1, 100), (70000, 16, 10), (300000, 5, 100), (70000, 17, 100), (150000, 64, 20), (70000, 33, 100)):
        rng = np.random.default_rng(n + nq)
        x = rng.standard_normal((n, 64)).astype(np.float32)
        q, g = planted_queries(x, nq)
        ix = _index(x)
        qt = torch.from_numpy(q).cuda()
        s_dev, l_dev = ix.search_tensors(qt, k)                              # direct, device outputs
        hs, hl = torch.empty((nq, k)).pin_memory(), torch.empty((nq, k), dtype=torch.int64).pin_memory()
        ix.search_into(qt, k, hs, hl)                                        # direct, pinned host outputs
        s_np, l_np = ix.search(q, k)                                         # pageable host outputs: list only + re-score kernel
        tau = ix.search_begin(qt, k)                                         # the two halves
        s_bf, l_bf = ix.search_finish(None)
        ixd = _index(x, mode=L.MODE_DENSE)
        s_d, l_d = ixd.search(q, k)
        assert (l_np[:, 0] == g).all() and torch.isfinite(tau).all()
        assert_topk_matches(q, x, s_np, l_np, k)
        for s_, l_ in ((s_dev.cpu().numpy(), l_dev.cpu().numpy()), (hs.numpy(), hl.numpy()), (s_bf.cpu().numpy(), l_bf.cpu().numpy()),
                       (s_d, l_d)):
            np.testing.assert_array_equal(l_, l_np)
            np.testing.assert_array_equal(s_, s_np)
        st = ix.last_stats()
        assert st['overflowed_queries'] == 0 and st['fused_pairs'] == 0


@pytest.mark.gpu
def test_hnsw_indexer_surface_is_exact_backed(L, tmp_path):
    """DenseHNSWFlatIndexer (the reference's --hnsw_index alternative, faiss_indexers.py:90-154): same surface and score
    semantics (squared L2 of the phi-augmented vectors, ascending), exact neighbours."""
    from lightningdot_amd.indexer import DenseHNSWFlatIndexer
    rng = np.random.default_rng(12)
    n, d, nq, k = 700, 48, 40, 15
    x = (rng.standard_normal((n, d)) * rng.uniform(0.5, 2.0, (n, 1))).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    data = [('id%d' % i, x[i]) for i in range(n)]
    ix = DenseHNSWFlatIndexer(d, buffer_size=256)
    ix.index_data(data)
    oc = O.DenseHNSWFlatIndexerOracle(d, buffer_size=256)
    oc.index_data(data)
    got, exp = ix.search_knn(q, k), oc.search_knn(q, k)
    for (gi, gs), (ei, es) in zip(got, exp):
        assert gi == ei
        np.testing.assert_allclose(gs, es, rtol=1e-4, atol=1e-3)
        assert (np.diff(gs) >= -1e-4).all()                      # ascending distances
    # serialisation keeps the id list and phi; indexing again afterwards is refused like the reference (:152-154,112-113)
    f = str(tmp_path / 'hn')
    ix.serialize(f)
    ix2 = DenseHNSWFlatIndexer(d)
    ix2.deserialize_from(f)
    got2 = ix2.search_knn(q, k)
    for (gi, gs), (hi, hs) in zip(got, got2):
        assert gi == hi
        np.testing.assert_allclose(gs, hs, rtol=1e-6, atol=1e-5)
    with pytest.raises(RuntimeError):
        ix2.index_data(data[:3])


@pytest.mark.gpu
def test_two_handles_from_two_threads(L):
    """Handles are independent (no global mutable state beyond the thread-local error string): two threads search their own
    indexes on their own streams concurrently and get the same results as alone."""
    import threading
    import torch
    rng = np.random.default_rng(4)
    xs = [rng.standard_normal((40000, 64)).astype(np.float32) for _ in range(2)]
    qs = [rng.standard_normal((700, 64)).astype(np.float32) for _ in range(2)]
    ixs = [_index(x) for x in xs]
    ref = [ix.search(q, 20) for ix, q in zip(ixs, qs)]
    out, errs = [None, None], []

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(8):
                    out[i] = ixs[i].search(qs[i], 20)
        except Exception as e:   # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for i in range(2):
        np.testing.assert_array_equal(out[i][1], ref[i][1])
        np.testing.assert_array_equal(out[i][0], ref[i][0])


def test_verify_flags_crowded_scores_and_escalates(L):
    """LDOT_OPT_VERIFY: well separated data is proven exact; scores that crowd closer than bf16 resolves are FLAGGED, the flagged
    queries are re-searched with a larger margin (FlatIPIndex.search(verify=True)) and what remains unproven is reported."""
    rng = np.random.default_rng(21)
    x = rng.standard_normal((20000, 256)).astype(np.float32)
    q, g = planted_queries(x, 300)
    ix = _index(x)
    s, l = ix.search(q, 10, verify=True)
    assert ix.last_unproven == 0 and ix.last_escalations == []
    assert_topk_matches(q, x, s, l, 10)
    # crowded: every row = the same large vector + small noise -> all scores within bf16 noise of each other
    base = (rng.standard_normal(256) * 8).astype(np.float32)
    xc = (base[None, :] + 0.002 * rng.standard_normal((20000, 256))).astype(np.float32)
    qc = (base[None, :] + 0.002 * rng.standard_normal((40, 256))).astype(np.float32)
    ixc = _index(xc)
    ixc.set_option(L.OPT_VERIFY, 1)
    ixc.search(qc, 10)
    flags, cnt = ixc.unproven(40)
    assert cnt == 40 and flags.all()                      # nothing can be proven on this data with the default margin
    ixc.set_option(L.OPT_VERIFY, 0)
    ixc.search(qc, 10, verify=True)
    assert ixc.last_escalations and ixc.last_escalations[0][1] == 40       # all 40 were re-searched with a larger margin
    # settings made through set_option survive a verified search, and the escalation starts from the margin in force
    ixc.set_option(L.OPT_MARGIN, 200)
    ixc.set_option(L.OPT_VERIFY, 1)
    ixc.search(qc, 10, verify=True)
    assert ixc.last_escalations[0][0] == 800
    assert ixc._opts[L.OPT_MARGIN] == 200 and ixc._opts[L.OPT_VERIFY] == 1
    ixc.search(qc, 10)
    assert ixc.unproven(40)[1] == 40                       # verification is still on (the user's setting), margin 200 again


def test_last_stats_counts_filter_records(L):
    rng = np.random.default_rng(22)
    x = rng.standard_normal((60000, 128)).astype(np.float32)
    q, g = planted_queries(x, 600)
    ix = _index(x, mode=L.MODE_FUSED, warm_rows=2048)
    ix.search(q, 10)
    st = ix.last_stats()
    assert st['fused_pairs'] > 0 and st['overflowed_queries'] == 0
    # every query needs at least k' - (warm-up hits) records to fill its list; far fewer than one per scored pair
    assert 600 * 5 < st['fused_candidates'] < st['fused_pairs'] // 20, st


def test_search_into_pinned_and_pageable_outputs_agree(L):
    """Host outputs: pinned (device-mapped) buffers are written by the re-score kernel directly, pageable ones through a staging
    buffer — same results; shape / dtype of the buffers is checked."""
    import torch
    rng = np.random.default_rng(31)
    x = rng.standard_normal((40000, 128)).astype(np.float32)
    q, g = planted_queries(x, 700)
    ix = _index(x)
    qd = torch.from_numpy(q).cuda()
    ps = torch.empty((700, 20), dtype=torch.float32).pin_memory()
    pl = torch.empty((700, 20), dtype=torch.int64).pin_memory()
    ix.search_into(qd, 20, ps, pl)
    ns, nl = np.empty((700, 20), np.float32), np.empty((700, 20), np.int64)
    ix.search_into(qd, 20, ns, nl)
    np.testing.assert_array_equal(ps.numpy(), ns)
    np.testing.assert_array_equal(pl.numpy(), nl)
    assert_topk_matches(q, x, ns, nl, 20)
    with pytest.raises(ValueError):
        ix.search_into(qd, 20, np.empty((700, 19), np.float32), nl)
    with pytest.raises(ValueError):
        ix.search_into(qd, 20, ns.astype(np.float64), nl)


def test_error_paths_of_the_new_entry_points(L, tmp_path):
    import ctypes
    rng = np.random.default_rng(32)
    x = rng.standard_normal((300, 32)).astype(np.float32)
    ix = _index(x)
    ix.search(x[:4], 5)
    with pytest.raises(L.LdotError) as e:            # LDOT_OPT_VERIFY is off
        ix.unproven(4)
    assert e.value.code == -5
    # a corrupt index file (header row count does not match the payload) is rejected, not allocated
    from lightningdot_amd.indexer import FlatIPIndex
    p = str(tmp_path / 'ix.bin')
    ix.save(p)
    raw = bytearray(open(p, 'rb').read())
    raw[12:20] = (10 ** 12).to_bytes(8, 'little')    # n
    open(p, 'wb').write(raw)
    with pytest.raises(L.LdotError) as e:
        FlatIPIndex.load(p)
    assert e.value.code == -4
    # search_finish without a pending search_begin
    with pytest.raises(L.LdotError):
        ix.search_finish()
    # a tensor on the wrong device is refused before it reaches the library
    import torch
    if torch.cuda.device_count() > 1:
        with pytest.raises(ValueError):
            ix.search_tensors(torch.zeros(2, 32, device='cuda:1'), 3)


@pytest.mark.parametrize('nq,n,d,k,mode', [
    (100, 30000, 64, 1000, 'DENSE'),     # one chunk longer than the survivor buffer: the sample pivot (mining shape, dvl/hn.py:53 at num_tops 1000)
    (60, 70000, 64, 1000, 'DENSE'),      # three dense chunks: the running list joins the selection
    (40, 40000, 32, 2048, 'DENSE'),      # k' = 2560: the 8192-slot variant, two chunks
    (300, 150000, 64, 600, 'FUSED'),     # fused scan with k' = 768: long-list pool select
    (300, 150000, 64, 1000, 'AUTO'),     # ... k' = 1280 (the image -> text mining search's shape, reduced)
    (4, 9000, 64, 700, 'DENSE'),         # a handful of queries
])
def test_long_lists_match_brute_force(L, nq, n, d, k, mode):
    """k' > 512: select_big.hip (pivot + one streaming pass + bit search in registers) behind the dense and the fused path"""
    rng = np.random.default_rng(n + k + nq)
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ix = _index(x, mode=getattr(L, 'MODE_' + mode))
    s, l = ix.search(q, k)
    assert_topk_matches(q, x, s, l, k)


@pytest.mark.parametrize('mode', ['DENSE', 'FUSED'])
def test_long_lists_with_thousands_of_equal_scores_take_the_slow_path(L, mode):
    """10 000 copies of one row + distinct rows: every copy scores the same, the survivors of any pivot overflow the buffer and the selection
    falls back to the streamed 64-bit bit search; ties go to the lower label (oracle.FlatIP's order)."""
    rng = np.random.default_rng(11)
    d, k = 64, 800
    base = rng.standard_normal((1, d)).astype(np.float32)
    x = np.concatenate([rng.standard_normal((30000, d)).astype(np.float32), np.repeat(base, 10000, axis=0),
                        rng.standard_normal((60000, d)).astype(np.float32)])
    q = np.concatenate([base + 0.01 * rng.standard_normal((1, d)).astype(np.float32) for _ in range(20)] +
                       [rng.standard_normal((280, d)).astype(np.float32)])
    ix = _index(x, mode=getattr(L, 'MODE_' + mode), optimistic=0)
    s, l = ix.search(q, k)
    so = O.FlatIP(d)
    so.add(x)
    so_s, so_l = so.search(q, k)
    np.testing.assert_allclose(s, so_s, rtol=0, atol=1e-3)
    # the 20 queries next to the repeated row: the copies fill the top k in label order
    assert np.array_equal(l[:20], np.tile(np.arange(30000, 30000 + k), (20, 1)))
    assert (l[20:, 0] == so_l[20:, 0]).all()
