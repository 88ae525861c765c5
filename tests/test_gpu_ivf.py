"""GPU tests of the approximate (inverted-file) index behind the reference's --hnsw_index surface
(dvl/indexer/faiss_indexers.py:90-154): exactness when every list is probed, recall on clustered data, L2-distance surface,
persistence, the raw C entry point against a brute-force scan of the probed ranges."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _clustered(n, d, ncl, seed, spread=0.35):
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((ncl, d)).astype(np.float32)
    x = (cent[rng.integers(0, ncl, n)] + spread * rng.standard_normal((n, d))).astype(np.float32)
    return x, rng


def test_probing_every_list_is_the_exact_search():
    import torch
    from lightningdot_amd.indexer import DenseFlatIndexer
    from lightningdot_amd.ivf import DenseIVFFlatIndexer
    x, rng = _clustered(20000, 96, 40, 1)
    q = (x[rng.integers(0, 20000, 300)] + 0.3 * rng.standard_normal((300, 96))).astype(np.float32)
    ids = [f'row{i}' for i in range(20000)]
    ivf = DenseIVFFlatIndexer(96, nlist=50, nprobe=50)
    ivf.index_tensor(ids, torch.from_numpy(x))
    assert int(ivf.list_offsets[-1]) == 20000 and len(ivf.index_id_to_db_id) == 20000
    flat = DenseFlatIndexer(96)
    flat.index_tensor(ids, torch.from_numpy(x).cuda())
    got, want = ivf.search_knn(q, 25), flat.search_knn(q, 25)
    full = q.astype(np.float64) @ x.astype(np.float64).T
    for (gi, gs), (wi, ws), frow in zip(got, want, full):
        np.testing.assert_allclose(gs, ws, rtol=0, atol=2e-4)
        assert gi[0] == wi[0]
        # same id set except inside fp32 summation-order ties
        if gi != wi:
            sc = {i: frow[int(i[3:])] for i in set(gi) ^ set(wi)}
            assert max(sc.values()) - min(sc.values()) < 1e-3
    # fewer rows than k in the probed lists -> padded like faiss (-1 labels map to the LAST id, :85)
    tiny = DenseIVFFlatIndexer(8, nlist=2, nprobe=1)
    tiny.index_tensor(['a', 'b', 'c'], torch.eye(3, 8))
    r = tiny.search_knn(np.eye(1, 8, dtype=np.float32), 5)
    assert len(r[0][0]) == 5 and r[0][1][-1] == np.float32(-3.4028234663852886e38)


def test_recall_on_clustered_data_and_hnsw_surface():
    import torch
    from lightningdot_amd.indexer import DenseFlatIndexer, DenseHNSWFlatIndexer
    x, rng = _clustered(60000, 128, 300, 2)
    q = (x[rng.integers(0, 60000, 500)] + 0.2 * rng.standard_normal((500, 128))).astype(np.float32)
    ids = list(range(60000))
    flat = DenseFlatIndexer(128)
    flat.index_tensor(ids, torch.from_numpy(x).cuda())
    want = flat.search_knn(q, 10)
    approx = DenseHNSWFlatIndexer(128, ef_search=64, approximate=True)        # probes 16 of ~245 lists
    approx.index_tensor(ids, torch.from_numpy(x))
    got = approx.search_knn(q, 10)
    recall = np.mean([len(set(g[0]) & set(w[0])) / 10.0 for g, w in zip(got, want)])
    assert recall > 0.9, recall
    assert np.mean([g[0][0] == w[0][0] for g, w in zip(got, want)]) > 0.95
    # the surface's score semantics: squared L2 distances of the phi-augmented vectors, ascending (faiss_indexers.py:137-150)
    phi = float((x.astype(np.float64) ** 2).sum(1).max())
    for (gi, gd), qq in list(zip(got, q))[:50]:
        assert (np.diff(gd) >= -1e-3).all()
        ip = x[gi[0]].astype(np.float64) @ qq.astype(np.float64)
        assert abs(gd[0] - ((qq.astype(np.float64) ** 2).sum() + phi - 2 * ip)) < 1e-2 * max(1.0, abs(gd[0]))
    with pytest.raises(RuntimeError):
        approx.index_tensor([1], torch.zeros(1, 128))                           # all data at once (:111-113)


def test_ivf_persistence_roundtrip(tmp_path):
    import torch
    from lightningdot_amd.ivf import DenseIVFFlatIndexer
    x, rng = _clustered(8000, 64, 30, 3)
    q = x[:40] + 0.1
    a = DenseIVFFlatIndexer(64, nlist=32, nprobe=6)
    a.index_tensor([f'i{i}' for i in range(8000)], torch.from_numpy(x))
    ra = a.search_knn(q, 10)
    a.serialize(str(tmp_path / 'ix'))
    b = DenseIVFFlatIndexer(64)
    b.deserialize_from(str(tmp_path / 'ix'))
    rb = b.search_knn(q, 10)
    assert [r[0] for r in ra] == [r[0] for r in rb]
    np.testing.assert_array_equal(np.stack([r[1] for r in ra]), np.stack([r[1] for r in rb]))


@pytest.mark.parametrize('nq', [33, 1, 8, 16, 17])
def test_search_lists_entry_point_vs_brute_force(nq):
    """ldot_index_search_lists on hand-made lists (uneven, one empty, a -1 probe): equals a numpy scan of exactly those rows.
    33 / 17 queries: scan + threshold + collect + final kernels; 1 query: exact fp32 scan + ONE finish kernel; 8 / 16 queries: bf16
    list scan + finish kernel with the exact re-score."""
    import torch
    from lightningdot_amd import _lib as L
    from lightningdot_amd.indexer import FlatIPIndex
    rng = np.random.default_rng(4)
    x = rng.standard_normal((5000, 200)).astype(np.float32)
    q = rng.standard_normal((33, 200)).astype(np.float32)[:nq]
    offs = np.array([0, 700, 700, 2100, 2164, 5000], dtype=np.int64)            # 5 lists: 700, 0, 1400, 64, 2836 rows
    probes = np.stack([rng.permutation(5)[:3] for _ in range(33)]).astype(np.int32)[:nq]
    probes[min(5, nq - 1), 1] = -1
    ix = FlatIPIndex(200)
    ix.add(x)
    k = 20
    s = torch.empty((nq, k), dtype=torch.float32, device='cuda')
    l = torch.empty((nq, k), dtype=torch.int64, device='cuda')
    qd, od, pd = torch.from_numpy(q).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(probes).cuda()
    L.check(ix._lib.ldot_index_search_lists(ix._h, ctypes.c_void_p(qd.data_ptr()), nq, L.F32, 0, ctypes.c_void_p(od.data_ptr()), 5,
                                            2836, ctypes.c_void_p(pd.data_ptr()), 3, k, ctypes.c_void_p(s.data_ptr()),
                                            ctypes.c_void_p(l.data_ptr()), L.DEVICE,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    s, l = s.cpu().numpy(), l.cpu().numpy()
    for i in range(nq):
        rows = np.concatenate([np.arange(offs[p], offs[p + 1]) for p in probes[i] if p >= 0]) if (probes[i] >= 0).any() else np.array([], int)
        sc = x[rows].astype(np.float64) @ q[i].astype(np.float64)
        order = np.lexsort((rows, -sc))[:k]
        nv = len(order)
        np.testing.assert_allclose(s[i, :nv], sc[order], rtol=0, atol=2e-4)
        assert set(l[i, :nv]) == set(rows[order]) or np.abs(np.sort(sc[order]) - np.sort(s[i, :nv].astype(np.float64))).max() < 2e-4
        assert l[i, 0] == rows[order[0]]
        assert (l[i, nv:] == -1).all()


def test_one_call_ivf_search_equals_coarse_search_plus_list_scan():
    """ldot_ivf_search = ordinary search over the augmented centroids ([q, 0, 1]) + ldot_index_search_lists with those probes"""
    import torch
    from lightningdot_amd.ivf import DenseIVFFlatIndexer
    x, rng = _clustered(30000, 64, 120, 7)
    ivf = DenseIVFFlatIndexer(64, nlist=150, nprobe=9)
    ivf.index_tensor(list(range(30000)), torch.from_numpy(x))
    for nq in (1, 16, 300):                                   # narrow coarse search, its widest batch, more than one 256-query chunk
        q = torch.from_numpy((x[rng.integers(0, 30000, nq)] + 0.2 * rng.standard_normal((nq, 64))).astype(np.float32)).cuda()
        s1, l1 = ivf.search_knn_tensors(q, 20, exact_when_cheaper=False)
        qa = torch.cat([q, torch.zeros(nq, 1, device='cuda'), torch.ones(nq, 1, device='cuda')], 1)
        _, probes = ivf.coarse.search_tensors(qa, 9)
        s2, l2 = ivf.search_lists_tensors(q, probes, 20)
        assert torch.equal(l1, l2) and torch.equal(s1, s2)
        # against a brute-force scan of the probed rows
        offs = ivf.list_offsets.cpu().numpy()
        xs = ivf.index.get_rows(0, 30000) if hasattr(ivf.index, 'get_rows') else None
        if xs is not None:
            xs = np.asarray(xs)
            for i in range(min(nq, 5)):
                rows = np.concatenate([np.arange(offs[p], offs[p + 1]) for p in probes[i].cpu().numpy()])
                sc = xs[rows].astype(np.float64) @ q[i].cpu().numpy().astype(np.float64)
                order = np.lexsort((rows, -sc))[:20]
                np.testing.assert_allclose(s1[i].cpu().numpy(), sc[order], rtol=0, atol=2e-4)
                assert int(l1[i, 0]) == rows[order[0]]


def test_list_scan_large_k_and_full_candidate_buffer_take_the_padded_path():
    """k' * run > candidate capacity (k = 1200) never enters the run-maxima selection; 20000 identical rows in the probed lists fill the
    candidate buffer, which is noticed (stats) and redone — both against numpy"""
    import torch
    from lightningdot_amd import _lib as L
    from lightningdot_amd.indexer import FlatIPIndex
    rng = np.random.default_rng(9)
    x = rng.standard_normal((24000, 32)).astype(np.float32)
    x[1000:21000] = x[1000]                                   # 20000 equal rows inside list 1
    offs = np.array([0, 500, 22500, 24000], dtype=np.int64)
    q = np.stack([x[1000] * 2.0, rng.standard_normal(32).astype(np.float32)])
    probes = np.array([[1, 0], [2, 1]], dtype=np.int32)
    ix = FlatIPIndex(32)
    ix.add(x)
    qd, od, pd = torch.from_numpy(q).cuda(), torch.from_numpy(offs).cuda(), torch.from_numpy(probes).cuda()
    for k, expect_over in ((50, 1), (1200, 0)):
        s = torch.empty((2, k), dtype=torch.float32, device='cuda')
        l = torch.empty((2, k), dtype=torch.int64, device='cuda')
        L.check(ix._lib.ldot_index_search_lists(ix._h, ctypes.c_void_p(qd.data_ptr()), 2, L.F32, 0, ctypes.c_void_p(od.data_ptr()), 3,
                                                22000, ctypes.c_void_p(pd.data_ptr()), 2, k, ctypes.c_void_p(s.data_ptr()),
                                                ctypes.c_void_p(l.data_ptr()), L.DEVICE,
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        assert ix.last_stats()['overflowed_queries'] == expect_over
        s, l = s.cpu().numpy(), l.cpu().numpy()
        for i in range(2):
            rows = np.concatenate([np.arange(offs[p], offs[p + 1]) for p in probes[i]])
            sc = x[rows].astype(np.float64) @ q[i].astype(np.float64)
            best = np.sort(sc)[::-1][:k]
            np.testing.assert_allclose(s[i], best, rtol=0, atol=2e-4)
            np.testing.assert_allclose(x[l[i]].astype(np.float64) @ q[i].astype(np.float64), s[i], rtol=0, atol=2e-4)
            assert len(set(l[i].tolist())) == k and set(l[i].tolist()) <= set(rows.tolist())


def test_ivf_search_rejects_a_mismatched_coarse_index():
    import torch
    from lightningdot_amd import _lib as L
    from lightningdot_amd.indexer import FlatIPIndex
    ix, bad = FlatIPIndex(16), FlatIPIndex(16)
    ix.add(np.eye(16, dtype=np.float32))
    bad.add(np.eye(16, dtype=np.float32))
    q = torch.zeros(1, 16, device='cuda')
    offs = torch.tensor([0, 16], dtype=torch.int64, device='cuda')
    s = torch.empty((1, 4), dtype=torch.float32, device='cuda')
    l = torch.empty((1, 4), dtype=torch.int64, device='cuda')
    rc = ix._lib.ldot_ivf_search(ix._h, bad._h, ctypes.c_void_p(q.data_ptr()), 1, L.F32, 0, ctypes.c_void_p(offs.data_ptr()), 16, 1, 4,
                                 ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(l.data_ptr()), L.DEVICE, None)
    assert rc != 0 and b'coarse' in ix._lib.ldot_last_error()


def test_large_batches_are_answered_by_the_exact_search_when_that_is_cheaper():
    import torch
    from lightningdot_amd.ivf import DenseIVFFlatIndexer
    x, rng = _clustered(200000, 64, 400, 13)
    ivf = DenseIVFFlatIndexer(64, nlist=400, nprobe=64)
    ivf.index_tensor(list(range(200000)), torch.from_numpy(x))
    qb = torch.from_numpy((x[rng.integers(0, 200000, 2000)] + 0.1 * rng.standard_normal((2000, 64))).astype(np.float32)).cuda()
    sb, lb = ivf.search_knn_tensors(qb, 10)                       # 2000 queries x 64 of 400 lists: scanning costs more than the exact search
    assert ivf.last_route == 'exact'
    se, le = ivf.index.search_tensors(qb, 10)
    assert torch.equal(lb, le) and torch.equal(sb, se)
    sl, ll = ivf.search_knn_tensors(qb, 10, exact_when_cheaper=False)
    assert ivf.last_route == 'lists'
    assert float((ll[:, 0] == le[:, 0]).float().mean()) > 0.95


def test_ivf_search_query_dtypes_and_normalisation():
    """ldot_ivf_search ingests bf16 / fp16 queries and the opt-in L2 normalisation like the exact search does: the result equals the
    fp32 call on the rounded (and normalised) queries"""
    import torch
    from lightningdot_amd import _lib as L
    from lightningdot_amd.ivf import DenseIVFFlatIndexer
    x, rng = _clustered(20000, 64, 80, 21)
    ivf = DenseIVFFlatIndexer(64, nlist=100, nprobe=12)
    ivf.index_tensor(list(range(20000)), torch.from_numpy(x))
    q32 = torch.from_numpy((x[rng.integers(0, 20000, 40)] * 1.7 + 0.2 * rng.standard_normal((40, 64))).astype(np.float32)).cuda()
    ix = ivf.index

    def call(q, dtype, normalize):
        s = torch.empty((q.shape[0], 10), dtype=torch.float32, device='cuda')
        l = torch.empty((q.shape[0], 10), dtype=torch.int64, device='cuda')
        L.check(ix._lib.ldot_ivf_search(ix._h, ivf.coarse._h, ctypes.c_void_p(q.data_ptr()), q.shape[0], dtype, normalize,
                                        ctypes.c_void_p(ivf.list_offsets.data_ptr()), int(ivf.max_list_len), 12, 10,
                                        ctypes.c_void_p(s.data_ptr()), ctypes.c_void_p(l.data_ptr()), L.DEVICE,
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return s, l

    for dt, code in ((torch.bfloat16, L.BF16), (torch.float16, L.F16)):
        qh = q32.to(dt).contiguous()
        s_a, l_a = call(qh, code, 0)
        s_b, l_b = call(qh.float().contiguous(), L.F32, 0)
        assert torch.equal(l_a, l_b) and torch.equal(s_a, s_b)
    s_n, l_n = call(q32, L.F32, 1)
    qn = torch.nn.functional.normalize(q32.double(), dim=1).float().contiguous()
    s_m, l_m = call(qn, L.F32, 0)
    assert torch.equal(l_n, l_m)
    assert float((s_n - s_m).abs().max()) < 1e-5


def test_hnsw_surface_approximate_roundtrip(tmp_path):
    """DenseHNSWFlatIndexer(approximate=True).serialize / deserialize_from carry the inverted file (list offsets, centroids, phi,
    nprobe), not just the list-sorted rows: a loaded index answers like the one that was saved and refuses re-indexing."""
    import torch
    from lightningdot_amd.indexer import DenseHNSWFlatIndexer
    g = torch.Generator(device='cpu').manual_seed(3)
    cent = torch.randn(40, 64, generator=g) * 3
    x = (cent[torch.randint(0, 40, (6000,), generator=g)] + torch.randn(6000, 64, generator=g)).cuda()
    q = x[::300] + 0.1 * torch.randn(20, 64, generator=g).cuda()
    a = DenseHNSWFlatIndexer(64, ef_search=32, approximate=True)
    a.index_tensor([f'r{i}' for i in range(6000)], x)
    want = a.search_knn(q, 10)
    a.serialize(str(tmp_path / 'apx'))
    b = DenseHNSWFlatIndexer(64, ef_search=32, approximate=True)
    b.deserialize_from(str(tmp_path / 'apx'))
    got = b.search_knn(q, 10)
    assert [w[0] for w in want] == [g_[0] for g_ in got]
    np.testing.assert_allclose(np.stack([w[1] for w in want]), np.stack([g_[1] for g_ in got]), rtol=1e-5, atol=1e-3)
    d, lab = b.search_knn_tensors(q, 10)
    assert [[b.index_id_to_db_id[i] for i in row] for row in lab.cpu().tolist()] == [g_[0] for g_ in got]
    with pytest.raises(RuntimeError):
        b.index_tensor(['x'], x[:1])


def test_long_lists_are_split_and_every_list_probed_is_still_exact(tmp_path):
    """Automatic nlist: lists longer than 4 x the mean are split in two (2-means on their own rows) until none is left; the index stays
    a partition of the rows (probing every list = the exact search), survives a save / load, and an explicit nlist is kept as given."""
    import torch
    from lightningdot_amd.indexer import DenseFlatIndexer
    from lightningdot_amd.ivf import DenseIVFFlatIndexer
    rng = np.random.default_rng(21)
    # one dominant cluster (half of the rows) + 60 small ones: k-means leaves a few very long lists
    big = 0.3 * rng.standard_normal((30000, 64)).astype(np.float32)
    small, _ = _clustered(30000, 64, 60, 22, spread=0.2)
    x = np.concatenate([big, 3.0 + small]).astype(np.float32)
    ids = list(range(len(x)))
    plain = DenseIVFFlatIndexer(64, nprobe=8, max_list_rows=0)
    plain.index_tensor(ids, torch.from_numpy(x))
    ivf = DenseIVFFlatIndexer(64, nprobe=8)
    ivf.index_tensor(ids, torch.from_numpy(x))
    cap = max(64, 4 * -(-len(x) // plain.nlist))
    assert plain.max_list_len > cap                                   # (the data does produce long lists)
    assert ivf.max_list_len <= cap and ivf.nlist > plain.nlist
    assert int(ivf.list_offsets[-1]) == len(x) and ivf.coarse.ntotal == ivf.nlist
    assert sorted(ivf.index_id_to_db_id) == ids
    q = (x[rng.integers(0, len(x), 200)] + 0.1 * rng.standard_normal((200, 64))).astype(np.float32)
    flat = DenseFlatIndexer(64)
    flat.index_tensor(ids, torch.from_numpy(x).cuda())
    want = flat.search_knn(q, 10)
    ivf.nprobe = ivf.nlist
    for (gi, gs), (wi, ws) in zip(ivf.search_knn(q, 10), want):
        assert gi[0] == wi[0]
        np.testing.assert_allclose(gs, ws, rtol=0, atol=2e-4)
    ivf.nprobe = 8
    got = ivf.search_knn(q, 10)
    recall = np.mean([len(set(g[0]) & set(w[0])) / 10.0 for g, w in zip(got, want)])
    assert recall > 0.8, recall
    ivf.serialize(str(tmp_path / 'ivf'))
    back = DenseIVFFlatIndexer(64)
    back.deserialize_from(str(tmp_path / 'ivf'))
    assert back.nlist == ivf.nlist and back.max_list_len == ivf.max_list_len
    for (gi, gs), (bi, bs) in zip(got, back.search_knn(q, 10)):
        assert gi == bi and np.array_equal(gs, bs)
    fixed = DenseIVFFlatIndexer(64, nlist=40, nprobe=8)
    fixed.index_tensor(ids, torch.from_numpy(x))
    assert fixed.nlist == 40


def test_full_coarse_candidate_buffer_is_found_after_the_deferred_check():
    """ldot_ivf_search checks the coarse search's candidate buffer only at the synchronisation point of the list stage.  5000 IDENTICAL
    centroids: every list scores the same, the coarse search's buffer (4096 candidates) fills up, the chain notices after the fact and
    repeats the search the plain way — the probes are then the lists 0 .. nprobe-1 (ties: ascending), the result the exact top-k of their
    rows."""
    import torch
    from lightningdot_amd.ivf import DenseIVFFlatIndexer
    rng = np.random.default_rng(23)
    n, d, nlist, nprobe, k = 20000, 64, 5000, 6, 10
    x = rng.standard_normal((n, d)).astype(np.float32)
    ivf = DenseIVFFlatIndexer(d, nlist=nlist, nprobe=nprobe)
    ivf.index.add(torch.from_numpy(x).cuda())
    ivf._update_id_mapping(list(range(n)))
    ivf.list_offsets = (torch.arange(nlist + 1, dtype=torch.int64) * (n // nlist)).cuda().contiguous()
    ivf.max_list_len, ivf.biased_list_len, ivf.phi = n // nlist, float(n // nlist), 0.0
    ivf._set_coarse(torch.ones(nlist, d + 1, device='cuda') * 0.1)
    q = rng.standard_normal((3, d)).astype(np.float32)
    s, l = ivf.search_knn_tensors(torch.from_numpy(q).cuda(), k, exact_when_cheaper=False)
    rows = nprobe * (n // nlist)
    full = q.astype(np.float64) @ x[:rows].astype(np.float64).T
    want = np.argsort(-full, axis=1, kind='stable')[:, :k]
    np.testing.assert_array_equal(l.cpu().numpy(), want)
    np.testing.assert_allclose(s.cpu().numpy(), np.take_along_axis(full, want, 1), rtol=0, atol=2e-4)
    assert ivf.coarse.last_stats()['overflowed_queries'] == 3
