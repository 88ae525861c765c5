"""LDOT_OPT_ROW_SHUFFLE (rows stored in a pseudo-random order behind a label table INSIDE the library) and ldot_index_last_regime,
through the C ABI as a ctypes caller binds it (FlatIPIndex = ldot_index_add / _search; no Python-side shuffling anywhere).
The search being replaced: faiss.IndexFlatIP.add / .search, dvl/indexer/faiss_indexers.py:77,83."""
import numpy as np
import pytest

from tests.util import assert_topk_matches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def L():
    from lightningdot_amd import _lib
    _lib.require_gpu()
    return _lib


def _sorted_clusters(n=200000, d=64, nq=600, nclust=800, seed=9):
    """rows sorted by a fine clustering: runs of ~250 similar rows — a run IS a 384-row tile of the fused scan"""
    rng = np.random.default_rng(seed)
    cent = rng.standard_normal((nclust, d)).astype(np.float32)
    assign = np.sort(rng.integers(0, nclust, n))
    x = (cent[assign] + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
    q = (x[rng.integers(0, n, nq)] + 0.3 * rng.standard_normal((nq, d))).astype(np.float32)
    return x, q


def _same_results(sa, la, sb, lb):
    np.testing.assert_array_equal(sa, sb)                      # the same fp32 scores, bit for bit
    np.testing.assert_array_equal(la, lb)                      # the same labels, ties included (lower label first in both)


def test_row_shuffle_at_add_time(L, tmp_path):
    from lightningdot_amd.indexer import FlatIPIndex
    x, q = _sorted_clusters()
    k = 50
    ref = FlatIPIndex(x.shape[1])
    ref.set_option(L.OPT_MODE, L.MODE_DENSE)                    # the always-correct path on the rows as they are
    ref.add(x)
    s0, l0 = ref.search(q, k)
    assert_topk_matches(q[:64], x, s0[:64], l0[:64], k)
    assert ref.last_regime()['path'] == 'dense' and ref.last_regime()['rows'] == 'as_added'

    ix = FlatIPIndex(x.shape[1])
    ix.set_option(L.OPT_MODE, L.MODE_FUSED)
    ix.set_option(L.OPT_ROW_SHUFFLE, 1)
    ix.add(x[:120000])                                          # two adds: each is shuffled within itself
    ix.add(x[120000:])
    s1, l1 = ix.search(q, k)
    st, rg = ix.last_stats(), ix.last_regime()
    assert st['overflowed_queries'] == 0                        # a fair order from the first search on
    assert rg == dict(path='fused', thresholds='optimistic', scan_order='storage', redone_queries=0, guaranteed_searches_left=0,
                      narrow_skips_left=0, scrambled_auto=False, rows='shuffled_at_add')
    _same_results(s0, l0, s1, l1)
    # few queries (the narrow search: its kernel-written outputs are bypassed, the re-score kernel translates rows to labels)
    s2, l2 = ix.search(q[:7], k)
    assert ix.last_regime()['path'] == 'fused_one_block'        # (the mode is forced)
    _same_results(s0[:7], l0[:7], s2, l2)
    ix.set_option(L.OPT_MODE, L.MODE_AUTO)
    s2, l2 = ix.search(q[:7], k)
    assert ix.last_regime()['path'] == 'narrow'
    _same_results(s0[:7], l0[:7], s2, l2)
    # rows are addressed by label from outside: get_rows, save / load
    np.testing.assert_array_equal(ix.get_rows(119990, 20), x[119990:120010])
    ix.save(str(tmp_path / 'shuffled.idx'))
    back = FlatIPIndex.load(str(tmp_path / 'shuffled.idx'))
    np.testing.assert_array_equal(back.get_rows(0, 1000), x[:1000])
    assert back.last_regime()['rows'] == 'as_added'             # the file holds the rows in insertion order
    # lists are row ranges: the inverted-file entry points refuse a shuffled store
    import ctypes
    import torch
    qd = torch.from_numpy(q[:1]).cuda()
    off = torch.tensor([0, 100, 200], dtype=torch.int64, device='cuda')
    pr = torch.zeros((1, 1), dtype=torch.int32, device='cuda')
    os_, ol = torch.empty((1, 5), device='cuda'), torch.empty((1, 5), dtype=torch.int64, device='cuda')
    vp = ctypes.c_void_p
    rc = ix._lib.ldot_index_search_lists(ix._h, vp(qd.data_ptr()), 1, L.F32, 0, vp(off.data_ptr()), 2, 100, vp(pr.data_ptr()), 1, 5,
                                         vp(os_.data_ptr()), vp(ol.data_ptr()), L.DEVICE, None)
    assert rc == -5 and b'shuffled' in ix._lib.ldot_last_error()      # LDOT_ESTATE
    with pytest.raises(L.LdotError):
        ix.set_option(L.OPT_ROW_SHUFFLE, 2)


def test_row_shuffle_engages_itself_when_both_tile_orders_fail(L):
    """default (auto): search 1 fails the optimistic check in storage order -> scrambled tile order; search 2 fails there too (a run is a
    tile) -> the library re-orders the stored rows once; search 3 flags nothing.  Results identical every time."""
    from lightningdot_amd.indexer import FlatIPIndex
    x, q = _sorted_clusters()
    k = 50
    ref = FlatIPIndex(x.shape[1])
    ref.set_option(L.OPT_MODE, L.MODE_DENSE)
    ref.add(x)
    s0, l0 = ref.search(q, k)
    ix = FlatIPIndex(x.shape[1])
    ix.set_option(L.OPT_MODE, L.MODE_FUSED)
    ix.add(x)
    seen = []
    for _ in range(4):
        s, l = ix.search(q, k)
        _same_results(s0, l0, s, l)
        seen.append((ix.last_stats()['overflowed_queries'], ix.last_regime()))
    assert seen[0][0] > 0 and seen[0][1]['scan_order'] == 'storage' and seen[0][1]['rows'] == 'as_added'
    assert seen[0][1]['redone_queries'] == seen[0][0]
    assert seen[1][0] > 0 and seen[1][1]['scan_order'] == 'scrambled_tiles' and seen[1][1]['scrambled_auto']
    assert seen[2][1]['rows'] == 'reshuffled_by_library' and seen[2][0] == 0 and seen[2][1]['thresholds'] == 'optimistic'
    assert seen[3][0] == 0 and seen[3][1]['guaranteed_searches_left'] == 0
    # rows added afterwards are shuffled too, labels keep counting in insertion order
    extra = (2.0 * x[:5000]).astype(np.float32)
    ix.add(extra)
    ref.add(extra)
    s, l = ix.search(extra[:300].copy(), 5)
    sr, lr = ref.search(extra[:300].copy(), 5)
    _same_results(sr, lr, s, l)
    assert (l >= len(x)).any() and ix.last_regime()['rows'] == 'reshuffled_by_library'
    np.testing.assert_array_equal(ix.get_rows(len(x), 5), extra[:5])
    # LDOT_OPT_ROW_SHUFFLE = 2: never — the index ends on guaranteed thresholds as in round 4
    off = FlatIPIndex(x.shape[1])
    off.set_option(L.OPT_MODE, L.MODE_FUSED)
    off.set_option(L.OPT_ROW_SHUFFLE, 2)
    off.add(x)
    for _ in range(3):
        s, l = off.search(q, k)
        _same_results(s0, l0, s, l)
    rg = off.last_regime()
    assert rg['rows'] == 'as_added' and (rg['guaranteed_searches_left'] > 0 or rg['thresholds'] == 'guaranteed')


def test_shuffled_store_breaks_ties_by_label(L):
    """every row four times: equal scores everywhere; the reported order must be (score desc, LABEL asc) whatever the storage order"""
    from lightningdot_amd.indexer import FlatIPIndex
    rng = np.random.default_rng(4)
    base = rng.standard_normal((1500, 96)).astype(np.float32)
    x = np.concatenate([base, base, base, base], axis=0)
    q = (base[:40] + 0.1 * rng.standard_normal((40, 96))).astype(np.float32)
    for mode in (L.MODE_DENSE, L.MODE_AUTO):
        a = FlatIPIndex(96)
        a.set_option(L.OPT_MODE, mode)
        a.add(x)
        b = FlatIPIndex(96)
        b.set_option(L.OPT_MODE, mode)
        b.set_option(L.OPT_ROW_SHUFFLE, 1)
        b.add(x)
        sa, la = a.search(q, 12)
        sb, lb = b.search(q, 12)
        _same_results(sa, la, sb, lb)
        assert (la[:, 0] % 1500 == np.arange(40)).all() and (np.diff(la[:, :4], axis=1) == 1500).all()
