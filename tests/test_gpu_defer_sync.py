"""LDOT_OPT_DEFER_SYNC: a search with pinned host outputs that does not wait for its results (the first direction of a retrieval
evaluation, dvl/trainer.py:160-170) — same results as the waiting search once the stream has been synchronised, on the dense path
(Flickr / COCO sized indexes) and on the fused path (where the library still synchronises internally for its overflow summary)."""
import numpy as np
import pytest
import torch

from lightningdot_amd import _lib as L

pytestmark = pytest.mark.gpu


def _pinned(n, k):
    return (torch.empty((n, k), dtype=torch.float32).pin_memory(), torch.empty((n, k), dtype=torch.int64).pin_memory())


@pytest.mark.parametrize('rows,nq', [(1000, 5000), (5000, 1000), (40000, 512)])
def test_deferred_search_equals_waiting_search(rows, nq):
    from lightningdot_amd.indexer import FlatIPIndex
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(rows, 768, device='cuda', generator=g)
    q = x[torch.arange(nq, device='cuda') % rows] + 0.7 * torch.randn(nq, 768, device='cuda', generator=g)
    k = 100
    ix = FlatIPIndex(768)
    ix.add(x)
    ref_s, ref_l = _pinned(nq, k)
    ix.search_into(q, k, ref_s, ref_l)
    out_s, out_l = _pinned(nq, k)
    out_s.fill_(float('nan'))
    out_l.fill_(-7)
    ix.search_into(q, k, out_s, out_l, sync=False)
    torch.cuda.current_stream().synchronize()
    assert np.array_equal(out_l.numpy(), ref_l.numpy())
    assert np.array_equal(out_s.numpy(), ref_s.numpy())
    # exact fp64 check of rank 1 (planted rows)
    assert (ref_l[:, 0] == (torch.arange(nq) % rows)).float().mean() > 0.99


def test_evaluation_pair_one_wait():
    """two indexes, two directions, the second search's wait covers the first (same stream); the option falls back to 0 with the next
    waiting search on a handle"""
    from lightningdot_amd.indexer import FlatIPIndex
    from lightningdot_amd.synthetic import s2_embeddings
    img, txt = s2_embeddings(1000, 768, 5, seed=7, device=torch.device('cuda'))
    ix_img, ix_txt = FlatIPIndex(768), FlatIPIndex(768)
    ix_img.add(img)
    ix_txt.add(txt)
    k = 100
    a_s, a_l = _pinned(txt.shape[0], k)
    b_s, b_l = _pinned(img.shape[0], k)
    ra_s, ra_l = _pinned(txt.shape[0], k)
    ix_img.search_into(txt, k, ra_s, ra_l)
    for _ in range(3):
        a_l.fill_(-7)
        ix_img.search_into(txt, k, a_s, a_l, sync=False)
        ix_txt.search_into(img, k, b_s, b_l)
        assert np.array_equal(a_l.numpy(), ra_l.numpy()) and np.array_equal(a_s.numpy(), ra_s.numpy())
    assert float((a_l[:, 0] == torch.arange(txt.shape[0]) // 5).float().mean()) == 1.0
    assert float(((b_l[:, 0] // 5) == torch.arange(img.shape[0])).float().mean()) == 1.0
    ix_img.search_into(txt, k, a_s, a_l)                # waiting again
    assert ix_img._opts.get(L.OPT_DEFER_SYNC) == 0


def test_defer_sync_rejects_bad_values():
    from lightningdot_amd.indexer import FlatIPIndex
    ix = FlatIPIndex(64)
    with pytest.raises(L.LdotError):
        ix.set_option(L.OPT_DEFER_SYNC, 2)
