"""Host-side logic of the inverted-file index (no GPU): the routing cost model."""


def test_ivf_route_cost_model_prefers_lists_only_for_few_queries_on_large_indexes():
    """DenseIVFFlatIndexer._exact_is_cheaper (pure host arithmetic): scanning probed lists pays off for a few queries against a large
    index; batches and small indexes are answered by the exact search"""
    from types import SimpleNamespace
    from lightningdot_amd.ivf import DenseIVFFlatIndexer
    big = SimpleNamespace(index=SimpleNamespace(ntotal=1_000_000), d=768, nlist=4000, biased_list_len=700.0)
    cheaper = lambda o, nq, nprobe: DenseIVFFlatIndexer._exact_is_cheaper(o, nq, nprobe)
    assert not cheaper(big, 1, 32) and not cheaper(big, 4, 8)
    assert not cheaper(big, 16, 32)                                # 8-16 queries scan the bf16 shadow of their lists (measured 0.19 vs 0.36 ms)
    assert cheaper(big, 16, 128) and cheaper(big, 512, 8) and cheaper(big, 10_000, 4)
    small = SimpleNamespace(index=SimpleNamespace(ntotal=50_000), d=128, nlist=900, biased_list_len=80.0)
    assert cheaper(small, 1, 4)                                    # fixed costs dominate: one pass over 12.8 MB is faster than any list scan
    huge = SimpleNamespace(index=SimpleNamespace(ntotal=50_000_000), d=768, nlist=28_000, biased_list_len=2000.0)
    assert not cheaper(huge, 64, 32)


def test_split_long_lists_bounds_the_list_length_and_keeps_a_partition():
    """_split_long_lists (pure torch, runs on CPU tensors too): every list longer than the cap is cut in two by a 2-means on its own rows
    until none is left; rows never move between unrelated lists, new centroids are the means of their members, identical rows are cut
    in the middle."""
    import torch
    from lightningdot_amd.ivf import _assign_l2, _split_long_lists
    g = torch.Generator().manual_seed(0)
    x = torch.cat([0.2 * torch.randn(3000, 16, generator=g),            # one dominant cluster
                   4.0 + torch.randn(1000, 16, generator=g),
                   torch.full((300, 16), -3.0)])                        # 300 IDENTICAL rows
    cent = torch.stack([x[:3000].mean(0), x[3000:4000].mean(0), x[4000:].mean(0)])
    assign = _assign_l2(x, cent)
    before = assign.clone()
    cent2, assign2 = _split_long_lists(x, cent.clone(), assign.clone(), cap=256, seed=0)
    counts = torch.bincount(assign2, minlength=cent2.shape[0])
    assert int(counts.sum()) == x.shape[0] and int(counts.max()) <= 256
    assert cent2.shape[0] > 3 and int((counts > 0).sum()) == cent2.shape[0]
    # a row stays inside the family of the list it started in: children of list l only hold rows of list l
    for l in range(cent2.shape[0]):
        assert len(set(before[assign2 == l].tolist())) == 1
    # centroids are the means of their members
    for l in range(cent2.shape[0]):
        torch.testing.assert_close(cent2[l], x[assign2 == l].mean(0), rtol=1e-4, atol=1e-4)
    # nothing to do below the cap
    c3, a3 = _split_long_lists(x, cent.clone(), assign.clone(), cap=5000, seed=0)
    assert c3.shape[0] == 3 and torch.equal(a3, assign)
