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
