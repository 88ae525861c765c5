#!/usr/bin/env python3
"""Row chunks with every query-group width: fused scan (chunked launches, cursors carried over) against the dense path of the same index, bit for bit,
for batch sizes that make fused_query_group choose 1 / 2 / 4 / 8 query blocks per XCD and several query groups per block slot."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(3)
bad = 0
for n, d in ((400_000, 256), (1_000_000, 128), (700_001, 768)):
    x = torch.randn(n, d, device='cuda')
    fused, dense = FlatIPIndex(d), FlatIPIndex(d)
    fused.set_option(L.OPT_MODE, L.MODE_FUSED); dense.set_option(L.OPT_MODE, L.MODE_DENSE)
    fused.add(x); dense.add(x)
    for nq in (300, 513, 700, 1100, 1500, 3000, 5000):
        q = x[torch.randint(0, n, (nq,), device='cuda')] + 0.8 * torch.randn(nq, d, device='cuda')
        for k in (10, 100):
            s1, l1 = fused.search_tensors(q, k)
            t0 = time.perf_counter(); s1, l1 = fused.search_tensors(q, k); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
            s2, l2 = dense.search_tensors(q, k)
            ok = bool((l1 == l2).all()) and bool((s1 == s2).all())
            st = fused.last_stats()
            bad += not ok
            print('%s n=%d d=%d nq=%d k=%d  fused %.2f ms  overflowed %d' % ('ok  ' if ok else 'FAIL', n, d, nq, k, ms, st['overflowed_queries']), flush=True)
    del fused, dense, x
    torch.cuda.empty_cache()
print('%d mismatching cases' % bad)
