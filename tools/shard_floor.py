#!/usr/bin/env python3
"""Per-rank cost of the 8-GPU case on one GPU: 125 000-row shard, 10 000 queries, top-100; re-score with and without
the global-threshold floor (emulated: a floor that keeps ~k'/8 candidates per query, what the all-reduce(MAX) gives on average)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
x = torch.randn(125000, 768, device='cuda'); q = torch.randn(10000, 768, device='cuda')
ix = FlatIPIndex(768); ix.add(x)
s, l = ix.search_tensors(q, 100)
floor = s[:, 15].contiguous() - 0.05          # ~16 candidates survive
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def both(fl):
    ix.search_begin(q, 100); return ix.search_finish(fl)
print('begin+finish, no floor: %.3f ms' % t(lambda: both(None)))
print('begin+finish, floor keeping ~16: %.3f ms' % t(lambda: both(floor)))
s2, l2 = both(floor)
print('survivors per query: %.1f' % float((l2 >= 0).sum(1).float().mean()))
