#!/usr/bin/env python3
"""per-rank time of the pooled-statistics shard search (8 x 125 000 rows, real exchanged statistics) against the shard's warm-up length"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
G, K, D, NQ = 8, 100, 768, 10000
PER = 1000000 // G
g = torch.Generator(device='cuda').manual_seed(0)
q = torch.randn(NQ, D, device='cuda', generator=g)
shards = []
for r in range(G):
    ix = FlatIPIndex(D); ix.add(torch.randn(PER, D, device='cuda', generator=g)); shards.append(ix)
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for warm in (4096, 3072, 2048):
    for ix in shards: ix.set_option(L.OPT_WARM_ROWS, warm)
    out = []
    for ix in shards:
        out.append(ix.search_begin_shard(q, K, G, G * PER)); ix.search_finish(None)
    stat = torch.stack(out, 0).amax(0)
    ix0 = shards[0]
    def run():
        ix0.search_begin_shard(q, K, G, G * PER)
        floor, cnt, kp = ix0.shard_floor(stat)
        return ix0.search_finish(floor), cnt, kp
    for rep in range(2):
        ms = t(run)
        (s, l), cnt, kp = run()
        tot = 0
        for ix in shards:
            ix.search_begin_shard(q, K, G, G * PER); _, c, _ = ix.shard_floor(stat); tot = tot + c; ix.search_finish(None)
        print('warm %d: %.3f ms per rank  admitted/query %.0f  unproven %d' % (warm, ms, ix0.last_stats()['fused_candidates'] / NQ, int((tot < kp).sum())), flush=True)
