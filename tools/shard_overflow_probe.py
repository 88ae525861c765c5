#!/usr/bin/env python3
"""pool overflows (queries a shard has to search again) and per-shard time of the pooled-statistics shard search at 8 x 125 000 rows, by warm-up length"""
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
for warm in ([int(a) for a in sys.argv[1:]] or [0, 4096]):
    for ix in shards:
        if warm: ix.set_option(L.OPT_WARM_ROWS, warm)
    out = []
    for ix in shards:
        out.append(ix.search_begin_shard(q, K, G, G * PER)); ix.search_finish(None)
    stat = torch.stack(out, 0).amax(0)
    for trial in range(2):
        row = []
        for ix in shards:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                ix.search_begin_shard(q, K, G, G * PER); floor, cnt, kp = ix.shard_floor(stat); ix.search_finish(floor)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
            st = ix.last_stats()
            row.append('%.2f ms/%d over/%d rec' % (ms, st['overflowed_queries'], st['fused_candidates'] // NQ))
        print('warm %s:' % (warm or 'default'), ' | '.join(row), flush=True)
