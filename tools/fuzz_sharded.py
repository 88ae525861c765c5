#!/usr/bin/env python3
"""Randomised parity sweep of the sharded exchange (ldot_index_search_begin_shard / ldot_index_shard_floor / _finish) in ONE process: the G
ranks are played one after the other, their statistics are reduced with MAX like the all-reduce would, and the merge of the partial
lists is compared with the plain search of the whole index (labels AND fp32 scores, bit for bit).  Random shard counts and sizes
(below and above the fused threshold, an empty shard now and then), batch sizes, k, row orders that are exchangeable between the
shards or NOT (cluster-sorted rows: the pooled statistics must be caught by the check and the repeated search must be exact).
usage: tools/fuzz_sharded.py [cases] [seed]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
fails = repeated = 0
t_all = time.time()
for it in range(cases):
    G = int(rng.integers(2, 9))
    d = int(rng.choice([64, 128, 256, 768]))
    k = int(rng.choice([1, 10, 100, 100, 128, 500]))
    nq = int(rng.choice([1, 16, 64, 200, 257, 700, 1024, 2304, 5000]))
    sizes = [int(rng.choice([0, 300, 5000, 20000, 33000, 40000, 60000, 125000])) for _ in range(G)]
    if sum(sizes) == 0: sizes[0] = 5000
    if sum(sizes) * d > 3e8: d = 64
    n = sum(sizes)
    order = int(rng.integers(0, 4))   # 0, 1: i.i.d. rows; 2: clustered, shuffled; 3: clustered, stored in cluster order (NOT exchangeable)
    g = torch.Generator(device='cuda').manual_seed(seed * 100003 + it)
    x = torch.randn(n, d, device='cuda', generator=g)
    if order >= 2:
        c = torch.randn(max(1, n // 2000), d, device='cuda', generator=g)
        a = torch.randint(0, c.shape[0], (n,), device='cuda', generator=g)
        if order == 3: a = torch.sort(a).values
        x = c[a] + 0.3 * x
    q = x[torch.randint(0, n, (nq,), device='cuda', generator=g)] + 0.5 * torch.randn(nq, d, device='cuda', generator=g)
    scan = int(rng.choice([0, 0, 2]))   # LDOT_OPT_SCAN_ORDER of the shards: auto, or the scrambled tile order from the first search on
    desc = f'G={G} sizes={sizes} nq={nq} d={d} k={k} order={order} scan={scan}'
    try:
        whole = FlatIPIndex(d); whole.add(x)
        es, el = whole.search_tensors(q, k)
        shards, off = [], [0]
        for sz in sizes:
            ix = FlatIPIndex(d)
            if scan: ix.set_option(L.OPT_SCAN_ORDER, scan)
            if sz: ix.add(x[off[-1]:off[-1] + sz])
            shards.append(ix); off.append(off[-1] + sz)

        def exchange(total):
            big = [sz for sz in sizes if sz > 0 and sz * 4 * G >= n]      # ShardedFlatIndexer._share
            sh = [sz / sum(big) if sz > 0 and sz * 4 * G >= n else 0.0 for sz in sizes]
            st = torch.stack([ix.search_begin_shard(q, k, G, total, share=sh[r]) for r, ix in enumerate(shards)], 0).amax(0)
            ps, pl, cnt = [], [], 0
            for r, ix in enumerate(shards):
                floor, c, kp = ix.shard_floor(st)
                cnt = cnt + c
                s, l = ix.search_finish(floor)
                ps.append(s); pl.append(torch.where(l >= 0, l + off[r], l))
            S, Lb = torch.cat(ps, 1), torch.cat(pl, 1)
            S = torch.where(Lb >= 0, S, torch.full_like(S, float('-inf')))
            # descending score, ties by the lower label (the order of ldot_merge_topk)
            key = torch.argsort(Lb + (Lb < 0) * (1 << 60), dim=1, stable=True)
            S, Lb = torch.gather(S, 1, key), torch.gather(Lb, 1, key)
            o = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :k]
            return torch.gather(S, 1, o), torch.gather(Lb, 1, o), int((cnt < kp).sum().item())

        ms, ml, bad = exchange(n)
        rep = bad > 0
        if rep:
            repeated += 1
            ms, ml, bad2 = exchange(0)
            assert bad2 == 0, 'the exchange on own thresholds reported unproven queries'
        valid = el >= 0
        assert torch.equal(ml[:, :k][valid], el[valid]), 'labels differ'
        assert torch.equal(ms[:, :k][valid], es[valid]), 'scores differ'
        print(f'ok   {desc} unproven={bad} repeated={rep}', flush=True)
    except AssertionError as e:
        fails += 1
        print(f'FAIL {desc}: {str(e)[:300]}', flush=True)
    del whole, shards
    torch.cuda.empty_cache()
print(f'{cases - fails}/{cases} passed ({repeated} repeated on own thresholds) in {time.time() - t_all:.0f} s')
sys.exit(1 if fails else 0)
