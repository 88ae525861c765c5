#!/usr/bin/env python3
"""What a 10 000-query search over 1M x 768 costs when the ROW ORDER is unfriendly to the fused scan's lane-private candidate pools,
and whether the results stay exact: rows in random order (baseline), clustered rows in random order, clustered rows SORTED BY
k-means list (the storage order of the inverted-file index, whose large batches are answered by the exact scan of those rows), a few
large contiguous clusters, and the adversarial ramp (every later row beats all earlier ones).  Prints one JSON line per case:
ms per search, overflowed queries, queries redone, mismatches of a 64-query sample against a brute-force fp64 scan."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import DenseFlatIndexer   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
D, K = 768, 100
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)


def check(x, q, labels, scores, sample=64):
    idx = torch.linspace(0, q.shape[0] - 1, sample, device=dev).long()
    full = q[idx].double() @ x.double().T
    top = full.topk(K, dim=1)
    got_true = torch.gather(full, 1, labels[idx])
    bad_rank1 = int((labels[idx, 0] != top.indices[:, 0]).sum())
    # a reported row must score at least the true k-th best (ties may swap); reported scores must be the true inner products
    not_topk = int((got_true < top.values[:, -1:] - 1e-6 * full.abs().max()).any(dim=1).sum())
    dscore = float((got_true - scores[idx].double()).abs().max())
    return dict(rank1_mismatch=bad_rank1, rows_outside_topk=not_topk, max_abs_dscore=dscore)


def run(name, x, q):
    ix = DenseFlatIndexer(D)
    ix.index.add(x)
    for _ in range(2):
        s, l = ix.index.search_tensors(q, K)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        s, l = ix.index.search_tensors(q, K)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    st = ix.index.last_stats()
    extra = {}
    if st['overflowed_queries']:          # the recovered results equal the always-correct dense path's, bit for bit (512-query sample)
        from lightningdot_amd import _lib as L
        sub = torch.linspace(0, q.shape[0] - 1, 512, device=dev).long()
        ix.index.set_option(L.OPT_MODE, L.MODE_DENSE)
        sd, ld = ix.index.search_tensors(q[sub], K)
        ix.index.set_option(L.OPT_MODE, L.MODE_AUTO)
        extra = dict(labels_equal_dense=bool((ld == l[sub]).all()), scores_equal_dense=bool((sd == s[sub]).all()))
    out = dict(case=name, rows=x.shape[0], queries=q.shape[0], ms=sorted(ts)[len(ts) // 2] * 1e3, **st, **extra, **check(x, q, l, s))
    print(json.dumps(out), flush=True)
    del ix
    torch.cuda.empty_cache()


# (a) i.i.d. rows, planted queries (the bench workload's statistics)
x = torch.randn(N, D, device=dev, generator=g)
q = x[(torch.arange(Q, device=dev) * 9973) % N] + 0.5 * torch.randn(Q, D, device=dev, generator=g)
run('iid_random_order', x, q)
# (b) mixture of 4000 Gaussians, random order
cent = torch.randn(4000, D, device=dev, generator=g)
assign = torch.randint(0, 4000, (N,), device=dev, generator=g)
x = cent[assign] + 0.5 * torch.randn(N, D, device=dev, generator=g)
q = x[torch.randint(0, N, (Q,), device=dev, generator=g)] + 0.3 * torch.randn(Q, D, device=dev, generator=g)
run('clustered_random_order', x, q)
# (c) the same rows sorted by cluster (what DenseIVFFlatIndexer stores)
order = torch.argsort(assign, stable=True)
run('clustered_sorted_by_list_4000', x[order].contiguous(), q)
# (d) 40 large contiguous clusters (25 000 rows each)
cent = torch.randn(40, D, device=dev, generator=g)
assign = torch.arange(N, device=dev) // ((N + 39) // 40)
x = cent[assign] + 0.5 * torch.randn(N, D, device=dev, generator=g)
q = x[torch.randint(0, N, (Q,), device=dev, generator=g)] + 0.3 * torch.randn(Q, D, device=dev, generator=g)
run('clustered_sorted_40_large', x, q)
# (e) adversarial ramp: every later row beats all earlier ones for every query
base = torch.randn(D, device=dev, generator=g)
base = base / base.norm()
x = 0.01 * torch.randn(N, D, device=dev, generator=g) + torch.linspace(0.0, 300.0, N, device=dev)[:, None] * base[None, :]
q = base[None, :] + 0.01 * torch.randn(Q, D, device=dev, generator=g)
run('adversarial_ramp', x, q)
