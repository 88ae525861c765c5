#!/usr/bin/env python3
"""Approximate (inverted-file) index vs the exact flat index on a clustered 1M x 768 set: build time, recall@10 / rank-1 agreement
with the EXACT top-10 (not with a planted row), single-query / 16-query latency per nprobe.  Prints JSON lines.

    python tools/ivf_bench.py [rows] [centroid_spread]
    python tools/ivf_bench.py [rows] curve            # recall@10 vs nprobe only, for several spreads (no latency loops)
    IVF_MAX_LIST_ROWS=0 python tools/ivf_bench.py ... # lists as k-means leaves them (no splitting of the long ones): the A/B of that step

Data: a mixture of 4000 Gaussians whose centroids are spread by `centroid_spread` (default 0.2) per dimension around the origin, rows =
centroid + 0.5 N(0,1) — OVERLAPPING clusters: the inner-product advantage of a query's own cluster (768 spread^2 ~ 11) is comparable to
the noise of the cross terms (~7), so the exact top-10 of a query is spread over several lists and recall really depends on nprobe.
(Well separated clusters, spread 1.0, give recall 1.0 at every nprobe — uninformative.)  Queries are fresh draws from the mixture."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import DenseFlatIndexer
from lightningdot_amd.ivf import DenseIVFFlatIndexer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
CAP = int(os.environ['IVF_MAX_LIST_ROWS']) if 'IVF_MAX_LIST_ROWS' in os.environ else None


def rows_scanned(ivf, q, nprobe):
    """mean number of rows in the nprobe lists a query probes"""
    qa = torch.cat([q, torch.zeros(len(q), 1, device='cuda'), torch.ones(len(q), 1, device='cuda')], 1)
    _, pr = ivf.coarse.search_tensors(qa, min(nprobe, ivf.nlist))
    lens = ivf.list_offsets[1:] - ivf.list_offsets[:-1]
    return float(lens[pr.clamp_min(0)].sum(1).float().mean())

if len(sys.argv) > 2 and sys.argv[2] == 'curve':
    for spread in (0.16, 0.2, 0.25, 0.35):
        g = torch.Generator(device='cuda').manual_seed(0)
        cent = spread * torch.randn(4000, 768, device='cuda', generator=g)
        x = cent[torch.randint(0, 4000, (N,), device='cuda', generator=g)] + 0.5 * torch.randn(N, 768, device='cuda', generator=g)
        q = cent[torch.randint(0, 4000, (512,), device='cuda', generator=g)] + 0.5 * torch.randn(512, 768, device='cuda', generator=g)
        flat = DenseFlatIndexer(768); flat.index_tensor(list(range(N)), x)
        _, el = flat.search_knn_tensors(q, 10)
        ivf = DenseIVFFlatIndexer(768, nprobe=32, max_list_rows=CAP); ivf.index_tensor(list(range(N)), x)
        inv = torch.as_tensor(ivf.index_id_to_db_id, device='cuda')
        row = dict(rows=N, centroid_spread=spread, nlist=ivf.nlist, longest_list=ivf.max_list_len, recall_at_10={}, rank1_agreement={},
                   rows_scanned={})
        for nprobe in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
            _, l = ivf.search_knn_tensors(q, 10, nprobe, exact_when_cheaper=False)
            orig = torch.where(l >= 0, inv[l.clamp_min(0)], l)
            row['recall_at_10'][nprobe] = round(float(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(orig, el)) / 5120), 4)
            row['rank1_agreement'][nprobe] = round(float((orig[:, 0] == el[:, 0]).float().mean()), 4)
            row['rows_scanned'][nprobe] = round(rows_scanned(ivf, q, nprobe))
        print(json.dumps(row), flush=True)
        del flat, ivf, x
        torch.cuda.empty_cache()
    sys.exit(0)
SPREAD = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
D, K = 768, 10
g = torch.Generator(device='cuda').manual_seed(0)
cent = SPREAD * torch.randn(4000, D, device='cuda', generator=g)
x = cent[torch.randint(0, 4000, (N,), device='cuda', generator=g)] + 0.5 * torch.randn(N, D, device='cuda', generator=g)
q = cent[torch.randint(0, 4000, (512,), device='cuda', generator=g)] + 0.5 * torch.randn(512, D, device='cuda', generator=g)
ids = list(range(N))
flat = DenseFlatIndexer(D); flat.index_tensor(ids, x)
es, el = flat.search_knn_tensors(q, K)
t0 = time.perf_counter()
ivf = DenseIVFFlatIndexer(D, nprobe=32, max_list_rows=CAP); ivf.index_tensor(ids, x); torch.cuda.synchronize()
build = time.perf_counter() - t0
inv = torch.as_tensor(ivf.index_id_to_db_id, device='cuda')          # sorted row -> original row

def lat(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e3

print(json.dumps(dict(rows=N, centroid_spread=SPREAD, nlist=ivf.nlist, longest_list=ivf.max_list_len, build_s=build,
                      exact_ms_1q=lat(lambda: flat.search_knn_tensors(q[:1], K)), exact_ms_16q=lat(lambda: flat.search_knn_tensors(q[:16], K)))), flush=True)
for nprobe in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    s, l = ivf.search_knn_tensors(q, K, nprobe, exact_when_cheaper=False)
    orig = torch.where(l >= 0, inv[l.clamp_min(0)], l)
    recall = float(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(orig, el)) / (512 * K))
    r1 = float((orig[:, 0] == el[:, 0]).float().mean())
    print(json.dumps(dict(nprobe=nprobe, recall_at_10=recall, rank1_agreement=r1, rows_scanned=round(rows_scanned(ivf, q, nprobe)),
                          ms_1q=lat(lambda: ivf.search_knn_tensors(q[:1], K, nprobe, exact_when_cheaper=False)),
                          ms_16q=lat(lambda: ivf.search_knn_tensors(q[:16], K, nprobe, exact_when_cheaper=False)),
                          ms_512q=lat(lambda: ivf.search_knn_tensors(q, K, nprobe, exact_when_cheaper=False), 20),
                          # default routing: the exact search answers when the cost model says it is faster
                          routed={n: (round(lat(lambda: ivf.search_knn_tensors(q[:n], K, nprobe), 20), 4), ivf.last_route)
                                  for n in (1, 4, 16, 512)})), flush=True)
