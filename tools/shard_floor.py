#!/usr/bin/env python3
"""Per-rank cost of the G-GPU case (default 8) on one GPU: G shards of 1M / G rows (all resident here, 4.7 GB), 10 000 queries, top-100.
The eight ranks are played one after the other; what the all-reduce(MAX) would deliver is computed from all eight shards'
statistics, so the floor every variant re-scores against is the REAL one (round 3's tool assumed a floor that keeps ~16 rows).
  (a) round 3 exchange: every shard on its own thresholds, floor = the largest k'-th best of a shard;
  (b) round 4 exchange, own thresholds: three numbers per query (ldot_index_search_begin_shard, total_rows = 0);
  (c) round 4 default: statistics pooled over the whole index (total_rows = 1M): one launch after the warm-up, no host round trip;
  (d) thresholds agreed after the warm-ups (ldot_index_search_warmup / _scan), the round-3 review's proposal.
Timed: rank 0's begin + floor + finish (the collectives themselves are not on this box).  Checked: the merge of the eight partial
lists of (c) equals the plain search of the 1M-row index bit for bit."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8           # usage: tools/shard_floor.py [ranks]  (1M rows / ranks per shard)
K, PER, D, NQ = 100, 1000000 // G, 768, 10000
g = torch.Generator(device='cuda').manual_seed(0)
q = torch.randn(NQ, D, device='cuda', generator=g)
shards = []
for r in range(G):
    ix = FlatIPIndex(D); ix.add(torch.randn(PER, D, device='cuda', generator=g)); shards.append(ix)
    if os.environ.get('SHARD_WARM'): ix.set_option(L.OPT_WARM_ROWS, int(os.environ['SHARD_WARM']))   # (A/B of the warm-up length)
ix0 = shards[0]


def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


def survivors(l):
    return float((l >= 0).sum(1).float().mean())


# (a) round-3 exchange
tau_all = torch.stack([ix.search_begin(q, K) for ix in shards], 0)
for ix in shards: ix.search_finish(None)
floor_a = tau_all.amax(0)
def run_a():
    ix0.search_begin(q, K); return ix0.search_finish(floor_a)
print('(a) own thresholds, floor = max k\'-th best:            %.3f ms' % t(run_a), ' admitted/query %.0f' % (ix0.last_stats()['fused_candidates'] / NQ),
      ' survivors/query %.1f' % survivors(run_a()[1]))

# (b) three-number exchange, own thresholds
def stats(total):
    out = []
    for ix in shards:
        out.append(ix.search_begin_shard(q, K, G, total)); ix.search_finish(None)
    return torch.stack(out, 0).amax(0)
stat_b = stats(0)
def run_b():
    ix0.search_begin_shard(q, K, G, 0)
    floor, cnt, kp = ix0.shard_floor(stat_b)
    return ix0.search_finish(floor), (cnt, kp)
print('(b) own thresholds, three-number floor:                 %.3f ms' % t(run_b), ' admitted/query %.0f' % (ix0.last_stats()['fused_candidates'] / NQ),
      ' survivors/query %.1f' % survivors(run_b()[0][1]))

# (c) pooled statistics
stat_c = stats(G * PER)
def run_c():
    ix0.search_begin_shard(q, K, G, G * PER)
    floor, cnt, kp = ix0.shard_floor(stat_c)
    return ix0.search_finish(floor), (cnt, kp)
print('(c) pooled statistics, three-number floor:              %.3f ms' % t(run_c), ' admitted/query %.0f' % (ix0.last_stats()['fused_candidates'] / NQ),
      ' survivors/query %.1f' % survivors(run_c()[0][1]))
total = 0
for ix in shards:          # the all-reduce(SUM) of the counts at or above the largest level
    ix.search_begin_shard(q, K, G, G * PER)
    _, cnt, kp = ix.shard_floor(stat_c)
    total = total + cnt
    ix.search_finish(None)
print('    queries the ranks cannot vouch for (fewer than k\' = %d rows at or above the largest level): %d;  rows up there per query: %.0f'
      % (kp, int((total < kp).sum()), float(total.float().mean())))

# (d) agreed thresholds after the warm-ups
stat_w = torch.stack([ix.search_warmup(q, K, G) for ix in shards], 0).amax(0)
for ix in shards[1:]:
    ix.search_scan(stat_w); ix.search_finish(None)
def run_d():
    ix0.search_warmup(q, K, G)
    tau = ix0.search_scan(stat_w)
    return ix0.search_finish(torch.maximum(tau, floor_a))
print('(d) thresholds agreed after the warm-ups, floor of (a): %.3f ms' % t(run_d), ' admitted/query %.0f' % (ix0.last_stats()['fused_candidates'] / NQ))

# exactness of (c): merge of the eight partial lists == plain search of the whole index
parts_s, parts_l = [], []
for r, ix in enumerate(shards):
    ix.search_begin_shard(q, K, G, G * PER)
    floor_c, _, _ = ix.shard_floor(stat_c)
    s, l = ix.search_finish(floor_c)
    parts_s.append(s); parts_l.append(torch.where(l >= 0, l + r * PER, l))
S = torch.cat(parts_s, 1); Lb = torch.cat(parts_l, 1)
S = torch.where(Lb >= 0, S, torch.full_like(S, float('-inf')))
order = torch.sort(S, dim=1, descending=True, stable=True).indices[:, :K]
ms, ml = torch.gather(S, 1, order), torch.gather(Lb, 1, order)
whole = FlatIPIndex(D)
g = torch.Generator(device='cuda').manual_seed(0)
torch.randn(NQ, D, device='cuda', generator=g)
for r in range(G):
    whole.add(torch.randn(PER, D, device='cuda', generator=g))
es, el = whole.search_tensors(q, K)
print('    merged lists of (c) == plain search of the 1M-row index: scores', bool(torch.equal(ms, es)), ' labels', bool(torch.equal(ml, el)))
