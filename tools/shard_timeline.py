"""One rank of an 8-way sharded search on one GPU, for a kernel timeline (tools/shard_timeline.sh): 125 000-row shard, 10 000 queries.
argv[1] = rows, argv[2] = 'local' (thresholds exchanged only after the candidate pass) | 'pooled' | 'own3' | 'agreed' (after the warm-up too; the other
seven ranks' warm-up statistics come from seven small indexes, reduced with MAX like the all-reduce would)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
mode = sys.argv[2] if len(sys.argv) > 2 else 'local'
G, K = 8, 100
x = torch.randn(n, 768, device='cuda'); q = torch.randn(10000, 768, device='cuda')
ix = FlatIPIndex(768); ix.add(x)
s, _ = ix.search_tensors(q, K)
floor = s[:, 15].contiguous() - 0.05
if mode == 'agreed':
    others = []
    for r in range(G - 1):
        o = FlatIPIndex(768); o.set_option(L.OPT_MODE, L.MODE_FUSED); o.add(torch.randn(8192, 768, device='cuda')); others.append(o)
    stat_o = torch.stack([o.search_warmup(q, K, G) for o in others], 0).amax(0)
torch.cuda.synchronize()
for _ in range(6):
    if mode == 'agreed':
        stat = torch.maximum(ix.search_warmup(q, K, G), stat_o)
        tau = ix.search_scan(stat)
        ix.search_finish(torch.maximum(tau, floor))
    elif mode in ('pooled', 'own3'):
        # the round-4 default: one exchange of three numbers per query; the all-reduce is played by this shard's own statistics with
        # the floor term replaced by one that keeps ~22 rows (what eight real shards give, tools/shard_floor.py)
        stat = ix.search_begin_shard(q, K, G, G * n if mode == 'pooled' else 0)
        stat[0] = torch.maximum(stat[0], s[:, 21])
        fl, cnt, kp = ix.shard_floor(stat)
        bad = (cnt < kp).sum()     # (what follows the all-reduce(SUM) of the counts)
        ix.search_finish(fl)
    else:
        ix.search_begin(q, K); ix.search_finish(floor)
torch.cuda.synchronize()
