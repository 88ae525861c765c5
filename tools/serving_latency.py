#!/usr/bin/env python3
"""Single-query serving latency (a13, dvl/utils.py:204-211): one 768-d query -> top-100 over an HBM-resident index of N rows,
query vector on the device -> results on the host (pinned buffers).  Prints JSON: median / p10 / p90 latency in ms and the
fraction of the 8 TB/s HBM peak the N * D * 2 B index stream corresponds to."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex

def run(n, nq=1, d=768, k=100, reps=200, mode=None):
    torch.manual_seed(0)
    x = torch.randn(n, d, device='cuda')
    ix = FlatIPIndex(d); ix.add(x)
    if mode is not None:
        from lightningdot_amd import _lib as L
        ix.set_option(L.OPT_MODE, mode)
    q = x[:nq] + 0.5 * torch.randn(nq, d, device='cuda')
    hs = torch.empty((nq, k), dtype=torch.float32).pin_memory(); hl = torch.empty((nq, k), dtype=torch.int64).pin_memory()
    for _ in range(10): ix.search_into(q, k, hs, hl)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ix.search_into(q, k, hs, hl); ts.append(time.perf_counter() - t0)
    ts.sort()
    ok = bool((hl[:, 0] == torch.arange(nq)).all())
    med = ts[len(ts) // 2]
    return dict(rows=n, queries=nq, mode={None: 'auto', 2: 'fused'}[mode], ms_median=med * 1e3, ms_p10=ts[len(ts) // 10] * 1e3, ms_p90=ts[9 * len(ts) // 10] * 1e3,
                hbm_frac_of_8TBps=n * d * 2 / med / 8e12, rank1_ok=ok)

if __name__ == '__main__':
    only = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    for n in ([only] if only else [1_000_000, 123_287]):
        for nq in (1, 16, 64):
            print(json.dumps(run(n, nq)), flush=True)
            if os.environ.get('LDOT_COMPARE_FUSED'):
                print(json.dumps(run(n, nq, mode=2)), flush=True)
