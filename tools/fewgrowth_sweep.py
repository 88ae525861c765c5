"""Latency of 100- and 256-query searches over 1M x 768 (one query block of the fused scan); with the ablation build
(python -m lightningdot_amd.build --ablation) LDOT_DEBUG_FEWGROWTH overrides the launch growth of that regime."""
import os, sys, time, json, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
x = torch.randn(1_000_000, 768, device='cuda')
ix = FlatIPIndex(768); ix.add(x)
for nq in (100, 256):
    q = x[:nq] + 0.5 * torch.randn(nq, 768, device='cuda')
    hs = torch.empty((nq, 100), dtype=torch.float32).pin_memory(); hl = torch.empty((nq, 100), dtype=torch.int64).pin_memory()
    for _ in range(5): ix.search_into(q, 100, hs, hl)
    ts = []
    for _ in range(100):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ix.search_into(q, 100, hs, hl); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(os.environ.get('LDOT_DEBUG_FEWGROWTH'), nq, round(ts[50] * 1e3, 4), bool((hl[:, 0] == torch.arange(nq)).all()), flush=True)
