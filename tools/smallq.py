#!/usr/bin/env python3
"""Latency of small query batches against a 1M x 768 index (single-query serving shape)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
torch.manual_seed(0)
x = torch.randn(rows, 768, device='cuda')
ix = FlatIPIndex(768)
ix.add(x)
for nq in (1, 8, 64, 256, 1024):
    q = x[:nq] + 0.5 * torch.randn(nq, 768, device='cuda')
    ix.search_tensors(q, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        s, l = ix.search_tensors(q, 100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    print('nq=%d  %.3f ms/search  rank1_ok=%s' % (nq, dt, bool((l[:, 0] == torch.arange(nq, device='cuda')).all())))
