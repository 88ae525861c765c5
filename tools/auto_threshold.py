#!/usr/bin/env python3
"""dense vs fused search time around the AUTO switch (index rows 8k..48k, various batch sizes)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
def t(ix, q, k, n=5):
    ix.search_tensors(q, k); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): ix.search_tensors(q, k)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for n in (5120, 8192, 12288, 16384, 20480, 24576, 32768, 49152):
    x = torch.randn(n, 768, device='cuda')
    for nq in (300, 1024, 5000, 25000):
        q = torch.randn(nq, 768, device='cuda')
        res = []
        for mode in (1, 2):
            ix = FlatIPIndex(768); ix.set_option(1, mode); ix.add(x)
            res.append(t(ix, q, 100)); del ix
        print('n=%6d nq=%6d  dense %.3f ms  fused %.3f ms  -> %s' % (n, nq, res[0], res[1], 'fused' if res[1] < res[0] else 'dense'))
