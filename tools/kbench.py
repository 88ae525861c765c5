#!/usr/bin/env python3
"""Quick kernel bench: fused search on a GPU-generated synthetic problem (no host data generation).
usage: python tools/kbench.py [rows] [queries] [reps]   -> prints score-kernel ms and algorithmic TFLOP/s"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.manual_seed(0)
x = torch.randn(rows, 768, device='cuda')
q = x[(torch.arange(nq, device='cuda') * 9973) % rows] + 0.5 * torch.randn(nq, 768, device='cuda')
ix = FlatIPIndex(768)
ix.set_option(L.OPT_MODE, L.MODE_FUSED)
ix.set_option(L.OPT_PROFILE, 1)
ix.add(x)
del x
s, l = ix.search_tensors(q, 100)
gt = (torch.arange(nq, device='cuda') * 9973) % rows
ok = bool((l[:, 0] == gt).all())
best = 1e9
tot = 0.0
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s, l = ix.search_tensors(q, 100)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    p = ix.last_profile()
    best = min(best, p['kernel_ms'])
    tot = dt
print('variant=%s rows=%d nq=%d  kernel_ms(best)=%.3f  TF=%.0f  step_ms(last)=%.2f  rank1_ok=%s ovf=%d' % (
    os.environ.get('LDOT_DEBUG_VARIANT', '0'), rows, nq, best, p['flops'] / best / 1e9, tot, ok,
    ix.last_stats()['overflowed_queries']))
