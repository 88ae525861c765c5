#!/usr/bin/env python3
"""How deep in the bf16-ordered candidate list does the exact top-100 reach?  Headline workload (1M x 768 i.i.d. rows, 10 000 planted
queries): for every query the largest bf16 rank of a row of its exact (fp32 re-scored) top-100, and the number of candidates within
2E of the 100th bf16 score, E = 4 * 2^-8 * |q| * max|x| / sqrt(d) (the statistical error bound of LDOT_OPT_VERIFY)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
N, Q, D, K = 1_000_000, 10_000, 768, 100
g = torch.Generator(device='cuda').manual_seed(1)
x = torch.randn(N, D, device='cuda', generator=g)
q = x[(torch.arange(Q, device='cuda') * 9973) % N] + 0.5 * torch.randn(Q, D, device='cuda', generator=g)
ix = FlatIPIndex(D); ix.add(x)
es, el = ix.search_tensors(q, K)                       # exact fp32 scores, k' = 128 candidates behind them
ix.set_option(L.OPT_RESCORE, 0)
ix.set_option(L.OPT_MARGIN, 412)
bs, bl = ix.search_tensors(q, 512)                     # bf16-input scores, bf16 order, deep list
pos = (bl[:, None, :] == el[:, :, None]).int().argmax(2) if False else None
deep = torch.zeros(Q, dtype=torch.int64, device='cuda')
for c in range(0, Q, 500):
    m = bl[c:c + 500, None, :] == el[c:c + 500, :, None]          # [q, 100, 512]
    assert bool(m.any(2).all())
    deep[c:c + 500] = m.int().argmax(2).max(1).values + 1
E = 4 * 2 ** -8 * q.norm(dim=1) * x.norm(dim=1).max() / D ** 0.5
within = (bs >= (bs[:, K - 1] - 2 * E)[:, None]).sum(1)
print('largest bf16 rank of a row of the exact top-100: max over queries %d, 99.9 %% quantile %d, mean %.1f'
      % (int(deep.max()), int(deep.float().quantile(0.999)), float(deep.float().mean())))
print('candidates within 2E of the 100th bf16 score (E = %.3f on average): mean %.1f, max %d' % (float(E.mean()), float(within.float().mean()), int(within.max())))
