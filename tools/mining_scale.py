#!/usr/bin/env python3
"""Hard-negative-mining-scale searches (dvl/hn.py:45-66 on Flickr30k train: ~29k images x ~145k captions, top-50,
both directions, queries un-deduplicated) on synthetic vectors: timing + planted-rank-1 check."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
n_img, cpi, d, k = 29000, 5, 768, 50
img = torch.randn(n_img, d, device='cuda')
cap = img.repeat_interleave(cpi, 0) + 0.9 * torch.randn(n_img * cpi, d, device='cuda')
gt_img = torch.arange(n_img, device='cuda').repeat_interleave(cpi)
for name, x, q, gt in (('txt->img', img, cap, gt_img), ('img->txt (un-deduplicated image queries)', cap, img.repeat_interleave(cpi, 0), None)):
    ix = FlatIPIndex(d); ix.add(x)
    if os.environ.get('MODE'): ix.set_option(1, int(os.environ['MODE']))
    ix.search_tensors(q[:1024], k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s, l = ix.search_tensors(q, k)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = bool((l[:, 0] == gt).all()) if gt is not None else bool(((l[:, 0] // cpi) == gt_img).all())
    print('%s: Q=%d N=%d k=%d  %.1f ms  (%.0f queries/s, %.0f TFLOP/s algorithmic)  rank1_ok=%s stats=%s' % (
        name, q.shape[0], x.shape[0], k, dt * 1e3, q.shape[0] / dt, 2.0 * q.shape[0] * x.shape[0] * d / dt / 1e12, ok, ix.last_stats()))
    del ix
