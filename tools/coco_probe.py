#!/usr/bin/env python3
"""COCO-shape evaluation (25 000 x 5 000 + 5 000 x 25 000, top-100) with LDOT_OPT_PROFILE off / on: ms per evaluation and the regime of each search"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
from lightningdot_amd.synthetic import s2_embeddings
dev = torch.device('cuda', 0)
D, K = 768, 100
img, txt = s2_embeddings(5000, D, 5, seed=7, device=dev)
for prof in (0, 1, 0):
    ix_img, ix_txt = FlatIPIndex(D), FlatIPIndex(D)
    ix_img.set_option(L.OPT_PROFILE, prof); ix_txt.set_option(L.OPT_PROFILE, prof)
    ix_img.add(img); ix_txt.add(txt)
    hs = [torch.empty((n, K), dtype=torch.float32).pin_memory() for n in (txt.shape[0], 5000)]
    hl = [torch.empty((n, K), dtype=torch.int64).pin_memory() for n in (txt.shape[0], 5000)]
    def t(fn, n=10):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    a = t(lambda: ix_img.search_into(txt, K, hs[0], hl[0])); ra = ix_img.last_regime()
    b = t(lambda: ix_txt.search_into(img, K, hs[1], hl[1])); rb = ix_txt.last_regime()
    print('profile', prof, 't2i %.3f ms' % a, ra['path'], 'i2t %.3f ms' % b, rb['path'], flush=True)
