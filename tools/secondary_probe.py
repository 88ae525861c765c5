#!/usr/bin/env python3
"""why are the S2 shapes slow inside the headline bench process?  flickr-shape step timed before / after a 1M-row index has been built and searched"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
from lightningdot_amd.synthetic import s2_embeddings
dev = torch.device('cuda', 0); D, K = 768, 100
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def s2(tag, n_img=1000):
    img, txt = s2_embeddings(n_img, D, 5, seed=7, device=dev)
    a, b = FlatIPIndex(D), FlatIPIndex(D); a.add(img); b.add(txt)
    hs = [torch.empty((n, K), dtype=torch.float32).pin_memory() for n in (txt.shape[0], n_img)]
    hl = [torch.empty((n, K), dtype=torch.int64).pin_memory() for n in (txt.shape[0], n_img)]
    x = t(lambda: a.search_into(txt, K, hs[0], hl[0])); y = t(lambda: b.search_into(img, K, hs[1], hl[1]))
    print(tag, 'n_img', n_img, 't2i %.3f ms  i2t %.3f ms' % (x, y), a.last_regime()['path'], b.last_regime()['path'], flush=True)
s2('fresh process')
s2('fresh process', 5000)
big = FlatIPIndex(D)
prof = int(os.environ.get('PROBE_PROFILE', '1'))
big.set_option(L.OPT_PROFILE, prof)
g = torch.Generator(device='cuda').manual_seed(1)
for i in range(8): big.add(torch.randn(125000, D, device=dev, generator=g))
q = torch.randn(10000, D, device=dev, generator=g)
hs = torch.empty((10000, K), dtype=torch.float32).pin_memory(); hl = torch.empty((10000, K), dtype=torch.int64).pin_memory()
print('headline search %.3f ms (profile %d)' % (t(lambda: big.search_into(q, K, hs, hl), 5), prof), flush=True)
s2('after the 1M search')
s2('after the 1M search', 5000)
del big; torch.cuda.empty_cache()
s2('big index destroyed')
