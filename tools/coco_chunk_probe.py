#!/usr/bin/env python3
"""COCO text->image (25 000 x 5 000, dense path): does a smaller query block per dense chunk (score chunk resident in the Infinity Cache between the
score kernel's writes and the select's reads) help?  LDOT_OPT_CHUNK_ROWS sets the rows per chunk and with it the queries per block (2^29 / chunk_rows)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
from lightningdot_amd.synthetic import s2_embeddings
dev = torch.device('cuda', 0); D, K = 768, 100
img, txt = s2_embeddings(5000, D, 5, seed=7, device=dev)
hs = torch.empty((txt.shape[0], K), dtype=torch.float32).pin_memory(); hl = torch.empty((txt.shape[0], K), dtype=torch.int64).pin_memory()
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    for chunk in (32768, 65536, 131072, 262144, 524288):
        ix = FlatIPIndex(D); ix.set_option(L.OPT_CHUNK_ROWS, chunk); ix.add(img)
        ms = t(lambda: ix.search_into(txt, K, hs, hl))
        print('chunk_rows %7d (query block %5d): t2i %.3f ms' % (chunk, max(256, (2**29 // chunk) // 256 * 256), ms), flush=True)
        del ix
