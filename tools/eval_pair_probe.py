#!/usr/bin/env python3
"""Per-evaluation wall times of the Flickr-shape evaluation pair (text->image + image->text, pinned host results), with the first search
waiting for its results (sync=True) or not (LDOT_OPT_DEFER_SYNC): median, worst, outliers.  usage: tools/eval_pair_probe.py [n_img] [reps]"""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex
from lightningdot_amd.synthetic import s2_embeddings
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
K, D = 100, 768
dev = torch.device('cuda')
img, txt = s2_embeddings(n_img, D, 5, seed=7, device=dev)
ix_img, ix_txt = FlatIPIndex(D), FlatIPIndex(D)
ix_img.add(img); ix_txt.add(txt)
hs = [torch.empty((n, K), dtype=torch.float32).pin_memory() for n in (txt.shape[0], n_img)]
hl = [torch.empty((n, K), dtype=torch.int64).pin_memory() for n in (txt.shape[0], n_img)]
for trial in range(2):
    for first_sync in (True, False):
        ts = []
        for i in range(reps + 5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ix_img.search_into(txt, K, hs[0], hl[0], sync=first_sync)
            ix_txt.search_into(img, K, hs[1], hl[1])
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = ts[5:]
        print('n_img %d first search %s: median %.4f ms  mean %.4f  worst %.3f  > 2x median: %d of %d' % (
            n_img, 'waits' if first_sync else 'deferred', statistics.median(ts), sum(ts) / len(ts), max(ts),
            sum(t > 2 * statistics.median(ts) for t in ts), len(ts)), flush=True)
