#!/usr/bin/env python3
"""ONE mining search (dvl/hn.py:45-66 at the Flickr30k train set's size) repeated a few times, for rocprofv3 kernel stats:
    python tools/mining_one.py t2i|i2t <k> exact|ids [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightningdot_amd.indexer import FlatIPIndex
from lightningdot_amd.synthetic import s2_embeddings
direction, k, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
search_mode = int(os.environ.get("MINING_SEARCH_MODE", "0"))      # LDOT_OPT_MODE: 0 auto, 1 dense, 2 fused
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
img, txt = s2_embeddings(29000, 768, 5, seed=11, device='cuda')
x, q = (img, txt) if direction == 't2i' else (txt, img)
ix = FlatIPIndex(768)
ix.add(x)
if search_mode:
    ix.set_option(1, search_mode)
for _ in range(reps):
    s, l = ix.search_tensors(q, k, ids_only=(mode == 'ids'))
torch.cuda.synchronize()
print(direction, k, mode, ix.last_regime(), ix.last_stats(), ix.last_set_stats() if mode == 'ids' else '')
