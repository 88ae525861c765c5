#!/usr/bin/env python3
"""Where do the 35-75 ms host stalls of bench.py's secondary S2 phase come from?  Replays the phase (fresh indexes, fresh pinned buffers, 3 warm-up
evaluations, 20 timed ones) several times in one process and prints the slow evaluations' positions; argv[1] = 'big' first runs a 1M-row search with
pageable host queries like the bench's pcie_inclusive leg."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex
from lightningdot_amd.synthetic import s2_embeddings
K, D = 100, 768
dev = torch.device('cuda')
if len(sys.argv) > 1 and sys.argv[1] == 'big':
    big = FlatIPIndex(D)
    big.add(torch.randn(1000000, D, device=dev))
    qh = torch.randn(10000, D).numpy()
    for _ in range(3):
        big.search(qh, K)
for rnd in range(int(os.environ.get('STALL_PROBE_ROUNDS', '6'))):
    for n_img in (1000, 5000):
        img, txt = s2_embeddings(n_img, D, 5, seed=7, device=dev)
        ix_img, ix_txt = FlatIPIndex(D), FlatIPIndex(D)
        ix_img.add(img); ix_txt.add(txt)
        hs = [torch.empty((n, K), dtype=torch.float32).pin_memory() for n in (txt.shape[0], n_img)]
        hl = [torch.empty((n, K), dtype=torch.int64).pin_memory() for n in (txt.shape[0], n_img)]
        ts = []
        for i in range(23):
            c0 = [time.clock_gettime_ns(c) for c in (time.CLOCK_MONOTONIC, time.CLOCK_BOOTTIME, time.CLOCK_REALTIME)]
            t0 = time.perf_counter()
            ix_img.search_into(txt, K, hs[0], hl[0], sync=False)
            t1 = time.perf_counter()
            ix_txt.search_into(img, K, hs[1], hl[1])
            t2 = time.perf_counter()
            ts.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3))
            if os.environ.get('STALL_PROBE_STAMPS') and (t2 - t0) > 5e-3 and i > 0:     # (tools/stall_trace.sh maps the window into the rocprofv3 trace)
                c1 = [time.clock_gettime_ns(c) for c in (time.CLOCK_MONOTONIC, time.CLOCK_BOOTTIME, time.CLOCK_REALTIME)]
                print('SLOW %d %d mono %d %d boot %d %d real %d %d' % (n_img, i, c0[0], c1[0], c0[1], c1[1], c0[2], c1[2]), flush=True)
        med = float(np.median([t[0] for t in ts]))
        slow = [(i, round(t[0], 2), round(t[1], 2)) for i, t in enumerate(ts) if t[0] > 3 * med]
        print('round %d n_img %d: median %.4f ms; slow evaluations (index, total ms, first call ms): %s' % (rnd, n_img, med, slow), flush=True)
        del ix_img, ix_txt
