#!/usr/bin/env python3
"""Randomised parity sweep of ldot_index_search against the fp64 truth (tests/util.assert_topk_matches): random
(n, nq, d, k, mode, dtype, normalise, incremental add) including the thresholds of the search orchestration (dense / fused
switch at 32768 rows, query-group widths, block / wave pool select at 256 queries, k' = 2048 cap).
usage: tools/fuzz_search.py [cases] [seed]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
from util import assert_topk_matches

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
N_CHOICES = [1, 5, 255, 256, 257, 1000, 4095, 4097, 8192, 8193, 12288, 16384, 20000, 32767, 32768, 33000, 50000, 98304, 110593, 200000,
             300000, 1100000]   # 8192 / 8193: the narrow search's candidate-buffer size; > 262144 rows: coarser run maxima
Q_CHOICES = [1, 1, 2, 3, 16, 17, 63, 64, 65, 255, 256, 257, 511, 1024, 1500, 2049, 4096, 16385, 20000]   # > 16384: the fused scan's query chunks
D_CHOICES = [8, 32, 63, 64, 100, 128, 768, 1024]
K_CHOICES = [1, 5, 10, 50, 100, 128, 500, 1000, 2048]
fails = 0
t_all = time.time()
for it in range(cases):
    n = int(rng.choice(N_CHOICES)); nq = int(rng.choice(Q_CHOICES)); d = int(rng.choice(D_CHOICES)); k = int(rng.choice(K_CHOICES))
    if n * nq * d > 3e11 or nq * n > 4e8:   # keep the fp64 reference affordable
        nq = max(1, int(4e8 // n)) if nq * n > 4e8 else nq
        if n * nq * d > 3e11: d = 64
    if n * d > 2e8: d = 64 if n > 500000 else 128            # host memory / generation time
    mode = int(rng.choice([L.MODE_AUTO, L.MODE_AUTO, L.MODE_FUSED, L.MODE_DENSE]))
    normalize = bool(rng.integers(0, 4) == 0)
    clustered = bool(rng.integers(0, 3) == 0)
    x = rng.standard_normal((n, d)).astype(np.float32)
    if clustered:   # rows around few centres: many close scores
        c = rng.standard_normal((max(1, n // 50), d)).astype(np.float32)
        x = (c[rng.integers(0, c.shape[0], n)] + 0.3 * x).astype(np.float32)
        if rng.integers(0, 2) == 0:   # ... stored in cluster order (run-correlated scores: the narrow search's hard case)
            x = x[np.argsort(x @ c[0], kind='stable')]
    q = (x[rng.integers(0, n, nq)] + 0.5 * rng.standard_normal((nq, d))).astype(np.float32)
    desc = f'n={n} nq={nq} d={d} k={k} mode={mode} norm={normalize} clustered={clustered}'
    try:
        ix = FlatIPIndex(d, normalize=normalize)
        ix.set_option(L.OPT_MODE, mode)
        if rng.integers(0, 3) == 0: ix.set_option(L.OPT_SCAN_ORDER, 2)   # scrambled tile order (large batches of the fused scan)
        parts = int(rng.integers(1, 4))
        cuts = sorted(set([0, n] + [int(v) for v in rng.integers(0, n + 1, parts - 1)]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            ix.add(x[a:b])
        assert ix.ntotal == n
        s, l = ix.search(q, k)
        if normalize:
            xn = x / np.maximum(np.linalg.norm(x.astype(np.float64), axis=1, keepdims=True), 1e-12)
            qn = q / np.maximum(np.linalg.norm(q.astype(np.float64), axis=1, keepdims=True), 1e-12)
            assert_topk_matches(qn, xn, s, l, k)
        else:
            assert_topk_matches(q, x, s, l, k)
        st = ix.last_stats()
        # the device-resident route (fp32 device queries, kernel-written device outputs: a few queries are not staged, the narrow scan
        # converts them itself) must give the same bits as the host route — also when it overflows and recovers
        sd, ld = ix.search_tensors(torch.from_numpy(q).cuda(), k)
        assert np.array_equal(ld.cpu().numpy(), l) and np.array_equal(sd.cpu().numpy(), s), 'device route differs from the host route'
        print(f'ok   {desc} stats={st}', flush=True)
    except AssertionError as e:
        fails += 1
        print(f'FAIL {desc}: {str(e)[:300]}', flush=True)
    del ix
print(f'{cases - fails}/{cases} passed in {time.time() - t_all:.0f} s')
sys.exit(1 if fails else 0)
