#!/usr/bin/env python3
"""A/B of the inverted-file query's list scan on ONE data set: bf16 shadow + exact re-score (default) vs the exact fp32 scan
(LDOT_DEBUG_IVF_FP32=1, ablation library).  usage: LDOT_LIBRARY=.../libldot_ablation.so [LDOT_DEBUG_IVF_FP32=1] python tools/ivf_ab.py [spread]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.ivf import DenseIVFFlatIndexer
N, D, K = 1_000_000, 768, 10
SPREAD = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
g = torch.Generator(device='cuda').manual_seed(0)
cent = SPREAD * torch.randn(4000, D, device='cuda', generator=g)
x = cent[torch.randint(0, 4000, (N,), device='cuda', generator=g)] + 0.5 * torch.randn(N, D, device='cuda', generator=g)
q = cent[torch.randint(0, 4000, (64,), device='cuda', generator=g)] + 0.5 * torch.randn(64, D, device='cuda', generator=g)
ivf = DenseIVFFlatIndexer(D, nprobe=32); ivf.index_tensor(list(range(N)), x); torch.cuda.synchronize()
def lat(fn, reps=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e3
out = dict(scan='fp32' if os.environ.get('LDOT_DEBUG_IVF_FP32') else 'bf16+rescore', spread=SPREAD, longest_list=ivf.max_list_len)
for nprobe in (8, 32, 128):
    out[f'ms_1q_nprobe{nprobe}'] = round(lat(lambda: ivf.search_knn_tensors(q[:1], K, nprobe, exact_when_cheaper=False)), 4)
    out[f'ms_16q_nprobe{nprobe}'] = round(lat(lambda: ivf.search_knn_tensors(q[:16], K, nprobe, exact_when_cheaper=False), 100), 4)
print(json.dumps(out), flush=True)
