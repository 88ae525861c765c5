"""Throughput of the fp32-input MFMA score matrix (ldot_dot_product_scores, csrc/loss.hip) at retrieval-evaluation shapes, next to
torch.matmul (rocBLAS fp32): would exact fp32 scoring of ALL pairs beat bf16 candidates + fp32 re-score gather for tiny indexes?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.loss import dot_product_scores
torch.manual_seed(0)
for nq, n in ((5000, 1000), (5000, 5000), (25000, 5000)):
    q = torch.randn(nq, 768, device='cuda'); x = torch.randn(n, 768, device='cuda')
    for name, fn in (('ldot fp32 MFMA', lambda: dot_product_scores(q, x)), ('torch.matmul fp32', lambda: q @ x.T)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f'{nq} x {n} x 768  {name:18s} {dt * 1e6:8.1f} us  {2 * nq * n * 768 / dt / 1e12:6.1f} TFLOP/s', flush=True)
