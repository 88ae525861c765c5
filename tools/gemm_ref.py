#!/usr/bin/env python3
"""Vendor-library reference points on this GPU (hipBLASLt through torch.matmul, bf16 in / bf16 out):
the practical MFMA ceiling on a large square GEMM and on the score-matrix shape (K = 768)."""
import torch, time
def run(m, n, k, iters=10):
    a = torch.randn(m, k, device='cuda', dtype=torch.bfloat16)
    b = torch.randn(n, k, device='cuda', dtype=torch.bfloat16)
    for _ in range(3): c = a @ b.T
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): c = a @ b.T
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print('M=%d N=%d K=%d: %.3f ms  %.0f TFLOP/s' % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9))
run(8192, 8192, 8192)
run(16384, 16384, 4096)
run(10240, 131072, 768)
run(10240, 262144, 768)
run(131072, 10240, 768)
