#!/usr/bin/env python3
"""In-batch contrastive loss (a4-a7) at the config-5 shape: HIP path (lightningdot_amd.loss) vs the plain torch
formulation of the reference (matmul + log_softmax + nll_loss), forward + backward, fp32."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.loss import BiEncoderNllLoss

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

torch.manual_seed(99)
ONCE = '--once' in sys.argv     # (under rocprofv3: few repetitions, and a per-call kernel timeline of the last one)
for n1, n2, d in ((512, 512, 768), (512, 1536, 768)) if ONCE else ((512, 512, 768), (512, 1536, 768), (64, 512, 768), (4096, 4096, 768)):
    q = torch.randn(n1, d, device='cuda', requires_grad=True)
    c = torch.randn(n2, d, device='cuda', requires_grad=True)
    pos = list(range(n1))
    lf = BiEncoderNllLoss()
    def ours():
        loss, correct, scores = lf.calc(q, c, None, pos, caption_score_weight=0.0)
        loss.backward()
    def ref():
        s = q @ c.t()
        ls = F.log_softmax(s, dim=1)
        loss = F.nll_loss(ls, torch.tensor(pos, device='cuda'), reduction='mean')
        mx, idx = torch.max(ls, 1)
        correct = (idx == torch.tensor(pos, device='cuda')).sum()
        loss.backward()
    print('n1=%d n2=%d d=%d: HIP path %.1f us, torch reference formulation %.1f us (fwd+bwd)' % (n1, n2, d, timeit(ours), timeit(ref)))
