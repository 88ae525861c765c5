import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
x = torch.randn(1_000_000, 768, device='cuda')
for mode in (0, 1):
    ix = FlatIPIndex(768); ix.set_option(1, mode); ix.add(x)
    for nq in (1, 8, 64, 256):
        q = x[:nq] + 0.5 * torch.randn(nq, 768, device='cuda')
        ix.search_tensors(q, 100); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): s, l = ix.search_tensors(q, 100)
        torch.cuda.synchronize()
        print('mode=%d nq=%d  %.3f ms  ok=%s' % (mode, nq, (time.perf_counter() - t0) / 5 * 1e3, bool((l[:, 0] == torch.arange(nq, device='cuda')).all())))
    del ix
