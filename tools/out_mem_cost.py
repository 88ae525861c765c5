"""Search time at the Flickr shapes with device outputs vs pinned host outputs (the re-score kernel writes host buffers directly):
what the transfer of the results costs (measured: 0.220 vs 0.241 ms at 5000 x 1000, 0.391 vs 0.449 ms at 5000 x 5000)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
for nq, n in ((5000, 1000), (5000, 5000)):
    x = torch.randn(n, 768, device='cuda'); q = torch.randn(nq, 768, device='cuda')
    ix = FlatIPIndex(768); ix.add(x)
    hs = torch.empty((nq, 100), dtype=torch.float32).pin_memory(); hl = torch.empty((nq, 100), dtype=torch.int64).pin_memory()
    def t(fn, reps=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    print(nq, n, 'device outputs %.3f ms' % t(lambda: ix.search_tensors(q, 100)), 'pinned host outputs %.3f ms' % t(lambda: ix.search_into(q, 100, hs, hl)), flush=True)
