"""LDOT_DEBUG_QG sweep (ablation library): one batch whose block count every group width divides, so that only the width's own cost shows."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(1000000, 768, device='cuda', generator=g)
ix = FlatIPIndex(768); ix.add(x)
q = torch.randn(nq, 768, device='cuda', generator=g)
for _ in range(3): ix.search_tensors(q, 100)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): ix.search_tensors(q, 100)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f'QG {os.environ.get("LDOT_DEBUG_QG", "auto")} nq {nq}: {dt*1e3:.3f} ms', flush=True)
