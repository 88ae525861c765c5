import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
x = torch.randn(n, 768, device='cuda'); q = torch.randn(10000, 768, device='cuda')
ix = FlatIPIndex(768); ix.add(x)
for _ in range(6):
    ix.search_begin(q, 100); ix.search_finish(None)
torch.cuda.synchronize()
