import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
x = torch.randn(1_000_000, 768, device='cuda')
ix = FlatIPIndex(768); ix.add(x)
q = x[:1] + 0.5 * torch.randn(1, 768, device='cuda')
for _ in range(6):
    ix.search_tensors(q, 100)
torch.cuda.synchronize()
