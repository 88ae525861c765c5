import os, sys, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
for n in (512, 4096, 12288):
    x = torch.randn(n, 768, device='cuda'); ix = FlatIPIndex(768); ix.add(x)
    q = torch.randn(1, 768, device='cuda')
    for _ in range(6): ix.search_tensors(q, 10)
torch.cuda.synchronize()
