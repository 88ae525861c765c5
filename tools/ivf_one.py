import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.ivf import DenseIVFFlatIndexer
g = torch.Generator(device='cuda').manual_seed(0)
N, D = 1_000_000, 768
cent = torch.randn(4000, D, device='cuda', generator=g)
x = cent[torch.randint(0, 4000, (N,), device='cuda', generator=g)] + 0.5 * torch.randn(N, D, device='cuda', generator=g)
ivf = DenseIVFFlatIndexer(D, nprobe=32); ivf.index_tensor(list(range(N)), x)
q = x[:1] + 0.3
for _ in range(20): ivf.search_knn_tensors(q, 10)
torch.cuda.synchronize()
