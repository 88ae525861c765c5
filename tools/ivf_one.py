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
# probed rows of this query (what the scan kernel gathers)
qa = torch.cat([q, torch.zeros(1, 1, device='cuda'), torch.ones(1, 1, device='cuda')], 1)
_, probes = ivf.coarse.search_tensors(qa, 32)
lens = (ivf.list_offsets[1:] - ivf.list_offsets[:-1])[probes[0]]
print('probed rows', int(lens.sum()), 'longest probed list', int(lens.max()), 'bytes', int(lens.sum()) * D * 4, flush=True)
