"""One-query inverted-file search on the mixture of tools/ivf_bench.py (4000 overlapping Gaussians, 1M x 768), the searches LAST so
that tools/timeline_tail.sh shows the kernels of a search at the end of the trace.  Usage: python tools/ivf_one.py [centroid_spread]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.ivf import DenseIVFFlatIndexer
SPREAD = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
g = torch.Generator(device='cuda').manual_seed(0)
N, D = 1_000_000, 768
cent = SPREAD * torch.randn(4000, D, device='cuda', generator=g)
x = cent[torch.randint(0, 4000, (N,), device='cuda', generator=g)] + 0.5 * torch.randn(N, D, device='cuda', generator=g)
q = cent[torch.randint(0, 4000, (1,), device='cuda', generator=g)] + 0.5 * torch.randn(1, D, device='cuda', generator=g)
ivf = DenseIVFFlatIndexer(D, nprobe=32); ivf.index_tensor(list(range(N)), x)
for _ in range(5): ivf.search_knn_tensors(q, 10, exact_when_cheaper=False)
torch.cuda.synchronize()
# probed rows of this query (what the scan kernel gathers)
qa = torch.cat([q, torch.zeros(1, 1, device='cuda'), torch.ones(1, 1, device='cuda')], 1)
_, probes = ivf.coarse.search_tensors(qa, 32)
lens = (ivf.list_offsets[1:] - ivf.list_offsets[:-1])[probes[0]]
print('nlist', ivf.nlist, 'longest list', ivf.max_list_len, 'probed rows', int(lens.sum()), 'longest probed list', int(lens.max()),
      'bytes', int(lens.sum()) * D * 4, flush=True)
for _ in range(20): ivf.search_knn_tensors(q, 10, exact_when_cheaper=False)
torch.cuda.synchronize()
