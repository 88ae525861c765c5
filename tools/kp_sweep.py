import sys, time, torch
sys.path.insert(0, '/root/repo')
from lightningdot_amd import _lib as L
from lightningdot_amd.indexer import FlatIPIndex
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(1000000, 768, device='cuda', generator=g)
q = x[(torch.arange(10000, device='cuda') * 9973) % 1000000] + 0.5 * torch.randn(10000, 768, device='cuda', generator=g)
ix = FlatIPIndex(768); ix.add(x)
for rnd in range(2):
    for k, margin in ((100, -1), (96, 0), (64, 0), (100, 60)):
        ix.set_option(L.OPT_MARGIN, margin)
        for _ in range(3): ix.search_tensors(q, k)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15): ix.search_tensors(q, k)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 15
        print(f'k {k} margin {margin}: {dt*1e3:.3f} ms', flush=True)
