import sys, time, torch
sys.path.insert(0, '/root/repo')
from lightningdot_amd.indexer import FlatIPIndex
g = torch.Generator(device='cuda').manual_seed(0)
x = torch.randn(1000000, 768, device='cuda', generator=g)
ix = FlatIPIndex(768); ix.add(x)
for nq in (300, 600, 1300, 2304, 3000, 5000, 6000, 10000):
    q = torch.randn(nq, 768, device='cuda', generator=g)
    for _ in range(3): ix.search_tensors(q, 100)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ix.search_tensors(q, 100)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f'nq {nq}: {dt*1e3:.3f} ms  {2*nq*1e6*768/dt/1e12:.0f} TFLOP/s end to end', flush=True)
