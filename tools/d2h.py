import torch, time
for mb in (4, 8, 64, 256):
    n = mb * 1024 * 1024
    d = torch.empty(n, dtype=torch.uint8, device='cuda')
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    for _ in range(3): h.copy_(d, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print('D2H %d MB: %.1f us  %.1f GB/s' % (mb, dt * 1e6, n / dt / 1e9))
    t0 = time.perf_counter()
    for _ in range(10): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print('H2D %d MB: %.1f us  %.1f GB/s' % (mb, dt * 1e6, n / dt / 1e9))
