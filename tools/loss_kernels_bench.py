#!/usr/bin/env python3
"""GPU time of the in-batch loss kernels at the config-5 shapes, host overhead excluded: N calls are enqueued back to back through the C
ABI (ctypes, no autograd objects) and the stream is synchronised once.  Beside them the same mathematics as torch ops (rocBLAS fp32 GEMMs +
log_softmax + nll_loss forward; softmax backward + two GEMMs backward), enqueued the same way."""
import ctypes, os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd import _lib as L
lib = L.load_library()
P = lambda t: ctypes.c_void_p(t.data_ptr() if t is not None else 0)
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

def gpu_us(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

torch.manual_seed(99)
SHAPES = ((4096, 4096, 768, 0.0),) if '--big' in sys.argv else ((512, 512, 768, 0.0), (512, 1536, 768, 0.0), (512, 1536, 768, 0.1))
for n1, n2, d, w in SHAPES:
    q = torch.randn(n1, d, device='cuda'); c = torch.randn(n2, d, device='cuda')
    cap = torch.randn(n2, d, device='cuda') if w else None
    pos = torch.arange(n1, device='cuda', dtype=torch.int32); pos64 = pos.long()
    S = torch.empty(n1, n2, device='cuda'); rl = torch.empty(n1, device='cuda'); lse = torch.empty(n1, device='cuda')
    cor = torch.empty(1, device='cuda', dtype=torch.int32); ls = torch.empty(1, device='cuda')
    g_row = torch.full((n1,), 1.0 / n1, device='cuda'); ds = torch.empty(n1, n2, device='cuda')
    dq = torch.empty_like(q); dc = torch.empty_like(c); dcap = torch.empty_like(c) if w else None
    def fwd():
        L.check(lib.ldot_inbatch_nll_fwd(P(q), P(c), P(cap), float(w), P(pos), n1, n2, d, P(S), P(rl), P(lse), P(cor), P(ls), st()))
    def bwd():
        L.check(lib.ldot_inbatch_nll_bwd(P(q), P(c), P(cap), float(w), P(pos), n1, n2, d, P(S), P(lse), P(g_row), P(None), P(ds), P(dq), P(dc),
                                         P(dcap), st()))
    def t_fwd():
        s = q @ c.t()
        if w: s = (1 - w) * s + w * (q @ cap.t())
        lp = F.log_softmax(s, dim=1)
        return s, lp, F.nll_loss(lp, pos64), (lp.argmax(1) == pos64).sum()
    s_t, lp_t, _, _ = t_fwd()
    def t_bwd():
        g = torch.exp(lp_t); g[torch.arange(n1), pos64] -= 1.0; g *= 1.0 / n1
        a = g @ c; b = g.t() @ q
        if w: a = (1 - w) * a + w * (g @ cap); b2 = g.t() @ q
        return a, b
    def gemm_only():
        L.check(lib.ldot_dot_product_scores(P(q), P(c), n1, n2, d, P(S), st()))
    fwd(); bwd(); gemm_only()
    for _ in range(200): gemm_only()        # (shows up in a kernel trace as sgemm_direct_kernel<true, true, false, 0>)
    print('n1=%d n2=%d d=%d w=%.1f: forward %.1f us (torch ops %.1f), backward %.1f us (torch ops %.1f)'
          % (n1, n2, d, w, gpu_us(fwd), gpu_us(t_fwd), gpu_us(bwd), gpu_us(t_bwd)), flush=True)
