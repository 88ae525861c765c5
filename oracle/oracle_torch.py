"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (timed baseline leg of bench.py).

Threaded fp32 restatement of the reference's CPU scorer for the ``cpu_baseline`` figure: the reference scores with
``faiss.IndexFlatIP.search`` (dvl/indexer/faiss_indexers.py:83; faiss-cpu==1.6.3, DVL.yml:80 — third-party, absent here), i.e. a blocked
fp32 sgemm over query tiles x index blocks followed by a per-query k-selection, parallelised over the host cores.  SURVEY §8d
prescribes this stand-in when faiss is not importable: ``torch.matmul`` over 4096-query x N-block tiles + ``torch.topk(k, sorted=True)``
with ``torch.set_num_threads(cores)`` (both threaded by torch's intra-op pool).  Same arithmetic as ``oracle_np.FlatIP`` (exact fp32
inner products); ties are not canonicalised (checked against ``oracle_np`` in tests/test_oracle_golden.py).

Only ``tests/`` and ``bench.py``'s cpu_baseline leg may import this module.
"""
from __future__ import annotations

import time
from typing import Tuple

import numpy as np
import torch

NEG_FLT_MAX = -float(np.finfo(np.float32).max)


def have_faiss() -> bool:
    try:
        import faiss  # noqa: F401
        return True
    except Exception:
        return False


def search_blocked(q: torch.Tensor, x: torch.Tensor, k: int, q_tile: int = 4096, n_block: int = 131072) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact fp32 inner-product top-k on the host: (scores [nq, k] descending, labels [nq, k] int64; -FLT_MAX / -1 padding)."""
    nq, n = q.shape[0], x.shape[0]
    kk = min(k, n)
    out_s = torch.full((nq, k), NEG_FLT_MAX, dtype=torch.float32)
    out_l = torch.full((nq, k), -1, dtype=torch.int64)
    for q0 in range(0, nq, q_tile):
        qt = q[q0:q0 + q_tile]
        best_s = best_l = None
        for n0 in range(0, n, n_block):
            s = torch.matmul(qt, x[n0:n0 + n_block].T)                 # fp32 sgemm (threaded)
            ps, pl = torch.topk(s, min(kk, s.shape[1]), dim=1, sorted=True)
            pl = pl + n0
            if best_s is None:
                best_s, best_l = ps, pl
            else:                                                      # running merge of the per-block lists
                cs, cl = torch.cat([best_s, ps], 1), torch.cat([best_l, pl], 1)
                best_s, idx = torch.topk(cs, kk, dim=1, sorted=True)
                best_l = torch.gather(cl, 1, idx)
        if best_s is not None:
            out_s[q0:q0 + qt.shape[0], :best_s.shape[1]] = best_s
            out_l[q0:q0 + qt.shape[0], :best_l.shape[1]] = best_l
    return out_s, out_l


def faiss_search(q: np.ndarray, x: np.ndarray, k: int):
    """The reference's own scorer (used instead of the stand-in when faiss is importable on the box)."""
    import faiss
    ix = faiss.IndexFlatIP(x.shape[1])
    ix.add(x)
    return ix.search(q, k)


def timed(q: torch.Tensor, x: torch.Tensor, k: int, threads: int, runs: int = 5):
    """Median wall time of `runs` searches after one warm-up, at `threads` intra-op threads -> (seconds, scores, labels)."""
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, threads))
    try:
        search_blocked(q[:min(64, q.shape[0])], x[:min(8192, x.shape[0])], k)      # warm the thread pool
        ts = []
        res = None
        for _ in range(runs):
            t0 = time.perf_counter()
            res = search_blocked(q, x, k)
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], res[0], res[1]
    finally:
        torch.set_num_threads(old)
