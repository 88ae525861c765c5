"""Approximate inner-product index — the MI355X counterpart of the reference's ``--hnsw_index`` alternative
(dvl/indexer/faiss_indexers.py:90-154, faiss.IndexHNSWFlat over phi-augmented vectors).

    DenseIVFFlatIndexer   same DenseIndexer surface (index_data / index_tensor / search_knn / serialize / deserialize_from)

A graph walk (HNSW) is a chain of dependent, scattered reads — a poor fit for a 256-CU machine.  The same job — look at a small part
of the index per query — is done here with an inverted file: the rows are clustered once (k-means in the reference's own
inner-product -> L2 space: x~ = [x, sqrt(phi - |x|^2)], phi = max |x|^2, faiss_indexers.py:114-131), stored SORTED BY LIST in an
ordinary exact index, and a query is scored exactly (fp32, ``ldot_index_search_lists``) against the rows of the ``nprobe`` lists whose
centroids are nearest to q~ = [q, 0].  Which lists to probe is itself an exact search over the centroids with the library
(L2-nearest = largest q~.c~ - |c~|^2 / 2, one more coordinate); coarse search, list scan (a per-query compact column space over the
probed lists) and selection are ONE library call, ``ldot_ivf_search``.  Scores are exact inner products of the rows that were looked
at; what is approximate is WHICH rows are looked at (recall < 1, like HNSW).  ``nprobe = nlist`` degenerates to the exact search.
A batch for which scanning the probed lists would take longer than the exact search (which reuses every index tile across the batch)
is answered by the exact search (``exact_when_cheaper``, on by default; ``last_route`` tells which ran).

Training (k-means) is build-time host code on the device through torch (plain library GEMMs); everything on the query path is the
HIP library.
"""
import ctypes
import logging
import pickle
from typing import List, Optional, Tuple

import numpy as np

from . import _lib as L
from .indexer import DenseIndexer, FlatIPIndex, _describe, _is_tensor, _stream_ptr

logger = logging.getLogger()


def _kmeans_l2(x, nlist: int, iters: int, seed: int):
    """Lloyd's algorithm on the device (x [n, d] fp32 CUDA tensor): -> centroids [nlist, d].  Assignment = argmax(x.c - |c|^2/2)."""
    import torch
    g = torch.Generator(device='cpu').manual_seed(seed)
    n = x.shape[0]
    cent = x[torch.randperm(n, generator=g)[:nlist].to(x.device)].clone()
    for _ in range(iters):
        assign = _assign_l2(x, cent)
        sums = torch.zeros_like(cent).index_add_(0, assign, x)
        cnt = torch.zeros(nlist, device=x.device, dtype=x.dtype).index_add_(0, assign, torch.ones(n, device=x.device, dtype=x.dtype))
        empty = cnt == 0
        cent = torch.where(empty[:, None], cent, sums / cnt.clamp_min(1)[:, None])
        if bool(empty.any()):            # re-seed empty clusters from random rows
            idx = torch.randint(0, n, (int(empty.sum()),), generator=g).to(x.device)
            cent[empty] = x[idx]
    return cent


def _split_long_lists(x, cent, assign, cap: int, seed: int, max_rounds: int = 8):
    """Bound the list length: every list longer than `cap` rows is split in two by a short 2-means on its own rows (repeated until none
    is left or nothing changes); the halves take their means as centroids, rows never move to other lists.  k-means on real embeddings
    leaves a heavy tail of list lengths (1M rows of an overlapping mixture in 4000 lists: longest 5472 rows against a mean of 250) and a
    query lands in — and next to — the long lists, so 32 probes read 64 000 rows instead of 8 000; a search that reads fewer rows per
    probe reaches the same recall with more probes at a lower cost (profiles/r03_ivf_recall_curve.jsonl).  -> (centroids, assign)"""
    import torch
    g = torch.Generator(device='cpu').manual_seed(seed + 1)
    for _ in range(max_rounds):
        counts = torch.bincount(assign, minlength=cent.shape[0])
        big = (counts > cap).nonzero()[:, 0].tolist()
        if not big:
            break
        order = torch.argsort(assign, stable=True)
        offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=x.device), counts.cumsum(0)]).tolist()
        new_cent, changed = [], False
        for li in big:
            idx = order[offs[li]:offs[li + 1]]
            xs = x[idx]
            c2 = xs[torch.randperm(xs.shape[0], generator=g)[:2].to(x.device)].clone()
            for _ in range(6):
                part = _assign_l2(xs, c2)
                for h in (0, 1):
                    m = part == h
                    if bool(m.any()):
                        c2[h] = xs[m].mean(0)
            part = _assign_l2(xs, c2)
            n1 = int((part == 1).sum())
            if n1 == 0 or n1 == xs.shape[0]:      # (identical rows: cannot be split by distance — cut the list in the middle)
                part = (torch.arange(xs.shape[0], device=x.device) >= xs.shape[0] // 2).long()
            c2 = torch.stack([xs[part == 0].mean(0), xs[part == 1].mean(0)])   # the halves' centroids = the means of their final members
            cent[li] = c2[0]
            assign[idx[part == 1]] = cent.shape[0] + len(new_cent)
            new_cent.append(c2[1])
            changed = True
        cent = torch.cat([cent, torch.stack(new_cent)], 0)
        if not changed:
            break
    return cent, assign


def _assign_l2(x, cent, chunk: int = 131072):
    import torch
    half = 0.5 * (cent * cent).sum(1)
    out = []
    for i in range(0, x.shape[0], chunk):
        out.append((x[i:i + chunk] @ cent.T - half[None, :]).argmax(1))
    return torch.cat(out)


class DenseIVFFlatIndexer(DenseIndexer):
    def __init__(self, vector_sz: int, buffer_size: int = 50000, nlist: Optional[int] = None, nprobe: int = 32,
                 train_iters: int = 10, train_rows_per_list: int = 256, seed: int = 0, max_list_rows: Optional[int] = None):
        super().__init__(buffer_size=buffer_size)
        self.d = vector_sz
        self.index = FlatIPIndex(vector_sz)              # the rows, sorted by list
        # (cluster-sorted rows are the order the optimistic thresholds of the exact scan must not assume away: a query's best rows
        # sit together, often early — the scan would flag and redo most queries; the library also backs off by itself.  The scrambled
        # scan order does not help here: it permutes 384-row tiles, and a list of ~250 rows IS a tile)
        self._configure_row_index()
        self.nlist, self.nprobe = nlist, nprobe
        self.train_iters, self.train_rows_per_list, self.seed = train_iters, train_rows_per_list, seed
        # longest list allowed (None: 4 x the mean list length, at least 64 rows, when nlist is chosen automatically; 0: lists as
        # k-means leaves them)
        self.max_list_rows = max_list_rows
        self.coarse: Optional[FlatIPIndex] = None        # the centroids in the augmented space (+ the -|c~|^2/2 coordinate)
        self.list_offsets = None                          # int64 [nlist + 1], device
        self.max_list_len = 0
        self.phi = 0.0
        self.biased_list_len = 0.0                        # sum(len^2) / sum(len): expected length of a probed list
        self.last_route = None                            # 'lists' or 'exact': what the last search did

    # ---- build (all data at once, like the reference's HNSW indexer :111-113) --------------------------------------------------
    def index_data(self, data: List[Tuple[object, np.array]]):
        import torch
        ids = [t[0] for t in data]
        if data and _is_tensor(data[0][1]):
            vecs = torch.stack([t[1].reshape(-1) for t in data], 0)
        else:
            vecs = torch.from_numpy(np.concatenate([np.reshape(t[1], (1, -1)) for t in data], 0).astype(np.float32))
        self.index_tensor(ids, vecs)

    def index_tensor(self, db_ids: List, vectors):
        import torch
        if self.coarse is not None:
            raise RuntimeError('the IVF index needs to index all data at once (like the reference\'s HNSW indexer)')
        if len(db_ids) != vectors.shape[0]:
            raise ValueError('ids / vectors length mismatch')
        x = vectors.detach().float().cuda()
        n = x.shape[0]
        # ~4 sqrt(n) lists (4000 for 1M rows, ~250 rows each): the scan pads every probed list to the LONGEST one, so many short
        # lists beat few long ones (1000 lists on clustered data: longest 27k rows, 0.39 ms per query at nprobe 32)
        nlist = self.nlist or int(min(16384, max(1, round(4 * n ** 0.5))))
        nlist = max(1, min(nlist, n))
        sq = (x * x).sum(1)
        self.phi = float(sq.max().item()) if n else 0.0
        aug = torch.cat([x, (self.phi - sq).clamp_min(0).sqrt()[:, None]], 1)   # faiss_indexers.py:123-126
        g = torch.Generator(device='cpu').manual_seed(self.seed)
        ntrain = min(n, nlist * self.train_rows_per_list)
        sample = aug if ntrain == n else aug[torch.randperm(n, generator=g)[:ntrain].to(x.device)]
        cent = _kmeans_l2(sample, nlist, self.train_iters, self.seed)
        assign = _assign_l2(aug, cent)
        # (an explicit nlist is kept as asked for unless max_list_rows is given too)
        cap = self.max_list_rows if self.max_list_rows is not None else (0 if self.nlist else max(64, 4 * -(-n // nlist)))
        if cap > 0:
            cent, assign = _split_long_lists(aug, cent, assign, int(cap), self.seed)
            nlist = cent.shape[0]
        order = torch.argsort(assign, stable=True)
        counts = torch.bincount(assign, minlength=nlist)
        self.list_offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=x.device), counts.cumsum(0)]).contiguous()
        self.max_list_len = int(counts.max().item()) if n else 0
        self.biased_list_len = float((counts.double() ** 2).sum().item() / max(n, 1))
        order_h = order.cpu().tolist()
        self._update_id_mapping([db_ids[i] for i in order_h])                    # label (sorted row) -> external id
        self.index.add(x[order])
        self.nlist = nlist
        self._set_coarse(cent)
        logger.info('IVF: %d rows in %d lists (longest %d), phi=%g', n, nlist, self.max_list_len, self.phi)

    def _set_coarse(self, cent):
        import torch
        self._centroids = cent
        self.coarse = FlatIPIndex(cent.shape[1] + 1)
        self.coarse.set_option(L.OPT_ROW_SHUFFLE, 2)     # ldot_ivf_search takes the probes as stored rows: the centroids must never be re-ordered
        self.coarse.add(torch.cat([cent, -0.5 * (cent * cent).sum(1, keepdim=True)], 1))

    # ---- search ------------------------------------------------------------------------------------------------------------------
    def _exact_is_cheaper(self, nq: int, nprobe: int) -> bool:
        """Cost model from the measured rates on MI355X (profiles/r03_ivf_bench.jsonl, r03_serving_latency.jsonl).  List search: ~0.08 ms
        of launches and latency, then the probed rows once per query — fp32 rows at ~4 TB/s for a few queries, the bf16 shadow at
        ~5.5 TB/s for 8-16 queries, fp32 rows at ~9 TB/s for batches (neighbouring queries probe the same lists: L2 hits).  Exact
        search: the bf16 index once at ~6.3 TB/s for <= 64 queries (+17 % per 16 queries of score traffic), ~1.3 PFLOP/s plus ~0.35 ms
        of fixed cost for larger batches."""
        n, d = self.index.ntotal, self.d
        # probed rows per query: a query lands in (and next to) a list with probability proportional to its size, so the expected
        # length of a probed list is the size-biased mean sum(len^2) / sum(len), not n / nlist — and its neighbours are long lists too
        # (measured on clustered data: twice the size-biased mean again)
        rows = min(n, 2.0 * nprobe * self.biased_list_len)
        if nq < 8:
            t_lists = 0.08e-3 + nq * rows * d * 4 / 4.0e12
        elif nq <= 16:
            t_lists = 0.10e-3 + nq * rows * d * 2 / 5.5e12
        else:
            t_lists = 0.08e-3 + nq * 0.35e-6 + nq * rows * d * 4 / 9.0e12
        if nq <= 64:
            t_exact = 0.06e-3 + n * d * 2 / 6.3e12 * (1 + 0.17 * ((nq - 1) // 16))
        else:
            t_exact = 0.35e-3 + 2.0 * nq * n * d / 1.3e15
        return t_exact < t_lists

    def search_knn_tensors(self, query_vectors, top_docs: int, nprobe: Optional[int] = None, exact_when_cheaper: bool = True):
        """(scores [nq, k] fp32 descending, row labels [nq, k] int64; -1 padding) as device tensors.  ``exact_when_cheaper``: answer
        with the exact search when the cost model says it is faster than scanning the probed lists (large batches)."""
        import torch
        if self.coarse is None:
            raise RuntimeError('the IVF index is empty')
        q = query_vectors if _is_tensor(query_vectors) else torch.from_numpy(np.asarray(query_vectors, dtype=np.float32))
        q = q.detach().float().cuda().contiguous()
        if q.dim() == 1:
            q = q[None]
        nq = q.shape[0]
        nprobe = min(int(nprobe or self.nprobe), self.nlist)
        if exact_when_cheaper and self._exact_is_cheaper(nq, nprobe):
            # a batch re-reads its probed fp32 rows once per query, the exact search reuses every bf16 index tile across the whole batch:
            # from a handful of queries on, the exact answer (recall 1.0, same row labels) is also the faster one
            self.last_route = 'exact'
            return self.index.search_tensors(q, top_docs)
        self.last_route = 'lists'
        scores = torch.empty((nq, top_docs), dtype=torch.float32, device=q.device)
        labels = torch.empty((nq, top_docs), dtype=torch.int64, device=q.device)
        ix = self.index
        # one library call: coarse query [q, 0 | 1] -> nprobe nearest lists -> exact scan of their rows -> top-k
        L.check(ix._lib.ldot_ivf_search(ix._h, self.coarse._h, ctypes.c_void_p(q.data_ptr()), nq, L.F32, 0,
                                        ctypes.c_void_p(self.list_offsets.data_ptr()), int(self.max_list_len), nprobe, int(top_docs),
                                        ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(labels.data_ptr()), L.DEVICE,
                                        _stream_ptr(ix.device)))
        return scores, labels

    def search_lists_tensors(self, query_vectors, probes, top_docs: int):
        """the scan step alone: probes [nq, nprobe] int32 list ids chosen by the caller (-1 = skip)"""
        import torch
        q = query_vectors.detach().float().cuda().contiguous()
        probes = probes.to(device=q.device, dtype=torch.int32).contiguous()
        nq, nprobe = probes.shape
        scores = torch.empty((nq, top_docs), dtype=torch.float32, device=q.device)
        labels = torch.empty((nq, top_docs), dtype=torch.int64, device=q.device)
        ix = self.index
        L.check(ix._lib.ldot_index_search_lists(ix._h, ctypes.c_void_p(q.data_ptr()), nq, L.F32, 0,
                                                ctypes.c_void_p(self.list_offsets.data_ptr()), int(self.nlist),
                                                int(self.max_list_len), ctypes.c_void_p(probes.data_ptr()), nprobe, int(top_docs),
                                                ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(labels.data_ptr()), L.DEVICE,
                                                _stream_ptr(ix.device)))
        return scores, labels

    def search_knn(self, query_vectors, top_docs: int, nprobe: Optional[int] = None):
        s, l = self.search_knn_tensors(query_vectors, top_docs, nprobe)
        s, l = s.cpu().numpy(), l.cpu().tolist()
        ids = self.index_id_to_db_id
        return [([ids[i] for i in row], s[j]) for j, row in enumerate(l)]        # (-1 -> last id, the reference's :85 behaviour)

    def _configure_row_index(self):
        """options of the row store (also re-applied to a freshly loaded index: ldot_index_load returns the defaults)"""
        self.index.set_option(L.OPT_OPTIMISTIC, 0)
        self.index.set_option(L.OPT_ROW_SHUFFLE, 2)      # lists are row ranges: the store must never be re-ordered

    # ---- persistence: the reference's two files + the clustering in the meta file ---------------------------------------------------
    def serialize(self, file: str):
        self.index.save(file + '.index.dpr')
        with open(file + '.index_meta.dpr', mode='wb') as f:
            pickle.dump({'ids': self.index_id_to_db_id, 'list_offsets': self.list_offsets.cpu().numpy(),
                         'centroids': self._centroids.cpu().numpy(), 'phi': self.phi, 'nprobe': self.nprobe}, f)

    def deserialize_from(self, file: str):
        import torch
        self.index = FlatIPIndex.load(file + '.index.dpr')
        self._configure_row_index()
        with open(file + '.index_meta.dpr', 'rb') as f:
            m = pickle.load(f)
        self.index_id_to_db_id = m['ids']
        assert len(self.index_id_to_db_id) == self.index.ntotal, 'Deserialized index_id_to_db_id should match faiss index size'
        self.list_offsets = torch.from_numpy(m['list_offsets']).cuda()
        self.max_list_len = int(np.diff(m['list_offsets']).max()) if len(m['list_offsets']) > 1 else 0
        lens = np.diff(m['list_offsets']).astype(np.float64)
        self.biased_list_len = float((lens ** 2).sum() / max(lens.sum(), 1.0))
        self.nlist = len(m['list_offsets']) - 1
        self.phi, self.nprobe = m['phi'], m['nprobe']
        self._set_coarse(torch.from_numpy(m['centroids']).cuda())
