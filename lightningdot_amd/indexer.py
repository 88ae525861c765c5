"""Dense inner-product indexers — host-side mirror of dvl/indexer/faiss_indexers.py (reference), backed by the
MI355X HIP library instead of faiss-cpu.

    DenseIndexer        <- faiss_indexers.py:22-60   (id map, serialize / deserialize_from, two-file naming)
    DenseFlatIndexer    <- faiss_indexers.py:63-87   (IndexFlatIP: index_data / search_knn)
    FlatIPIndex         <- the faiss.IndexFlatIP object itself (ctor :67, add :77, search :83, ntotal :52,
                           write_index :41 / read_index :51)

Same names, argument meaning and return types as the reference (results sorted by descending score, one
``(ids, scores)`` tuple per query, scores as float32 ndarray).  Additions: device tensors are accepted everywhere
(no numpy round trip), ``search_knn_tensors`` returns device tensors, ``normalize=True`` is the opt-in
L2 variant.  There is no CPU fallback: constructing an index without the HIP library / a GPU raises.
"""
import ctypes
import logging
import pickle
from typing import List, Optional, Tuple

import numpy as np

from . import _lib as L

logger = logging.getLogger()


def _is_tensor(x):
    return type(x).__module__.startswith('torch') and hasattr(x, 'data_ptr')


def _stream_ptr(device=None):
    import torch
    if torch.cuda.is_available():
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    return ctypes.c_void_p(0)


def _describe(x, d, device=None):
    """-> (keepalive, pointer, n_rows, dtype_code, mem_code) for a [n, d] array / tensor (device tensors must live on the
    index's device: the library runs every call there, whatever the caller's current device is)."""
    if _is_tensor(x):
        import torch
        t = x.detach()
        if t.dim() == 1:
            t = t.view(1, -1)
        if t.dim() != 2 or t.shape[1] != d:
            raise ValueError(f'expected [n, {d}] vectors, got {tuple(t.shape)}')
        code = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}.get(t.dtype)
        if code is None:
            t = t.float()
            code = L.F32
        t = t.contiguous()
        if t.is_cuda and device is not None and t.device.index != device:
            raise ValueError(f'tensor on cuda:{t.device.index} but the index lives on cuda:{device}')
        return t, ctypes.c_void_p(t.data_ptr()), t.shape[0], code, (L.DEVICE if t.is_cuda else L.HOST)
    a = np.asarray(x)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.ndim != 2 or a.shape[1] != d:
        raise ValueError(f'expected [n, {d}] vectors, got {a.shape}')
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, ctypes.c_void_p(a.ctypes.data), a.shape[0], L.F32, L.HOST


class FlatIPIndex:
    """Exact inner-product index resident in HBM (fp32 master copy + bf16 MFMA shadow).  Stands where the
    reference holds a ``faiss.IndexFlatIP`` (faiss_indexers.py:67)."""

    def __init__(self, d: int, normalize: bool = False, _handle=None):
        self._lib = L.load_library()
        self.normalize = bool(normalize)
        if _handle is not None:
            self._h = _handle
        else:
            h = ctypes.c_void_p()
            L.check(self._lib.ldot_index_create(int(d), ctypes.byref(h)))
            self._h = h
        self.d = int(self._lib.ldot_index_dim(self._h))
        self._pending = None
        self._opts = {}        # what the caller set through set_option (search(verify=True) restores them afterwards)
        try:
            import torch
            self.device = torch.cuda.current_device() if torch.cuda.is_available() else None   # where the library created it
        except Exception:  # pragma: no cover
            self.device = None

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h is not None and getattr(self, '_lib', None) is not None:
            try:
                self._lib.ldot_index_destroy(h)
            except Exception:  # pragma: no cover
                pass

    @property
    def ntotal(self) -> int:
        return int(self._lib.ldot_index_ntotal(self._h))

    def set_option(self, option: int, value: int):
        L.check(self._lib.ldot_index_set_option(self._h, int(option), int(value)))
        self._opts[int(option)] = int(value)

    def reset(self):
        L.check(self._lib.ldot_index_reset(self._h))

    def add(self, vectors):
        keep, ptr, n, dt, mem = _describe(vectors, self.d, self.device)
        L.check(self._lib.ldot_index_add(self._h, ptr, n, dt, mem, int(self.normalize), _stream_ptr(self.device)))
        if mem == L.DEVICE:
            import torch
            torch.cuda.current_stream().synchronize()   # `keep` may be a temporary
        del keep

    def search(self, queries, k: int, verify: bool = False):
        """faiss-style: (scores [nq, k] float32, labels [nq, k] int64) as numpy arrays.

        ``verify=True``: the bf16 candidate pass is exact whenever the true top-k lies inside the bf16 top-k' (k' = k + margin).
        With verification the library flags every query whose k-th exact score is not above the candidate threshold by a
        statistical bf16 error bound (LDOT_OPT_VERIFY, see ldot.h: 4 sigma of independent rounding errors, not the worst case) and
        the flagged queries are searched again with a 4x larger margin than the one in force, repeatedly, up to the library's
        maximum; ``last_unproven`` then holds the number of queries that are still flagged.  Margin and verify settings made
        through ``set_option`` are in force again when the call returns."""
        keep, ptr, nq, dt, mem = _describe(queries, self.d, self.device)
        scores = np.empty((nq, k), dtype=np.float32)
        labels = np.empty((nq, k), dtype=np.int64)
        user_verify, user_margin = self._opts.get(L.OPT_VERIFY, 0), self._opts.get(L.OPT_MARGIN, -1)
        if verify:
            self.set_option(L.OPT_VERIFY, 1)
        try:
            L.check(self._lib.ldot_index_search(self._h, ptr, nq, dt, mem, int(self.normalize), int(k),
                                                ctypes.c_void_p(scores.ctypes.data), ctypes.c_void_p(labels.ctypes.data),
                                                L.HOST, _stream_ptr(self.device)))
            if verify and nq:
                self._escalate(keep, k, scores, labels, user_margin)
        finally:
            if verify:
                self.set_option(L.OPT_VERIFY, user_verify)
                self.set_option(L.OPT_MARGIN, user_margin)
        del keep
        return scores, labels

    def unproven(self, nq: int):
        """(flags [nq] int32, count) of the last search made with LDOT_OPT_VERIFY = 1."""
        flags = np.zeros((nq,), dtype=np.int32)
        cnt = ctypes.c_int64(0)
        L.check(self._lib.ldot_index_last_unproven(self._h, ctypes.c_void_p(flags.ctypes.data), ctypes.byref(cnt)))
        return flags, int(cnt.value)

    def _escalate(self, queries, k, scores, labels, user_margin=-1):
        flags, cnt = self.unproven(scores.shape[0])
        todo = np.nonzero(flags)[0]
        margin = max(28, k // 4) if user_margin < 0 else user_margin      # the margin the first search ran with
        self.last_escalations = []
        while len(todo) and margin < L.MAX_MARGIN:
            margin = min(4 * margin, L.MAX_MARGIN)
            self.set_option(L.OPT_MARGIN, margin)
            sub = queries[todo] if not _is_tensor(queries) else queries[todo.tolist()]
            keep, ptr, n, dt, mem = _describe(sub, self.d, self.device)
            s2 = np.empty((n, k), dtype=np.float32)
            l2 = np.empty((n, k), dtype=np.int64)
            L.check(self._lib.ldot_index_search(self._h, ptr, n, dt, mem, int(self.normalize), int(k),
                                                ctypes.c_void_p(s2.ctypes.data), ctypes.c_void_p(l2.ctypes.data), L.HOST,
                                                _stream_ptr(self.device)))
            scores[todo], labels[todo] = s2, l2
            f2, _ = self.unproven(n)
            self.last_escalations.append((margin, int(n)))
            todo = todo[np.nonzero(f2)[0]]
        self.last_unproven = int(len(todo))

    def search_into(self, queries, k: int, out_scores, out_labels, sync: bool = True):
        """Search with caller-owned HOST outputs (float32 [nq, k] / int64 [nq, k]; numpy arrays or CPU tensors — pinned memory
        makes the result copies asynchronous: the library re-scores in chunks and ships every chunk while the next one is
        re-scored).  Returns when the results are in the buffers.

        ``sync=False`` (LDOT_OPT_DEFER_SYNC, pinned outputs): return once the search is enqueued; the results are in the buffers
        after the next ``search_into(..., sync=True)`` on the same stream or a stream synchronisation.  ``queries`` must be a CUDA
        tensor that stays alive and unchanged until then.  An evaluation runs its first direction this way."""
        if bool(self._opts.get(L.OPT_DEFER_SYNC, 0)) != (not sync):
            self.set_option(L.OPT_DEFER_SYNC, 0 if sync else 1)
        keep, ptr, nq, dt, mem = _describe(queries, self.d, self.device)

        def host_ptr(buf, itemsize):
            if _is_tensor(buf):
                if buf.is_cuda or not buf.is_contiguous() or tuple(buf.shape) != (nq, k) or buf.element_size() != itemsize:
                    raise ValueError('output buffers must be contiguous CPU tensors of shape [nq, k]')
                return ctypes.c_void_p(buf.data_ptr())
            if not buf.flags['C_CONTIGUOUS'] or buf.shape != (nq, k) or buf.itemsize != itemsize:
                raise ValueError('output buffers must be C-contiguous arrays of shape [nq, k]')
            return ctypes.c_void_p(buf.ctypes.data)

        L.check(self._lib.ldot_index_search(self._h, ptr, nq, dt, mem, int(self.normalize), int(k),
                                            host_ptr(out_scores, 4), host_ptr(out_labels, 8), L.HOST, _stream_ptr(self.device)))
        del keep
        return out_scores, out_labels

    def search_tensors(self, queries, k: int, ids_only: bool = False):
        """Device-resident variant: queries is a CUDA tensor; returns (scores, labels) CUDA tensors.

        ``ids_only=True`` (LDOT_OPT_RESULT_SET): for consumers that keep the labels and drop scores and order (hard-negative mining,
        dvl/hn.py:54-63).  ``labels`` hold the same top-k SET; only the candidates whose bf16 score lies within the verify bound of
        the k-th are re-scored from the fp32 rows.  Order: the certain ones by bf16 candidate score (that IS their reported
        score), then the boundary's winners by exact score."""
        import torch
        keep, ptr, nq, dt, mem = _describe(queries, self.d, self.device)
        if mem != L.DEVICE:
            raise ValueError('search_tensors expects a CUDA tensor')
        scores = torch.empty((nq, k), dtype=torch.float32, device=keep.device)
        labels = torch.empty((nq, k), dtype=torch.int64, device=keep.device)
        was = bool(self._opts.get(L.OPT_RESULT_SET, 0))
        if was != bool(ids_only):
            self.set_option(L.OPT_RESULT_SET, int(bool(ids_only)))
        try:
            L.check(self._lib.ldot_index_search(self._h, ptr, nq, dt, mem, int(self.normalize), int(k),
                                                ctypes.c_void_p(scores.data_ptr()), ctypes.c_void_p(labels.data_ptr()),
                                                L.DEVICE, _stream_ptr(self.device)))
        finally:
            if was != bool(ids_only):
                self.set_option(L.OPT_RESULT_SET, int(was))
        return scores, labels

    def last_set_stats(self):
        """(candidates re-scored exactly, live candidates) of the last ``ids_only`` search (drains the device)."""
        a = (ctypes.c_int64 * 2)()
        L.check(self._lib.ldot_index_last_set_stats(self._h, a))
        return dict(rescored=int(a[0]), candidates=int(a[1]))

    def search_begin(self, queries, k: int):
        """First half of a sharded search (CUDA tensors): generates this shard's candidates and returns their thresholds,
        a float32 CUDA tensor [nq] (the k'-th best candidate score per query, -inf while fewer than k' candidates)."""
        import torch
        keep, ptr, nq, dt, mem = _describe(queries, self.d, self.device)
        if mem != L.DEVICE:
            raise ValueError('search_begin expects a CUDA tensor')
        tau = torch.empty((nq,), dtype=torch.float32, device=keep.device)
        L.check(self._lib.ldot_index_search_begin(self._h, ptr, nq, dt, mem, int(self.normalize), int(k),
                                                  ctypes.c_void_p(tau.data_ptr()), _stream_ptr(self.device)))
        self._pending = (nq, int(k), keep.device)
        return tau

    def search_begin_shard(self, queries, k: int, parts: int, total_rows: int = 0, share: float = None):
        """First half of a sharded search for ONE of ``parts`` shards (CUDA tensors): the candidate pass + the statistics the ranks
        all-reduce with MAX, a float32 CUDA tensor [3, nq] (ldot.h: the k'-th best, minus the ceil(k'/parts)-th best, the level above
        which this shard's list is complete).  ``total_rows`` > 0 = the rows of all shards: large batches scan on statistics pooled
        over the whole index (fewer admitted records, one launch); ``shard_floor`` then tells whether that was safe.  ``share`` = the
        part of the k' rows this shard vouches for (default 1/parts; the shares of all shards must add up to at least 1)."""
        import torch
        keep, ptr, nq, dt, mem = _describe(queries, self.d, self.device)
        if mem != L.DEVICE:
            raise ValueError('search_begin_shard expects a CUDA tensor')
        stat = torch.empty((3, nq), dtype=torch.float32, device=keep.device)
        share = 1.0 / parts if share is None else float(share)
        L.check(self._lib.ldot_index_search_begin_shard(self._h, ptr, nq, dt, mem, int(self.normalize), int(k), int(parts), share,
                                                        int(total_rows), ctypes.c_void_p(stat.data_ptr()), _stream_ptr(self.device)))
        self._pending = (nq, int(k), keep.device)
        return stat

    def shard_floor(self, stat):
        """All-reduced (MAX) statistics of ``search_begin_shard`` -> (floor [nq] float32, count [nq] int32, k'): the floor for
        ``search_finish`` and this shard's number of list entries at or above the largest level of any shard — the ranks add the counts
        up (all-reduce SUM); a query whose sum is below k' is unproven and every rank searches again with total_rows = 0."""
        import torch
        if self._pending is None:
            raise L.LdotError(-5, 'shard_floor without a pending search_begin_shard')
        nq, _, dev = self._pending
        stat = stat.contiguous()
        assert stat.shape == (3, nq) and stat.dtype == torch.float32 and stat.is_cuda
        floor = torch.empty((nq,), dtype=torch.float32, device=dev)
        count = torch.empty((nq,), dtype=torch.int32, device=dev)
        kp = ctypes.c_int(0)
        L.check(self._lib.ldot_index_shard_floor(self._h, ctypes.c_void_p(stat.data_ptr()), ctypes.c_void_p(floor.data_ptr()),
                                                 ctypes.c_void_p(count.data_ptr()), ctypes.byref(kp), _stream_ptr(self.device)))
        return floor, count, int(kp.value)

    def search_warmup(self, queries, k: int, parts: int):
        """First step of a sharded search over ``parts`` shards (CUDA tensors): ingests the queries, warms up on this shard and returns
        the statistics the shards exchange — a float32 CUDA tensor [2, nq] to be all-reduced with MAX (ldot.h: row 0 = the k'-th best
        warm-up score, row 1 = minus the ceil(k'/parts)-th best; neutral values when this shard takes a one-pass path)."""
        import torch
        keep, ptr, nq, dt, mem = _describe(queries, self.d, self.device)
        if mem != L.DEVICE:
            raise ValueError('search_warmup expects a CUDA tensor')
        stat = torch.empty((2, nq), dtype=torch.float32, device=keep.device)
        L.check(self._lib.ldot_index_search_warmup(self._h, ptr, nq, dt, mem, int(self.normalize), int(k), int(parts),
                                                   ctypes.c_void_p(stat.data_ptr()), _stream_ptr(self.device)))
        self._pending = (nq, int(k), keep.device)
        return stat

    def search_scan(self, stat=None):
        """Second step: the candidate pass, from the thresholds the shards agreed on (``stat`` = the all-reduced tensor of
        search_warmup; None = this shard's own).  Returns the thresholds like search_begin does."""
        import torch
        if self._pending is None:
            raise L.LdotError(-5, 'search_scan without a pending search_warmup')
        nq, _, dev = self._pending
        tau = torch.empty((nq,), dtype=torch.float32, device=dev)
        sptr = ctypes.c_void_p(0)
        if stat is not None:
            stat = stat.to(device=dev, dtype=torch.float32).contiguous()
            assert stat.shape == (2, nq)
            sptr = ctypes.c_void_p(stat.data_ptr())
        L.check(self._lib.ldot_index_search_scan(self._h, sptr, ctypes.c_void_p(tau.data_ptr()), _stream_ptr(self.device)))
        return tau

    def search_finish(self, floor=None):
        """Second half: re-scores the candidates at or above ``floor`` ([nq] float32 CUDA tensor, e.g. the all-reduce MAX of
        the shards' thresholds; None = all) and returns this shard's partial top-k (scores, labels) as CUDA tensors."""
        import torch
        if self._pending is None:
            raise L.LdotError(-5, 'search_finish without a pending search_begin')
        (nq, k, dev), self._pending = self._pending, None
        scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
        labels = torch.empty((nq, k), dtype=torch.int64, device=dev)
        fptr = ctypes.c_void_p(0)
        if floor is not None:
            floor = floor.to(device=dev, dtype=torch.float32).contiguous()
            assert floor.shape == (nq,)
            fptr = ctypes.c_void_p(floor.data_ptr())
        L.check(self._lib.ldot_index_search_finish(self._h, fptr, ctypes.c_void_p(scores.data_ptr()),
                                                   ctypes.c_void_p(labels.data_ptr()), L.DEVICE, _stream_ptr(self.device)))
        return scores, labels

    def search_finish_blocked(self, floor, blocks, block_rows: int, block_bytes: int, label_base: int):
        """search_finish into the send buffer of a sharded search's all-to-all (``blocks``: a uint8 CUDA tensor of
        ceil(nq / block_rows) * block_bytes bytes, layout in ldot.h); labels come out global (+ label_base)."""
        if self._pending is None:
            raise L.LdotError(-5, 'search_finish without a pending search_begin')
        (nq, k, dev), self._pending = self._pending, None
        assert blocks.is_cuda and blocks.is_contiguous() and blocks.numel() * blocks.element_size() >= -(-nq // block_rows) * block_bytes
        fptr = ctypes.c_void_p(0)
        if floor is not None:
            floor = floor.to(device=dev, dtype=floor.dtype).contiguous()
            assert floor.shape == (nq,) and floor.dtype.is_floating_point and floor.element_size() == 4
            fptr = ctypes.c_void_p(floor.data_ptr())
        L.check(self._lib.ldot_index_search_finish_blocked(self._h, fptr, ctypes.c_void_p(blocks.data_ptr()), int(block_rows),
                                                           int(block_bytes), int(label_base), _stream_ptr(self.device)))

    def get_rows(self, row0: int, n: int) -> np.ndarray:
        out = np.empty((n, self.d), dtype=np.float32)
        L.check(self._lib.ldot_index_get_rows(self._h, int(row0), int(n), ctypes.c_void_p(out.ctypes.data), L.HOST,
                                              _stream_ptr(self.device)))
        return out

    def last_stats(self):
        a = (ctypes.c_int64 * 4)()
        L.check(self._lib.ldot_index_last_stats(self._h, a))
        return dict(fused_candidates=a[0], overflowed_queries=a[1], dense_pairs=a[2], fused_pairs=a[3])

    _PATHS = ('none', 'narrow', 'dense', 'fused_one_block', 'fused')
    _THRESHOLDS = ('n/a', 'guaranteed', 'optimistic', 'pooled')
    _ORDERS = ('n/a', 'storage', 'scrambled_tiles')
    _ROWS = ('as_added', 'shuffled_at_add', 'reshuffled_by_library')

    def last_regime(self):
        """Which regime the last search ran in + the adaptive state of the handle (ldot_index_last_regime)."""
        a = (ctypes.c_int64 * 8)()
        L.check(self._lib.ldot_index_last_regime(self._h, a))
        return dict(path=self._PATHS[a[0]], thresholds=self._THRESHOLDS[a[1]], scan_order=self._ORDERS[a[2]], redone_queries=a[3],
                    guaranteed_searches_left=a[4], narrow_skips_left=a[5], scrambled_auto=bool(a[6]), rows=self._ROWS[a[7]])

    def last_profile(self):
        a = (ctypes.c_double * 4)()
        L.check(self._lib.ldot_index_last_profile(self._h, a))
        return dict(launches=a[0], kernel_ms=a[1], flops=a[2], bytes=a[3])

    def save(self, path: str):
        L.check(self._lib.ldot_index_save(self._h, path.encode()))

    @classmethod
    def load(cls, path: str, normalize: bool = False):
        lib = L.load_library()
        h = ctypes.c_void_p()
        L.check(lib.ldot_index_load(path.encode(), ctypes.byref(h)))
        return cls(0, normalize=normalize, _handle=h)


class DenseIndexer(object):
    """faiss_indexers.py:22-60"""

    def __init__(self, buffer_size: int = 50000):
        self.buffer_size = buffer_size
        self.index_id_to_db_id = []
        self.index = None

    def index_data(self, data: List[Tuple[object, np.array]]):
        raise NotImplementedError

    def search_knn(self, query_vectors: np.array, top_docs: int) -> List[Tuple[List[object], List[float]]]:
        raise NotImplementedError

    def search_knn_tensors(self, query_vectors, top_docs: int):
        """Device-tensor variant of ``search_knn`` (what the harness, the mining and the serving path call): -> (scores [nq, k],
        row labels [nq, k] int64, -1 = padding) as CUDA tensors, scores with the semantics of the class's ``search_knn``; labels
        index ``index_id_to_db_id``."""
        raise NotImplementedError

    def serialize(self, file: str):
        """Two files like the reference (:35-43): ``file + '.index.dpr'`` (own LDOTIDX1 binary instead of the
        third-party faiss format) and ``file + '.index_meta.dpr'`` (pickled id list)."""
        logger.info('Serializing index to %s', file)
        index_file = file + '.index.dpr'
        meta_file = file + '.index_meta.dpr'
        self.index.save(index_file)
        with open(meta_file, mode='wb') as f:
            pickle.dump(self.index_id_to_db_id, f)

    def deserialize_from(self, file: str):
        logger.info('Loading index from %s', file)
        index_file = file + '.index.dpr'
        meta_file = file + '.index_meta.dpr'
        self.index = FlatIPIndex.load(index_file, normalize=getattr(self.index, 'normalize', False))
        logger.info('Loaded index of type %s and size %d', type(self.index), self.index.ntotal)
        with open(meta_file, 'rb') as reader:
            self.index_id_to_db_id = pickle.load(reader)
        assert len(
            self.index_id_to_db_id) == self.index.ntotal, 'Deserialized index_id_to_db_id should match faiss index size'

    def _update_id_mapping(self, db_ids: List):
        self.index_id_to_db_id.extend(db_ids)


class DenseFlatIndexer(DenseIndexer):
    """faiss_indexers.py:63-87"""

    def __init__(self, vector_sz: int, buffer_size: int = 50000, normalize: bool = False, shuffle_seed: Optional[int] = None):
        """``shuffle_seed`` (not in the reference): the rows of every index_data / index_tensor call are stored in a pseudo-random
        order, together with their ids — invisible from outside (search_knn reports external ids; only the order of EQUAL scores
        changes) and free, since the id list is kept next to the rows anyway.  For data that arrives sorted in short runs of similar
        rows (by class, by k-means list): the scan's thresholds and candidate pools assume that the rows seen so far are a fair sample
        of the index, long runs are handled by the library's scrambled tile order (LDOT_OPT_SCAN_ORDER), runs about as long as a
        384-row tile only by a row-granular order like this one."""
        super(DenseFlatIndexer, self).__init__(buffer_size=buffer_size)
        self.index = FlatIPIndex(vector_sz, normalize=normalize)
        self.shuffle_seed = shuffle_seed

    def _permutation(self, n: int):
        return np.random.default_rng([int(self.shuffle_seed), len(self.index_id_to_db_id)]).permutation(n)

    def index_data(self, data: List[Tuple[object, np.array]]):
        n = len(data)
        if self.shuffle_seed is not None and n > 1:
            data = [data[i] for i in self._permutation(n)]
        # same chunking as the reference (:72-77); vectors may be numpy rows or (device) tensors
        for i in range(0, n, self.buffer_size):
            chunk = data[i:i + self.buffer_size]
            db_ids = [t[0] for t in chunk]
            if chunk and _is_tensor(chunk[0][1]):
                import torch
                vectors = torch.stack([t[1].reshape(-1) for t in chunk], dim=0)
            else:
                vectors = np.concatenate([np.reshape(t[1], (1, -1)) for t in chunk], axis=0)
            self._update_id_mapping(db_ids)
            self.index.add(vectors)
        indexed_cnt = len(self.index_id_to_db_id)
        logger.info('Total data indexed %d', indexed_cnt)

    def index_tensor(self, db_ids: List, vectors):
        """Device path: ids + one [n, d] tensor, no per-row Python objects."""
        if len(db_ids) != vectors.shape[0]:
            raise ValueError('ids / vectors length mismatch')
        db_ids = list(db_ids)
        if self.shuffle_seed is not None and len(db_ids) > 1:
            perm = self._permutation(len(db_ids))
            db_ids = [db_ids[i] for i in perm]
            if _is_tensor(vectors):
                import torch
                vectors = vectors[torch.as_tensor(perm, device=vectors.device)]
            else:
                vectors = np.asarray(vectors)[perm]
        self._update_id_mapping(db_ids)
        self.index.add(vectors)

    def search_knn(self, query_vectors: np.array, top_docs: int) -> List[Tuple[List[object], List[float]]]:
        scores, indexes = self.index.search(query_vectors, top_docs)
        # convert to external ids exactly like :85 — a padding label (-1, fewer than top_docs rows indexed) maps to
        # the LAST id through Python's negative indexing, the reference's observable behaviour
        ids = self.index_id_to_db_id
        db_ids = [[ids[i] for i in query_top_idxs] for query_top_idxs in indexes.tolist()]
        result = [(db_ids[i], scores[i]) for i in range(len(db_ids))]
        return result

    def search_knn_tensors(self, query_vectors, top_docs: int, ids_only: bool = False):
        """(scores [nq, k], row labels [nq, k]) as device tensors; map labels with ``index_id_to_db_id``.  ``ids_only``: the top-k SET
        for consumers that drop scores and order (FlatIPIndex.search_tensors)."""
        return self.index.search_tensors(query_vectors, top_docs, ids_only=ids_only)


class DenseHNSWFlatIndexer(DenseIndexer):
    """faiss_indexers.py:90-154 — the reference's ``--hnsw_index`` alternative: faiss IndexHNSWFlat over vectors augmented
    with one extra dimension sqrt(phi - |x|^2) (phi = max |x|^2), queries augmented with 0, so that L2 order = inner
    product order; ``search_knn`` returns the SQUARED L2 distances of the augmented vectors, ascending.

    Default: the graph is not needed — the exact scan of a 1M x 768 index takes 0.5 ms per query on an MI355X — so the
    class keeps the reference's surface (constructor arguments, all-data-at-once rule, phi bookkeeping, score semantics,
    id mapping) on top of the exact flat index.  Neighbours are the exact ones (recall 1.0 instead of HNSW's ~0.99);
    distances are |q|^2 + phi - 2 q.x from the exact fp32 inner products.

    ``approximate=True``: a real approximate index behind the same surface — the inverted-file index of ``ivf.py`` (the machine's
    counterpart of the graph: ``ef_search // 4`` lists are probed per query); same phi augmentation, same distances, recall < 1."""

    def __init__(self, vector_sz: int, buffer_size: int = 50000, store_n: int = 512, ef_search: int = 128,
                 ef_construction: int = 200, approximate: bool = False):
        super(DenseHNSWFlatIndexer, self).__init__(buffer_size=buffer_size)
        self.index = FlatIPIndex(vector_sz)
        self.store_n, self.ef_search, self.ef_construction = store_n, ef_search, ef_construction   # store_n / ef_construction: unused
        self._ivf = None
        if approximate:
            from .ivf import DenseIVFFlatIndexer
            self._ivf = DenseIVFFlatIndexer(vector_sz, buffer_size, nprobe=max(1, ef_search // 4))
            self.index = self._ivf.index
        self.phi = 0               # the reference's re-index guard (:106,112-113,154)
        self._phi_value = 0.0      # max squared row norm of the indexed data

    def _check_first(self):
        if self.phi > 0:
            raise RuntimeError('DPR HNSWF index needs to index all data at once,'
                               'results will be unpredictable otherwise.')

    def index_data(self, data: List[Tuple[object, np.array]]):
        self._check_first()
        n = len(data)
        phi = 0
        for _, doc_vector in data:                     # :114-118 (float32 arithmetic like the reference)
            v = doc_vector.detach().cpu().numpy() if _is_tensor(doc_vector) else np.asarray(doc_vector)
            phi = max(phi, (v ** 2).sum())
        logger.info('HNSWF DotProduct -> L2 space phi={}'.format(phi))
        self.phi = 0                                   # (:119: the reference resets its guard here)
        self._phi_value = max(float(phi), self._phi_value)
        if self._ivf is not None:
            self._ivf.index_data(data)
            self.index_id_to_db_id = self._ivf.index_id_to_db_id
            return
        for i in range(0, n, self.buffer_size):
            chunk = data[i:i + self.buffer_size]
            if chunk and _is_tensor(chunk[0][1]):
                import torch
                vectors = torch.stack([t[1].reshape(-1) for t in chunk], dim=0)
            else:
                vectors = np.concatenate([np.reshape(t[1], (1, -1)) for t in chunk], axis=0)
            self._update_id_mapping([t[0] for t in chunk])
            self.index.add(vectors)
            logger.info('data indexed %d', len(self.index_id_to_db_id))
        logger.info('Total data indexed %d', len(self.index_id_to_db_id))

    def index_tensor(self, db_ids: List, vectors):
        """Device path: ids + one [n, d] tensor."""
        self._check_first()
        if len(db_ids) != vectors.shape[0]:
            raise ValueError('ids / vectors length mismatch')
        if vectors.shape[0]:
            self._phi_value = max(self._phi_value, float((vectors.float() ** 2).sum(dim=1).max().item()))
        if self._ivf is not None:
            self._ivf.index_tensor(db_ids, vectors)
            self.index_id_to_db_id = self._ivf.index_id_to_db_id
            return
        self._update_id_mapping(list(db_ids))
        self.index.add(vectors)

    def search_knn_tensors(self, query_vectors, top_docs: int, ids_only: bool = False):
        """(squared L2 distances of the augmented vectors [nq, k] ascending, row labels [nq, k]) as CUDA tensors: the neighbours
        of ``search_knn`` without the per-result Python objects.  Padding: label -1, distance FLT_MAX (what faiss pads with).
        ``ids_only`` (exact-backed index only): the neighbour SET, order and distances approximate (FlatIPIndex.search_tensors)."""
        import torch
        if _is_tensor(query_vectors):
            qt = query_vectors.detach()
        else:
            qt = torch.from_numpy(np.ascontiguousarray(query_vectors, dtype=np.float32))
        if qt.dim() == 1:
            qt = qt[None]
        if not qt.is_cuda:
            qt = qt.cuda(self.index.device)
        if self._ivf is not None:
            ip, labels = self._ivf.search_knn_tensors(qt, top_docs)
        else:
            ip, labels = self.index.search_tensors(qt, top_docs, ids_only=ids_only)
        qn = (qt.float() ** 2).sum(dim=1)                                  # fp32 like the reference's numpy arithmetic
        dist = qn[:, None] + torch.tensor(self._phi_value, dtype=torch.float32, device=ip.device) - 2.0 * ip
        dist = torch.where(labels >= 0, dist, dist.new_full((), 3.4028234663852886e38))
        return dist, labels

    def search_knn(self, query_vectors: np.array, top_docs: int) -> List[Tuple[List[object], List[float]]]:
        dist, labels = self.search_knn_tensors(query_vectors, top_docs)
        scores, indexes = dist.cpu().numpy(), labels.cpu().numpy()
        ids = self.index_id_to_db_id
        db_ids = [[ids[i] for i in query_top_idxs] for query_top_idxs in indexes.tolist()]
        return [(db_ids[i], scores[i]) for i in range(len(db_ids))]

    def serialize(self, file: str):
        """Exact-backed: the reference's two files.  ``approximate=True``: the inverted file's own two files (rows sorted by list +
        a meta file that also carries the list offsets, the centroids, phi and nprobe — without them the rows are useless)."""
        if self._ivf is not None:
            return self._ivf.serialize(file)
        return super(DenseHNSWFlatIndexer, self).serialize(file)

    def deserialize_from(self, file: str):
        if self._ivf is not None:
            self._ivf.deserialize_from(file)
            self.index = self._ivf.index
            self.index_id_to_db_id = self._ivf.index_id_to_db_id
        else:
            super(DenseHNSWFlatIndexer, self).deserialize_from(file)
        # to trigger the error on subsequent indexing (:152-154)
        self.phi = 1
        phi = 0.0
        for r0 in range(0, self.index.ntotal, 65536):
            rows = self.index.get_rows(r0, min(65536, self.index.ntotal - r0))
            phi = max(phi, float((rows ** 2).sum(axis=1).max()))
        self._phi_value = phi
