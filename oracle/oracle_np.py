"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A plain numpy (fp32, with fp64 helpers) restatement of the reference's retrieval hot path
(intersun/LightningDOT).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package ``lightningdot_amd`` never does.

Pinning status (see DESIGN.md "Oracle"):
  * loss / train-step composition / recall harness / hard-negative post-processing / pooling+projection /
    indexer wrapper logic: PINNED against outputs of the reference itself, generated in the build
    container by ``oracle/gen_golden.py`` (reference imported with sys.modules stubs) and committed
    under ``tests/golden/``.
  * ``FlatIP`` (the arithmetic of ``faiss.IndexFlatIP`` from faiss-cpu==1.6.3, DVL.yml:80): the
    library is third-party, un-vendored and not installable here -> **parity unpinned** for that
    single boundary; the restatement follows its documented semantics (exact fp32 inner product,
    k best in descending order, label -1 / score -FLT_MAX padding when k > ntotal) and the call
    sites dvl/indexer/faiss_indexers.py:67,77,83.

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
import random as _random
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

NEG_FLT_MAX = -np.finfo(np.float32).max


# --------------------------------------------------------------------------------------
# scoring + loss  (dvl/models/bi_encoder.py:54-68, 615-656; dvl/utils.py:114-169)
# --------------------------------------------------------------------------------------
def dot_product_scores(q: np.ndarray, ctx: np.ndarray, cosine: bool = False) -> np.ndarray:
    """bi_encoder.py:54-68 — r = q @ ctx.T ; the cosine branch (never taken by the reference,
    SURVEY F2) divides by the outer product of the row norms."""
    q = np.asarray(q, dtype=np.float32)
    ctx = np.asarray(ctx, dtype=np.float32)
    r = q @ ctx.T
    if cosine:
        n1 = np.linalg.norm(q, axis=-1)
        n2 = np.linalg.norm(ctx, axis=-1)
        return r / np.outer(n1, n2)
    return r


def _log_softmax(x: np.ndarray) -> np.ndarray:
    m = x.max(axis=1, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(axis=1, keepdims=True))


def biencoder_nll_loss(q, ctx, cap, positive_idx: Sequence[int], caption_score_weight: float = 0.1,
                       reduction: str = 'mean', dtype=np.float32):
    """BiEncoderNllLoss.calc — bi_encoder.py:615-656.

    scores = (1-w)*S_img + w*S_cap if captions given and w != 0 (:624-629); log_softmax(dim=1) (:649);
    nll_loss(reduction) (:651); correct = sum(argmax == positive) (:654-655).
    Returns (loss, correct_count, scores).  ``dtype=np.float64`` gives the high-precision variant
    used for tolerance budgeting."""
    q = np.asarray(q, dtype=dtype)
    ctx = np.asarray(ctx, dtype=dtype)
    s_img = q @ ctx.T
    if cap is not None and caption_score_weight != 0:
        s_cap = q @ np.asarray(cap, dtype=dtype).T
        scores = (1 - caption_score_weight) * s_img + caption_score_weight * s_cap
        scores = scores.astype(dtype)
    else:
        scores = s_img
    lsm = _log_softmax(scores)
    pos = np.asarray(positive_idx, dtype=np.int64)
    picked = -lsm[np.arange(len(pos)), pos]
    if reduction == 'mean':
        loss = picked.mean(dtype=dtype)
    elif reduction == 'sum':
        loss = picked.sum(dtype=dtype)
    elif reduction == 'none':
        loss = picked
    else:
        raise ValueError(reduction)
    # torch.max returns the FIRST maximal index on ties, as does np.argmax
    correct = int((lsm.argmax(axis=1) == pos).sum())
    return loss, correct, scores


def biencoder_nll_grads(q, ctx, cap, positive_idx, caption_score_weight=0.1, reduction='mean',
                        grad_loss=None, grad_scores=None, dtype=np.float64):
    """Analytic gradients of ``biencoder_nll_loss`` (what torch autograd computes for
    bi_encoder.py:615-656): dS = (softmax - onehot) * g_row (+ grad_scores), dq = dS_img@ctx + dS_cap@cap ..."""
    q = np.asarray(q, dtype=dtype)
    ctx = np.asarray(ctx, dtype=dtype)
    use_cap = cap is not None and caption_score_weight != 0
    w = caption_score_weight if use_cap else 0.0
    s = q @ ctx.T
    if use_cap:
        cap = np.asarray(cap, dtype=dtype)
        s = (1 - w) * s + w * (q @ cap.T)
    n1 = s.shape[0]
    p = np.exp(_log_softmax(s))
    pos = np.asarray(positive_idx, dtype=np.int64)
    p[np.arange(n1), pos] -= 1.0
    if reduction == 'mean':
        g = np.full((n1,), (1.0 if grad_loss is None else float(grad_loss)) / n1, dtype=dtype)
    elif reduction == 'sum':
        g = np.full((n1,), 1.0 if grad_loss is None else float(grad_loss), dtype=dtype)
    else:
        g = np.ones((n1,), dtype=dtype) if grad_loss is None else np.asarray(grad_loss, dtype=dtype)
    ds = p * g[:, None]
    if grad_scores is not None:
        ds = ds + np.asarray(grad_scores, dtype=dtype)
    dq = (1 - w) * (ds @ ctx)
    dctx = (1 - w) * (ds.T @ q)
    dcap = None
    if use_cap:
        dq = dq + w * (ds @ cap)
        dcap = w * (ds.T @ q)
    return dq, dctx, dcap


def calc_loss(caption_score_weight, q, ctx, cap, positive_idx, hard_negative_idx=None, reduction='mean'):
    """dvl/utils.py:114-169 — world-size-1 pass-through to BiEncoderNllLoss.calc with
    args.caption_score_weight (:158-167); the DDP branch (:121-156) is dead code."""
    return biencoder_nll_loss(q, ctx, cap, positive_idx, caption_score_weight, reduction)


def train_step_loss(txt, img, cap, bs: int, num_hard_negatives: int, pos_ctx_indices,
                    caption_score_weight: float = 0.0):
    """train_itm.py:195-222 — two _calc_loss calls (img->txt with img[:bs] as queries, txt->img with
    txt[:bs]) when num_hard_negatives > 0, un-sliced otherwise; averaged."""
    txt = np.asarray(txt, np.float32)
    img = np.asarray(img, np.float32)
    if num_hard_negatives > 0:
        l_t, c_t, s_t = calc_loss(caption_score_weight, img[:bs], txt, cap, pos_ctx_indices)
        l_i, c_i, s_i = calc_loss(caption_score_weight, txt[:bs], img, cap, pos_ctx_indices)
    else:
        l_t, c_t, s_t = calc_loss(caption_score_weight, img, txt, cap, pos_ctx_indices)
        l_i, c_i, s_i = calc_loss(caption_score_weight, txt, img, cap, pos_ctx_indices)
    is_correct = (c_t + c_i) / 2
    loss = np.float32(0.5) * l_t + np.float32(0.5) * l_i
    scores = s_t * np.float32(0.5) + s_i * np.float32(0.5)
    return loss, is_correct, scores, (l_t, l_i)


# --------------------------------------------------------------------------------------
# pooling + projection  (dvl/models/bi_encoder.py:82-88,120-122,137-143,188-190)
# --------------------------------------------------------------------------------------
def cls_pool(sequence_output: np.ndarray, l2_normalize: bool = False, eps: float = 1e-12) -> np.ndarray:
    """bi_encoder.py:120 / :188 — pooled = sequence_output[:, 0, :].  No L2 normalisation in the
    reference; ``l2_normalize`` is the opt-in north_star variant (x / max(||x||, eps))."""
    x = np.asarray(sequence_output)[:, 0, :].astype(np.float32)
    if l2_normalize:
        n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
        x = (x / np.maximum(n, eps)).astype(np.float32)
    return x


def _gelu_erf(x):
    # uniter_model/model/layer.py:31-37 — x * 0.5 * (1 + erf(x / sqrt(2)))
    from scipy.special import erf
    return x * 0.5 * (1.0 + erf(x / math.sqrt(2.0)))


def encode_proj(pooled, w0, b0, ln_g, ln_b, w3, b3, eps: float = 1e-12):
    """bi_encoder.py:82-88 — Linear(768->1536) -> erf-GELU -> LayerNorm(1536, eps 1e-12) -> Linear(1536->D)."""
    x = np.asarray(pooled, np.float64)
    h = x @ np.asarray(w0, np.float64).T + b0
    h = _gelu_erf(h)
    mu = h.mean(axis=-1, keepdims=True)
    var = ((h - mu) ** 2).mean(axis=-1, keepdims=True)
    h = (h - mu) / np.sqrt(var + eps) * ln_g + ln_b
    return (h @ np.asarray(w3, np.float64).T + b3).astype(np.float32)


# --------------------------------------------------------------------------------------
# flat inner-product index  (faiss.IndexFlatIP semantics; call sites faiss_indexers.py:67,77,83)
# --------------------------------------------------------------------------------------
class FlatIP:
    """Exact fp32 inner-product index.  PARITY UNPINNED vs faiss-cpu==1.6.3 (not installable here).
    Tie policy (faiss' heap order for exact ties is unspecified): lowest index wins."""

    def __init__(self, d: int):
        self.d = int(d)
        self._chunks: List[np.ndarray] = []
        self._x: Optional[np.ndarray] = None

    @property
    def ntotal(self) -> int:
        return sum(c.shape[0] for c in self._chunks)

    def add(self, v: np.ndarray):
        v = np.ascontiguousarray(v, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] != self.d:
            raise ValueError('dimension mismatch')
        self._chunks.append(v)
        self._x = None

    @property
    def x(self) -> np.ndarray:
        if self._x is None:
            self._x = (np.concatenate(self._chunks, axis=0) if self._chunks
                       else np.zeros((0, self.d), np.float32))
            self._chunks = [self._x] if self._x.shape[0] else []
        return self._x

    def search(self, q: np.ndarray, k: int, q_block: int = 1024, n_block: int = 65536):
        """Blocked sgemm + per-row selection, the same structure as faiss' flat search
        (blocked BLAS sgemm + heap).  Returns (scores[nq,k] fp32 desc, labels[nq,k] int64)."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        x = self.x
        nq, n = q.shape[0], x.shape[0]
        out_s = np.full((nq, k), NEG_FLT_MAX, dtype=np.float32)
        out_i = np.full((nq, k), -1, dtype=np.int64)
        if nq == 0 or n == 0 or k == 0:
            return out_s, out_i
        kk = min(k, n)
        for q0 in range(0, nq, q_block):
            qb = q[q0:q0 + q_block]
            best_s = np.empty((qb.shape[0], 0), np.float32)
            best_i = np.empty((qb.shape[0], 0), np.int64)
            for n0 in range(0, n, n_block):
                s = qb @ x[n0:n0 + n_block].T
                idx = np.broadcast_to(np.arange(n0, n0 + s.shape[1], dtype=np.int64), s.shape)
                cs = np.concatenate([best_s, s], axis=1)
                ci = np.concatenate([best_i, idx], axis=1)
                if cs.shape[1] > kk:
                    # keep the kk best under (score desc, index asc): lexsort is exact but slow,
                    # so partition on score first with a safety band for ties at the boundary.
                    part = np.argpartition(-cs, kk - 1, axis=1)[:, :kk]
                    thr = np.take_along_axis(cs, part, axis=1).min(axis=1, keepdims=True)
                    keep_s, keep_i = [], []
                    for r in range(cs.shape[0]):
                        m = cs[r] >= thr[r, 0]
                        rs, ri = cs[r][m], ci[r][m]
                        o = np.lexsort((ri, -rs))[:kk]
                        keep_s.append(rs[o])
                        keep_i.append(ri[o])
                    best_s = np.stack(keep_s)
                    best_i = np.stack(keep_i)
                else:
                    best_s, best_i = cs, ci
            order = np.lexsort((best_i, -best_s), axis=1) if best_s.shape[1] else None
            if order is not None:
                bs_ = np.take_along_axis(best_s, order, axis=1)[:, :kk]
                bi_ = np.take_along_axis(best_i, order, axis=1)[:, :kk]
                out_s[q0:q0 + qb.shape[0], :kk] = bs_
                out_i[q0:q0 + qb.shape[0], :kk] = bi_
        return out_s, out_i


def search_fast(q: np.ndarray, x: np.ndarray, k: int, q_block: int = 2048, n_block: int = 131072):
    """Throughput-oriented variant of ``FlatIP.search`` for the cpu_baseline timing leg: blocked fp32
    sgemm + argpartition, final ordering by (score desc).  Same arithmetic; ties not canonicalised."""
    q = np.ascontiguousarray(q, dtype=np.float32)
    nq, n = q.shape[0], x.shape[0]
    kk = min(k, n)
    out_s = np.full((nq, k), NEG_FLT_MAX, dtype=np.float32)
    out_i = np.full((nq, k), -1, dtype=np.int64)
    for q0 in range(0, nq, q_block):
        qb = q[q0:q0 + q_block]
        bs_ = np.empty((qb.shape[0], 0), np.float32)
        bi_ = np.empty((qb.shape[0], 0), np.int64)
        for n0 in range(0, n, n_block):
            s = qb @ x[n0:n0 + n_block].T
            if s.shape[1] > kk:
                part = np.argpartition(-s, kk - 1, axis=1)[:, :kk]
                ps = np.take_along_axis(s, part, axis=1)
                pi = part.astype(np.int64) + n0
            else:
                ps, pi = s, np.broadcast_to(np.arange(n0, n0 + s.shape[1], dtype=np.int64), s.shape)
            bs_ = np.concatenate([bs_, ps], axis=1)
            bi_ = np.concatenate([bi_, pi], axis=1)
            if bs_.shape[1] > 4 * kk:
                part = np.argpartition(-bs_, kk - 1, axis=1)[:, :kk]
                bs_ = np.take_along_axis(bs_, part, axis=1)
                bi_ = np.take_along_axis(bi_, part, axis=1)
        order = np.argsort(-bs_, axis=1, kind='stable')[:, :kk]
        out_s[q0:q0 + qb.shape[0], :kk] = np.take_along_axis(bs_, order, axis=1)
        out_i[q0:q0 + qb.shape[0], :kk] = np.take_along_axis(bi_, order, axis=1)
    return out_s, out_i


def merge_topk(partials: Sequence[Tuple[np.ndarray, np.ndarray]], k: int):
    """Global top-k of the union of per-shard top-k lists: inputs are
    (scores[nq,k_r], global_labels[nq,k_r]) per shard; padding entries have label -1.
    Order: score desc, label asc."""
    s = np.concatenate([p[0] for p in partials], axis=1).astype(np.float32)
    i = np.concatenate([p[1] for p in partials], axis=1).astype(np.int64)
    s = np.where(i < 0, NEG_FLT_MAX, s)
    big = np.where(i < 0, np.iinfo(np.int64).max, i)
    order = np.lexsort((big, -s), axis=1)[:, :k]
    out_s = np.take_along_axis(s, order, axis=1)
    out_i = np.take_along_axis(i, order, axis=1)
    if out_s.shape[1] < k:
        pad = k - out_s.shape[1]
        out_s = np.concatenate([out_s, np.full((s.shape[0], pad), NEG_FLT_MAX, np.float32)], axis=1)
        out_i = np.concatenate([out_i, np.full((s.shape[0], pad), -1, np.int64)], axis=1)
    return out_s, out_i


class DenseFlatIndexerOracle:
    """dvl/indexer/faiss_indexers.py:22-87 — DenseIndexer/DenseFlatIndexer wrapper logic on ``FlatIP``.
    ``search_knn`` maps labels through ``index_id_to_db_id[i]`` exactly like :85 (so a -1 padding label
    becomes the LAST id through Python negative indexing — the reference's latent behaviour)."""

    def __init__(self, vector_sz: int, buffer_size: int = 50000):
        self.buffer_size = buffer_size
        self.index_id_to_db_id: list = []
        self.index = FlatIP(vector_sz)

    def index_data(self, data):
        n = len(data)
        for i in range(0, n, self.buffer_size):           # :72
            db_ids = [t[0] for t in data[i:i + self.buffer_size]]
            vectors = np.concatenate([np.reshape(t[1], (1, -1)) for t in data[i:i + self.buffer_size]], axis=0)
            self.index_id_to_db_id.extend(db_ids)          # :76
            self.index.add(vectors)                        # :77

    def search_knn(self, query_vectors, top_docs: int):
        scores, indexes = self.index.search(query_vectors, top_docs)        # :83
        db_ids = [[self.index_id_to_db_id[i] for i in row] for row in indexes]  # :85
        return [(db_ids[i], scores[i]) for i in range(len(db_ids))]        # :86


# --------------------------------------------------------------------------------------
# eval harness  (dvl/trainer.py:113-190)
# --------------------------------------------------------------------------------------
def eval_on_stream(batches: Iterable[dict], vector_size: int, img2txt: Optional[Dict] = None,
                   num_tops: int = 100, caption_score_weight: float = 0.0, no_eval: bool = False,
                   indexer_cls=DenseFlatIndexerOracle):
    """eval_model_on_dataloader — dvl/trainer.py:113-190, with the towers factored out: each batch is
    a dict with 'txt_index' (list of ids), 'img_fname' (list), 'q' [B,D] (text vectors), 'ctx' [B,D]
    (image vectors), optional 'cap' [B,D], and 'n_txt' (= batch['txts']['input_ids'].shape[0], :148).

    Reproduced quirks: image queries are NOT de-duplicated (:138-139), index sides are de-duplicated by
    dict key with last-write-wins (:151-152), recall denominators are the number of unique query ids
    (:179,188), return order (recall_txt, recall_img) where recall_txt = text-query -> image (:190)."""
    total_loss, total_correct, n_batches, total_samples = 0.0, 0, 0, 0
    labels_img_name, labels_txt_name = [], []
    img_embedding, txt_embedding = {}, {}
    query_txt, query_txt_id, query_img, query_img_id = [], [], [], []
    for b in batches:
        q = np.asarray(b['q'], np.float32)
        c = np.asarray(b['ctx'], np.float32)
        cap = b.get('cap')
        query_txt.extend([v.reshape(-1) for v in q])
        query_txt_id.extend(b['txt_index'])
        query_img.extend([v.reshape(-1) for v in c])
        query_img_id.extend(b['img_fname'])
        loss, correct, _ = calc_loss(caption_score_weight, q, c, cap, list(range(len(q))))
        total_loss += float(loss)
        total_correct += int(correct)
        n_batches += 1
        total_samples += int(b.get('n_txt', len(q)))
        img_embedding.update({k: v for k, v in zip(b['img_fname'], c)})
        txt_embedding.update({k: v for k, v in zip(b['txt_index'], q)})
        labels_img_name.extend(b['img_fname'])
        labels_txt_name.extend(b['txt_index'])
    total_loss = total_loss / n_batches
    correct_ratio = total_correct / float(total_samples)
    indexer_img = indexer_cls(vector_size)
    indexer_txt = indexer_cls(vector_size)
    query_txt_np = np.array(query_txt)
    indexer_img.index_data(list(img_embedding.items()))
    query_img_np = np.array(query_img)
    indexer_txt.index_data(list(txt_embedding.items()))
    if no_eval:
        return total_loss, correct_ratio, (indexer_img, indexer_txt), (None, None), (None, None)
    res_txt = indexer_img.search_knn(query_txt_np, num_tops)
    rank_txt_res = {query_txt_id[i]: r[0] for i, r in enumerate(res_txt)}
    res_img = indexer_txt.search_knn(query_img_np, num_tops)
    rank_img_res = {query_img_id[i]: r[0] for i, r in enumerate(res_img)}
    recall_txt = {1: 0, 5: 0, 10: 0}
    for i, q in enumerate(query_txt_id):
        for top in recall_txt:
            recall_txt[top] += labels_img_name[i] in rank_txt_res[q][:top]
    for top in recall_txt:
        recall_txt[top] = recall_txt[top] / len(rank_txt_res)
    recall_img = {1: 0, 5: 0, 10: 0}
    for q in np.unique(query_img_id):
        for top in recall_img:
            recall_img[top] += any(t in rank_img_res[q][:top] for t in img2txt[q])
    for top in recall_img:
        recall_img[top] = recall_img[top] / len(rank_img_res)
    return total_loss, correct_ratio, (indexer_img, indexer_txt), (recall_txt, recall_img), \
        (rank_txt_res, rank_img_res)


# --------------------------------------------------------------------------------------
# hard-negative post-processing  (dvl/hn.py:45-66)
# --------------------------------------------------------------------------------------
def num_hard_sampled(num_hard_negatives: int) -> int:
    """dvl/hn.py:53"""
    return min(max(num_hard_negatives * 2 + 10, 50), 1000)


def hard_negative_postprocess(hard_neg_img: Dict, hard_neg_txt: Dict, train_txt2img: Dict,
                              train_img2txt: Dict, num_hard_negatives: int, rng: Optional[_random.Random] = None,
                              sample=None):
    """dvl/hn.py:57-63.  ``hard_neg_img`` = {txt_id: [img ids]} (rank_txt_res), ``hard_neg_txt`` =
    {img_id: [txt ids]} (rank_img_res).  :57 removes the positive image from each text's list in place
    (first occurrence); :58 replaces each image's list by list(set(v) - set(own captions)) — NOTE the
    set() makes the order (and therefore random.sample's pick) implementation-defined in the
    reference; the oracle returns the *sets* before sampling plus a sample drawn from the sorted
    candidates, and tests compare set-level only."""
    hn_img = {k: list(v) for k, v in hard_neg_img.items()}
    for k, v in hn_img.items():
        if train_txt2img[k] in v:
            v.remove(train_txt2img[k])
    hn_txt = {k: set(v) - set(train_img2txt[k]) for k, v in hard_neg_txt.items()}
    rng = rng or _random.Random(0)
    sample = sample or rng.sample        # (``sample``: a deterministic stand-in for random.sample of :62-63, for output-level comparisons)
    sampled_txt = {k: sample(sorted(v), num_hard_negatives) for k, v in hn_txt.items()}
    sampled_img = {k: sample(v, num_hard_negatives) for k, v in hn_img.items()}
    return hn_img, hn_txt, sampled_txt, sampled_img


class DenseHNSWFlatIndexerOracle:
    """CPU restatement of dvl/indexer/faiss_indexers.py:90-154 with EXACT search standing in for faiss.IndexHNSWFlat
    (third-party, approximate; parity unpinned like IndexFlatIP): rows augmented with sqrt(phi - |x|^2), queries with 0,
    squared L2 distances ascending (= inner products descending), ids through the id list."""

    def __init__(self, vector_sz: int, buffer_size: int = 50000):
        self.buffer_size = buffer_size
        self.index_id_to_db_id = []
        self.rows = np.zeros((0, vector_sz + 1), dtype=np.float32)
        self.phi = 0

    def index_data(self, data):
        if self.phi > 0:
            raise RuntimeError('DPR HNSWF index needs to index all data at once,'
                               'results will be unpredictable otherwise.')
        phi = 0
        for _, v in data:                                            # :114-118
            phi = max(phi, (np.asarray(v) ** 2).sum())
        self.phi = 0                                                 # :119
        for i in range(0, len(data), self.buffer_size):              # :122-134
            chunk = data[i:i + self.buffer_size]
            vectors = [np.reshape(np.asarray(t[1], dtype=np.float32), (1, -1)) for t in chunk]
            norms = [(v ** 2).sum() for v in vectors]
            aux = [np.sqrt(phi - n) for n in norms]
            aug = np.concatenate([np.hstack((v, np.reshape(a, (-1, 1)))) for v, a in zip(vectors, aux)], axis=0)
            self.index_id_to_db_id.extend([t[0] for t in chunk])
            self.rows = np.concatenate([self.rows, aug.astype(np.float32)], axis=0)

    def search_knn(self, query_vectors, top_docs):
        q = np.asarray(query_vectors, dtype=np.float32)
        qa = np.hstack((q, np.zeros((len(q), 1), dtype=np.float32)))                       # :141-142
        d2 = ((qa[:, None, :].astype(np.float64) - self.rows[None, :, :].astype(np.float64)) ** 2).sum(-1)
        order = np.argsort(d2, axis=1, kind='stable')[:, :top_docs]
        scores = np.take_along_axis(d2, order, axis=1).astype(np.float32)
        return [([self.index_id_to_db_id[i] for i in row], scores[j]) for j, row in enumerate(order.tolist())]


# --------------------------------------------------------------------------------------
# re-ranker hook  (rerank.py:160-204 first stage, :256-290 re-ranking; a flat script in the reference, so there is no
# function to import and no golden vector: this restatement is the checker of lightningdot_amd/rerank.py)
# --------------------------------------------------------------------------------------
RERANK_RECALL_TOPS = (1, 5, 10, 20, 50, 100)     # rerank.py:160-161
RERANK_THRESHOLDS = (10, 20, 50, 100)            # rerank.py:257,273


def rerank_first_stage(batches: Iterable[dict], indexer_img, indexer_txt, img2txt: Dict, txt2img: Dict):
    """rerank.py:168-204 with the towers factored out (batches as in ``eval_on_stream``: 'txt_index', 'img_fname', 'q' text vectors,
    'ctx' image vectors): top-max(RECALL_TOPS) ids per text query / image query from the oracle indexers (:189-190), the dicts keep the
    result of an id's last occurrence (:195,200), hits are counted over every occurrence (:196-197,201-202).
    -> (ranking_res_img {txt_id: [img ids]}, ranking_res_txt {img_id: [txt ids]}, recall_img2, recall_txt2, total_len)"""
    recall_img2 = {t: 0 for t in RERANK_RECALL_TOPS}
    recall_txt2 = {t: 0 for t in RERANK_RECALL_TOPS}
    ranking_res_img, ranking_res_txt = {}, {}
    total_len = 0
    n_top = max(RERANK_RECALL_TOPS)
    for b in batches:
        res_img = [r[0] for r in indexer_img.search_knn(np.asarray(b['q'], np.float32), n_top)]
        res_txt = [r[0] for r in indexer_txt.search_knn(np.asarray(b['ctx'], np.float32), n_top)]
        total_len += len(res_img)
        for r, txt_index in zip(res_img, b['txt_index']):
            ranking_res_img[txt_index] = r
            for top in recall_img2:
                recall_img2[top] += txt2img[txt_index] in r[:top]
        for r, img_index in zip(res_txt, b['img_fname']):
            ranking_res_txt[img_index] = r
            for top in recall_txt2:
                recall_txt2[top] += any([txt_id in r[:top] for txt_id in img2txt[img_index]])
    return ranking_res_img, ranking_res_txt, recall_img2, recall_txt2, total_len


def rerank_recall(rankings: Dict, query_ids: Sequence, score, is_hit, denominator: int,
                  thresholds: Sequence[int] = RERANK_THRESHOLDS):
    """rerank.py:256-270 (image retrieval: query_ids = txt_ids, is_hit = txt2img[q] in ids, denominator = total_len) and :272-290
    (text retrieval: query_ids = img_ids, is_hit = any own caption in ids, denominator = len(img_ids)): for every threshold the external
    scorer's top 10 (``scores.topk(10, 0)``: descending, :263,280) of the first ``threshold`` first-stage candidates, Recall@{1,5,10}.
    ``score(query_id, candidate_id)`` is the external scorer (scores_mat[...] of :262, or scores_ir[...].get(id, -1000) of :260)."""
    out = {}
    for threshold in thresholds:
        recall_rerank = {1: 0, 5: 0, 10: 0}
        for qid in query_ids:
            cands = rankings[qid][:threshold]
            scores = np.asarray([score(qid, c) for c in cands], dtype=np.float32)
            idx = np.argsort(-scores, kind='stable')[:10]
            kept = [cands[i] for i in idx]
            for top in recall_rerank:
                recall_rerank[top] += bool(is_hit(qid, kept[:top]))
        out[threshold] = {t: v / float(denominator) for t, v in recall_rerank.items()}
    return out
