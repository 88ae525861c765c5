"""Evaluation harness — host-side mirror of dvl/trainer.py:93-190 on the MI355X indexer / loss.

    get_indexer                <- dvl/trainer.py:93-110
    eval_model_on_dataloader   <- dvl/trainer.py:113-190

Same arguments, same return tuple ``(loss, correct_ratio, (indexer_img, indexer_txt), (recall_txt, recall_img),
(rank_txt_res, rank_img_res))`` and the same quirks: image queries are NOT de-duplicated before the
search (:138-139) but only the result of an id's last occurrence survives in the rank dicts (:168,171) — so only that
occurrence is searched here —, index sides are de-duplicated by dict key with last-write-wins (:151-152), denominators are
the numbers of unique query ids (:179,188), ``recall_txt`` is text-query -> image retrieval (:190).

What changed underneath: embeddings never leave the device (the per-vector ``.detach().cpu().numpy()`` of
:135,138,151-152 is gone), both indexes are built from device tensors and searched by the fused HIP path, the searches return
device label tensors, Recall@k is a device reduction over them and the rank dicts are lazy views (``RankDict``).
"""
from collections.abc import Mapping
from typing import Dict, Optional

import numpy as np
import torch

from .indexer import DenseFlatIndexer, DenseHNSWFlatIndexer
from .loss import BiEncoderNllLoss, _calc_loss


def _dedup_last(ids):
    """dict.update semantics of :151-152: key order = first insertion, value = last write."""
    pos = {}
    for i, k in enumerate(ids):
        pos[k] = i
    return list(pos.keys()), list(pos.values())


class RankDict(Mapping):
    """The reference's ``rank_*_res`` dict ({query id: [db ids of its top results]}, dvl/trainer.py:168,171) over a device label
    tensor with ONE row per distinct query id: key order = first occurrence of a query id, value = the result of its LAST
    occurrence (dict-comprehension semantics) — which is why only the last occurrence of every id is searched at all (the
    reference searches the image vector of every (caption, image) pair, 5 identical searches per image at :138-139,170, and then
    keeps one).  Id lists are materialised per key on access; ``labels`` / ``last_rows`` give the tensor view for device-side
    consumers (recall below, hard-negative mining in hn.py)."""

    def __init__(self, keys, labels: torch.Tensor, db_ids: list):
        self._pos = {q: i for i, q in enumerate(keys)}
        assert len(self._pos) == labels.shape[0], 'one result row per distinct query id'
        self.labels = labels                 # [n distinct ids, k] int64 row labels on the device (-1 = padding)
        self.db_ids = db_ids
        self._host = None

    def keys_in_row_order(self):
        """the query ids, key i <-> row i of ``labels`` (dicts keep insertion order)"""
        return list(self._pos)

    def last_rows(self, query_ids):
        """result row of every id in ``query_ids`` (the result of the id's last occurrence in the query stream)"""
        return [self._pos[q] for q in query_ids]

    def _host_labels(self):
        if self._host is None:
            self._host = self.labels.cpu().numpy()      # one D2H, on the first id lookup
        return self._host

    def __getitem__(self, q):
        ids = self.db_ids
        return [ids[i] for i in self._host_labels()[self._pos[q]].tolist()]

    def __iter__(self):
        return iter(self._pos)

    def __len__(self):
        return len(self._pos)


def get_indexer(bi_encoder, eval_dataloader, args, hnsw_index, img_retrieval=True):
    """dvl/trainer.py:93-110 (``hnsw_index`` selects the reference's DenseHNSWFlatIndexer surface, exact-backed here)."""
    bi_encoder.eval()
    ids, vecs = [], []
    for batch in eval_dataloader:
        with torch.no_grad():
            local_q_vector, local_ctx_vectors, local_caption_vectors = bi_encoder(batch)
        if img_retrieval:
            ids.extend(batch['img_fname'])
            vecs.append(local_ctx_vectors.detach())
        else:
            ids.extend(batch['txt_index'])
            vecs.append(local_q_vector.detach())
    allv = torch.cat(vecs, 0)
    keys, last = _dedup_last(ids)
    indexer = (DenseHNSWFlatIndexer if hnsw_index else DenseFlatIndexer)(args.vector_size)
    indexer.index_tensor(keys, allv[torch.as_tensor(last, device=allv.device)])
    return indexer


def eval_model_on_dataloader(bi_encoder, eval_dataloader, args, img2txt: Optional[Dict] = None, num_tops=100,
                             no_eval=False, rank_sets_only=False):
    """dvl/trainer.py:113-190.  ``rank_sets_only`` (not in the reference; what the mining of dvl/hn.py:54-63 needs): the rank dicts hold
    the top-``num_tops`` SETS — the searches re-score only their boundary candidates (LDOT_OPT_RESULT_SET) and the order inside a list is
    approximate, so Recall@{1,5,10} is not computed: the recall pair comes back as (None, None)."""
    total_loss = 0.0
    bi_encoder.eval()
    total_correct_predictions = 0
    batches, total_samples = 0, 0
    labels_img_name, labels_txt_name = [], []
    query_txt, query_txt_id = [], []
    query_img, query_img_id = [], []
    loss_terms, correct_terms = [], []
    for i, batch in enumerate(eval_dataloader):
        with torch.no_grad():
            local_q_vector, local_ctx_vectors, local_caption_vectors = bi_encoder(batch)
            query_txt.append(local_q_vector.detach().reshape(local_q_vector.shape[0], -1))
            query_txt_id.extend(batch['txt_index'])
            query_img.append(local_ctx_vectors.detach().reshape(local_ctx_vectors.shape[0], -1))
            query_img_id.extend(batch['img_fname'])
            loss_function = BiEncoderNllLoss()
            loss, correct_cnt, score = _calc_loss(args, loss_function, local_q_vector, local_ctx_vectors,
                                                  local_caption_vectors, list(range(len(local_q_vector))), None)
        loss_terms.append(loss.detach())          # .item() deferred: one sync at the end instead of one per batch
        correct_terms.append(correct_cnt.detach())
        batches += 1
        total_samples += batch['txts']['input_ids'].shape[0]
        labels_img_name.extend(batch['img_fname'])
        labels_txt_name.extend(batch['txt_index'])

    total_loss = sum(float(l.item()) for l in loss_terms) / batches
    total_correct_predictions = sum(int(c.sum().item()) for c in correct_terms)
    correct_ratio = total_correct_predictions / float(total_samples)

    query_txt_t = torch.cat(query_txt, 0)
    query_img_t = torch.cat(query_img, 0)
    indexer_cls = DenseHNSWFlatIndexer if getattr(args, 'hnsw_index', False) else DenseFlatIndexer   # trainer.py:122-127
    indexer_img = indexer_cls(args.vector_size)
    indexer_txt = indexer_cls(args.vector_size)
    img_keys, img_last = _dedup_last(query_img_id)
    txt_keys, txt_last = _dedup_last(query_txt_id)
    dev = query_txt_t.device
    img_vecs = query_img_t[torch.as_tensor(img_last, device=dev)]      # one row per distinct id (its last occurrence)
    txt_vecs = query_txt_t[torch.as_tensor(txt_last, device=dev)]
    indexer_img.index_tensor(img_keys, img_vecs)
    indexer_txt.index_tensor(txt_keys, txt_vecs)

    if no_eval:
        return total_loss, correct_ratio, (indexer_img, indexer_txt), (None, None), (None, None)

    # Both searches stay on the device (scores / row labels [nq, num_tops]); the reference's per-result Python objects (:167-170:
    # nq x num_tops ids per direction) are built lazily, only for the ids a caller actually indexes (RankDict).
    # Only the LAST occurrence of every query id is searched: the reference's dict comprehensions (:168,171) keep exactly that
    # result, so its 5x duplicated image queries (:138-139) cost 5x the work for the same rank dict.  The de-duplicated query
    # rows are the rows just indexed on the other side (same id stream, same last-write-wins rule).
    _, lab_txt = indexer_img.search_knn_tensors(txt_vecs, num_tops, ids_only=rank_sets_only)     # text query -> image rows
    _, lab_img = indexer_txt.search_knn_tensors(img_vecs, num_tops, ids_only=rank_sets_only)     # image query -> text rows
    rank_txt_res = RankDict(txt_keys, lab_txt, indexer_img.index_id_to_db_id)
    rank_img_res = RankDict(img_keys, lab_img, indexer_txt.index_id_to_db_id)
    if rank_sets_only:
        return total_loss, correct_ratio, (indexer_img, indexer_txt), (None, None), (rank_txt_res, rank_img_res)

    # Recall@{1,5,10} (:173-188) as device reductions over the label tensors.  A result list belongs to a query ID (dict
    # semantics, last occurrence wins); ids are unique per index, so "id in list[:top]" is "row label in labels[:, :top]"; a padding
    # label (-1, fewer than num_tops rows) reads as the LAST id through the reference's negative indexing (faiss_indexers.py:85).
    tops = (1, 5, 10)
    img_row = {k: r for r, k in enumerate(img_keys)}
    want = torch.as_tensor([img_row.get(n, -2) for n in labels_img_name], device=dev)
    used = lab_txt[torch.as_tensor(rank_txt_res.last_rows(query_txt_id), device=dev)]
    used = torch.where(used < 0, used.new_tensor(len(img_keys) - 1), used)
    hit_pos = (used == want[:, None]).int().argmax(dim=1)                  # first matching position (0 when there is none)
    hit_any = (used == want[:, None]).any(dim=1)
    recall_txt = {top: int((hit_any & (hit_pos < top)).sum().item()) / len(rank_txt_res) for top in tops}

    uniq_img = list(np.unique(query_img_id))
    txt_row = {k: r for r, k in enumerate(txt_keys)}
    ncap = max([len(img2txt[q]) for q in uniq_img] + [1])
    caps = torch.full((len(uniq_img), ncap), -2, dtype=torch.int64)
    for i, q in enumerate(uniq_img):
        rows = [txt_row.get(t, -2) for t in img2txt[q]]
        caps[i, :len(rows)] = torch.as_tensor(rows, dtype=torch.int64)
    caps = caps.to(dev)
    used = lab_img[torch.as_tensor(rank_img_res.last_rows(uniq_img), device=dev)]
    used = torch.where(used < 0, used.new_tensor(len(txt_keys) - 1), used)
    recall_img = {}
    for top in tops:
        m = (used[:, :top, None] == caps[:, None, :]).any(dim=2).any(dim=1)
        recall_img[top] = int(m.sum().item()) / len(rank_img_res)

    return total_loss, correct_ratio, (indexer_img, indexer_txt), (recall_txt, recall_img), (rank_txt_res, rank_img_res)
