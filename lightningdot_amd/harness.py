"""Evaluation harness — host-side mirror of dvl/trainer.py:93-190 on the MI355X indexer / loss.

    get_indexer                <- dvl/trainer.py:93-110
    eval_model_on_dataloader   <- dvl/trainer.py:113-190

Same arguments, same return tuple ``(loss, correct_ratio, (indexer_img, indexer_txt), (recall_txt, recall_img),
(rank_txt_res, rank_img_res))`` and the same quirks: image queries are NOT de-duplicated before the
search (:138-139), index sides are de-duplicated by dict key with last-write-wins (:151-152), denominators are the
numbers of unique query ids (:179,188), ``recall_txt`` is text-query -> image retrieval (:190).

What changed underneath: embeddings never leave the device (the per-vector ``.detach().cpu().numpy()`` of
:135,138,151-152 is gone), both indexes are built from device tensors and searched by the fused HIP path.
"""
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
                             no_eval=False):
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
    indexer_img.index_tensor(img_keys, query_img_t[torch.as_tensor(img_last, device=dev)])
    indexer_txt.index_tensor(txt_keys, query_txt_t[torch.as_tensor(txt_last, device=dev)])

    if no_eval:
        return total_loss, correct_ratio, (indexer_img, indexer_txt), (None, None), (None, None)

    res_txt = indexer_img.search_knn(query_txt_t, num_tops)
    rank_txt_res = {query_txt_id[i]: r[0] for i, r in enumerate(res_txt)}
    res_img = indexer_txt.search_knn(query_img_t, num_tops)
    rank_img_res = {query_img_id[i]: r[0] for i, r in enumerate(res_img)}

    recall_txt = {1: 0, 5: 0, 10: 0}
    for i, q in enumerate(query_txt_id):
        for top in recall_txt:
            recall_txt[top] += labels_img_name[i] in rank_txt_res[q][:top]
    for top in recall_txt:
        recall_txt[top] = recall_txt[top] / len(rank_txt_res)

    recall_img = {1: 0, 5: 0, 10: 0}
    for i, q in enumerate(np.unique(query_img_id)):
        for top in recall_img:
            recall_img[top] += any([txt_id in rank_img_res[q][:top] for txt_id in img2txt[q]])
    for top in recall_img:
        recall_img[top] = recall_img[top] / len(rank_img_res)

    return total_loss, correct_ratio, (indexer_img, indexer_txt), (recall_txt, recall_img), (rank_txt_res, rank_img_res)
