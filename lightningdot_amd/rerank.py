"""Re-ranker hook — host-side mirror of the candidate export / re-scoring loops of the reference's rerank.py.

    first_stage_rankings  <- rerank.py:168-204  (encode a test loader, top-max(thresholds) ids per text and per image,
                                                 first-stage Recall@{1,5,10,20,50,100})
    rerank_recall         <- rerank.py:256-290  (for threshold in [10, 20, 50, 100]: re-score the first `threshold`
                                                 candidates with an external cross-encoder's scores, keep its top 10,
                                                 Recall@{1,5,10})

The external scorer (UNITER / OSCAR in the reference, loaded from pickles with hard-coded paths) is passed in as a
callable ``score(query_id, candidate_id) -> float``; everything up to the candidate lists runs on the MI355X retrieval path.
"""
from typing import Callable, Dict, Iterable, List, Sequence

import torch

RECALL_TOPS = (1, 5, 10, 20, 50, 100)     # rerank.py:160-161
THRESHOLDS = (10, 20, 50, 100)            # rerank.py:257,273


def first_stage_rankings(bi_encoder, indexer_img, indexer_txt, dataloader: Iterable, img2txt: Dict, txt2img: Dict):
    """-> (ranking_res_img {txt_id: [img ids]}, ranking_res_txt {img_id: [txt ids]}, recall_img, recall_txt, total_len)"""
    recall_img = {t: 0 for t in RECALL_TOPS}
    recall_txt = {t: 0 for t in RECALL_TOPS}
    ranking_res_img, ranking_res_txt = {}, {}
    total_len = 0
    n_top = max(RECALL_TOPS)
    for batch in dataloader:
        with torch.no_grad():
            txt_vec, img_vec, _ = bi_encoder(batch)
        res_img = [r[0] for r in indexer_img.search_knn(txt_vec.detach(), n_top)]
        res_txt = [r[0] for r in indexer_txt.search_knn(img_vec.detach(), n_top)]
        total_len += len(res_img)
        for r, txt_index in zip(res_img, batch['txt_index']):
            ranking_res_img[txt_index] = r
            for top in recall_img:
                recall_img[top] += txt2img[txt_index] in r[:top]
        for r, img_index in zip(res_txt, batch['img_fname']):
            ranking_res_txt[img_index] = r
            for top in recall_txt:
                recall_txt[top] += any([txt_id in r[:top] for txt_id in img2txt[img_index]])
    return ranking_res_img, ranking_res_txt, recall_img, recall_txt, total_len


def rerank_recall(rankings: Dict[object, List], score: Callable[[object, object], float],
                  is_hit: Callable[[object, Sequence], bool], thresholds: Sequence[int] = THRESHOLDS,
                  denominator: int = None, missing: float = -1000.0):
    """rerank.py:256-290: for every threshold keep the external scorer's top 10 of the first `threshold` first-stage
    candidates and count Recall@{1,5,10}.  ``is_hit(query_id, ids)`` says whether ``ids`` contain a positive of the query
    (``txt2img[q] in ids`` for image retrieval, ``any(t in ids for t in img2txt[q])`` for text retrieval).
    -> {threshold: {1: r, 5: r, 10: r}}"""
    den = len(rankings) if denominator is None else denominator
    out = {}
    for threshold in thresholds:
        recall = {1: 0, 5: 0, 10: 0}
        for qid, cands in rankings.items():
            first = cands[:threshold]
            scores = torch.tensor([float(s) if (s := score(qid, c)) is not None else missing for c in first])
            idx = scores.topk(min(10, len(first)), 0)[1]
            kept = [first[i.item()] for i in idx]
            for top in recall:
                recall[top] += bool(is_hit(qid, kept[:top]))
        out[threshold] = {t: v / float(den) for t, v in recall.items()}
    return out


# ---- device path: candidates stay label tensors, the external scores come as a matrix ----------------------------------------------
def first_stage_candidates(bi_encoder, indexer_img, indexer_txt, dataloader: Iterable, img2txt: Dict, txt2img: Dict):
    """rerank.py:168-204 without per-result Python objects: encodes the loader, searches top-max(RECALL_TOPS) both ways on the device
    and returns ``dict(txt_ids, img_ids, labels_img [n_txt, 100], labels_txt [n_img_queries, 100], pos_img [n_txt, 1],
    pos_txt [n_img_queries, P], recall_img, recall_txt, total_len)`` — label tensors index ``indexer_*.index_id_to_db_id``; the
    recall counters are the reference's (hits over every query occurrence, :195-204)."""
    n_top = max(RECALL_TOPS)
    txt_ids, img_ids, lab_img, lab_txt = [], [], [], []
    for batch in dataloader:
        with torch.no_grad():
            txt_vec, img_vec, _ = bi_encoder(batch)
        lab_img.append(indexer_img.search_knn_tensors(txt_vec.detach(), n_top)[1])
        lab_txt.append(indexer_txt.search_knn_tensors(img_vec.detach(), n_top)[1])
        txt_ids.extend(batch['txt_index'])
        img_ids.extend(batch['img_fname'])
    lab_img, lab_txt = torch.cat(lab_img), torch.cat(lab_txt)
    dev = lab_img.device
    img_row = {k: r for r, k in enumerate(indexer_img.index_id_to_db_id)}
    txt_row = {k: r for r, k in enumerate(indexer_txt.index_id_to_db_id)}
    pos_img = torch.as_tensor([[img_row.get(txt2img[t], -2)] for t in txt_ids], dtype=torch.int64, device=dev)
    ncap = max([len(img2txt[i]) for i in img_ids] + [1])
    pos_txt = torch.full((len(img_ids), ncap), -2, dtype=torch.int64)
    for j, i in enumerate(img_ids):
        rows = [txt_row.get(t, -2) for t in img2txt[i]]
        pos_txt[j, :len(rows)] = torch.as_tensor(rows, dtype=torch.int64)
    pos_txt = pos_txt.to(dev)

    def hits(labels, pos):
        return {top: int((labels[:, :top, None] == pos[:, None, :]).any(dim=2).any(dim=1).sum().item()) for top in RECALL_TOPS}
    # The reference's text-retrieval loop runs over the DISTINCT image ids and divides by their number (rerank.py:275-289; its
    # ranking_res_txt keeps the LAST occurrence of an id, :191-192): the row selection and the denominator that reproduce it
    last = {}
    for j, i in enumerate(img_ids):
        last[i] = j
    img_unique_rows = torch.as_tensor(list(last.values()), dtype=torch.int64, device=dev)
    return dict(txt_ids=txt_ids, img_ids=img_ids, labels_img=lab_img, labels_txt=lab_txt, pos_img=pos_img, pos_txt=pos_txt,
                recall_img=hits(lab_img, pos_img), recall_txt=hits(lab_txt, pos_txt), total_len=len(txt_ids),
                img_unique_rows=img_unique_rows, img_unique_ids=list(last.keys()))


def rerank_recall_device(labels: torch.Tensor, ext_scores: torch.Tensor, positives: torch.Tensor,
                         thresholds: Sequence[int] = THRESHOLDS, denominator: int = None, missing: float = -1000.0,
                         rows: torch.Tensor = None, pad_as_last_id: bool = True):
    """rerank.py:256-290 on the device, for an external scorer given as a matrix (the reference's ``scores_mat`` form, :229-233):
    ``labels`` [nq, K] first-stage candidates (index rows, -1 = padding), ``ext_scores`` [nq, n_db] the cross-encoder's score of
    (query, index row), ``positives`` [nq, P] the rows that count as hits (-2 = unused).  For every threshold the scorer's top 10 of
    the first ``threshold`` candidates are kept and Recall@{1,5,10} counted.  -> {threshold: {1: r, 5: r, 10: r}}

    ``rows`` selects the query rows that count (text retrieval: ``first_stage_candidates(...)['img_unique_rows']`` — one row per distinct
    image id, the last occurrence, which is what the reference's dict keeps and its loop over ``img_ids`` visits, rerank.py:275-289);
    the default denominator is the number of rows counted, i.e. ``len(img_ids)`` of the reference with that selection and ``total_len``
    for image retrieval.  A padding label (-1: the index holds fewer rows than candidates were asked for) is what the reference's
    ``index_id_to_db_id[-1]`` turns into the LAST db id (dvl/indexer/faiss_indexers.py:85); ``pad_as_last_id`` mirrors that, False gives
    such candidates the ``missing`` score instead."""
    if rows is not None:
        labels, ext_scores, positives = labels[rows], ext_scores[rows], positives[rows]
    if pad_as_last_id:
        labels = torch.where(labels >= 0, labels, labels.new_full((), ext_scores.shape[1] - 1))
    den = labels.shape[0] if denominator is None else denominator
    out = {}
    for threshold in thresholds:
        cand = labels[:, :threshold]
        s = torch.gather(ext_scores, 1, cand.clamp_min(0))
        s = torch.where(cand >= 0, s, s.new_full((), missing))
        idx = s.topk(min(10, cand.shape[1]), dim=1).indices
        kept = torch.gather(cand, 1, idx)
        out[threshold] = {top: int((kept[:, :top, None] == positives[:, None, :]).any(dim=2).any(dim=1).sum().item()) / float(den)
                          for top in (1, 5, 10)}
    return out
