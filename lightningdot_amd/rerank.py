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
