"""Hard-negative mining — host-side mirror of dvl/hn.py:45-66 on the MI355X retrieval path.

    sampled_hard_negatives   <- dvl/hn.py:45-66   (num_tops = min(max(2*nh+10, 50), 1000), :53)

The mining search (the largest retrieval in the reference: every train caption x every train image and back, top-k
up to 1000, every epoch) runs through ``eval_model_on_dataloader`` -> DenseFlatIndexer -> fused HIP search; the results stay
on the device as label tensors, positives are stripped with masks and the nh negatives per query are drawn there
(``device_hard_negatives``) — the reference's per-id Python lists (145k x 1000 objects per direction at Flickr-train scale) are
only built when a host sampler is requested.
"""
import collections
import random
from typing import Callable, Dict, Iterable

from .harness import eval_model_on_dataloader


def num_hard_sampled(num_hard_negatives: int) -> int:
    """dvl/hn.py:53"""
    return min(max(num_hard_negatives * 2 + 10, 50), 1000)


def postprocess_hard_negatives(hard_neg_img: Dict, hard_neg_txt: Dict, train_img2txt: Dict, train_txt2img: Dict,
                               num_hard_negatives: int, sample: Callable = random.sample):
    """dvl/hn.py:57-63: strip the positive image from each text's ranked list (in place, first occurrence, :57),
    strip an image's own captions from its ranked texts (set difference, :58), then sample nh of each (:62-63)."""
    [v.remove(train_txt2img[k]) for k, v in hard_neg_img.items() if train_txt2img[k] in v]
    hard_neg_txt = {k: list(set(v) - set(train_img2txt[k])) for k, v in hard_neg_txt.items()}
    hard_negs_txt = {k: sample(v, num_hard_negatives) for k, v in hard_neg_txt.items()}
    hard_negs_img = {k: sample(v, num_hard_negatives) for k, v in hard_neg_img.items()}
    return hard_negs_txt, hard_negs_img


def admissible_populations(rank_txt_res, rank_img_res, train_img2txt: Dict, train_txt2img: Dict):
    """What dvl/hn.py:57-58 leaves of the two mining searches, as device masks over their label tensors (RankDict.labels):
      * text query k: its ranked images without the positive image train_txt2img[k] (:57);
      * image query k: its ranked texts without its own captions train_img2txt[k] (:58).
    -> ((txt_ids, labels [n_txt, k], ok [n_txt, k]), (img_ids, labels [n_img, k], ok [n_img, k])): row i belongs to query id ids[i], the
    population the reference samples from is ``{db_ids[l] for l in labels[i][ok[i]]}`` (padding labels, -1, are never admissible)."""
    import numpy as np
    import torch
    dev = rank_txt_res.labels.device

    def mask(labels, banned):
        # labels [n, k] (-1 = padding), banned [n, m] rows that must not be drawn (-2 = unused slot)
        ok = labels >= 0
        for c in range(banned.shape[1]):       # (m is 1 or the captions per image: a handful of [n, k] comparisons, no [n, k, m] temporary)
            ok &= labels != banned[:, c:c + 1]
        return ok

    # text side: one banned row (the positive image)
    txt_ids = rank_txt_res.keys_in_row_order()
    img_row = {k: r for r, k in enumerate(rank_txt_res.db_ids)}
    banned = torch.from_numpy(np.fromiter((img_row.get(train_txt2img[t], -2) for t in txt_ids), dtype=np.int64, count=len(txt_ids)))
    ok_txt = mask(rank_txt_res.labels, banned.to(dev)[:, None])
    # image side: the image's own captions are banned
    img_ids = rank_img_res.keys_in_row_order()
    txt_row = {k: r for r, k in enumerate(rank_img_res.db_ids)}
    ncap = max([len(train_img2txt[i]) for i in img_ids] + [1])
    banned = np.full((len(img_ids), ncap), -2, dtype=np.int64)
    for j, i in enumerate(img_ids):
        rows = [txt_row.get(t, -2) for t in train_img2txt[i]]
        banned[j, :len(rows)] = rows
    ok_img = mask(rank_img_res.labels, torch.from_numpy(banned).to(dev))
    return (txt_ids, rank_txt_res.labels, ok_txt), (img_ids, rank_img_res.labels, ok_img)


def device_hard_negatives(rank_txt_res, rank_img_res, train_img2txt: Dict, train_txt2img: Dict, num_hard_negatives: int,
                          generator=None):
    """dvl/hn.py:57-63 on the device label tensors of the two mining searches: positives stripped by ``admissible_populations``
    (:57-58), then nh of the rest drawn per query (:62-63).
    Sampling is uniform without replacement like random.sample — iid keys on the admissible positions, the nh largest win — from a
    seeded device generator (the reference draws from Python's global RNG; pass ``sample=`` to sampled_hard_negatives for a
    host-side draw in the reference's order).  Only the nh winners per query become Python objects."""
    import numpy as np
    import torch
    nh = num_hard_negatives

    def draw(labels, ok):
        if int(ok.sum(dim=1).min().item()) < nh:
            raise ValueError('Sample larger than population or is negative')      # what random.sample raises (dvl/hn.py:62-63)
        keys = torch.rand(labels.shape, device=labels.device, generator=generator)
        keys = torch.where(ok, keys, keys.new_full((), -1.0))
        pick = keys.topk(nh, dim=1).indices
        return torch.gather(labels, 1, pick).cpu().numpy()

    def as_dict(query_ids, db_ids, won):
        # one object-array gather instead of n x nh list lookups
        names = np.empty(len(db_ids), dtype=object)
        names[:] = db_ids
        return dict(zip(query_ids, names[won].tolist()))

    (txt_ids, lab_txt, ok_txt), (img_ids, lab_img, ok_img) = admissible_populations(rank_txt_res, rank_img_res, train_img2txt,
                                                                                     train_txt2img)
    hard_negs_img = as_dict(txt_ids, rank_txt_res.db_ids, draw(lab_txt, ok_txt))
    hard_negs_txt = as_dict(img_ids, rank_img_res.db_ids, draw(lab_img, ok_img))
    return hard_negs_txt, hard_negs_img


def sampled_hard_negatives(train_dataloaders_hn: Iterable, args, bi_encoder, train_img2txt, train_txt2img, sample: Callable = None,
                           generator=None):
    """dvl/hn.py:45-66.  ``train_dataloaders_hn`` yields one evaluation-style dataloader per training set (the
    reference builds them itself from LMDB paths at :46-50; dataset construction is outside the hot path).
    Returns ({img_fname: [txt_id]*nh}, {txt_id: [img_fname]*nh}) exactly as consumed by ItmFastDataset.new_epoch
    (dvl/data/itm.py:60-62).

    Default: positive stripping and sampling run on the device over the searches' label tensors (device_hard_negatives); the two
    searches report the top-num_tops SETS (ids-only: the scores are dropped at :54-55 anyway, so only the candidates at the boundary of a
    set are re-scored from the fp32 rows).  With
    ``sample=`` (e.g. ``random.sample``) the reference's host-side post-processing runs instead, in the reference's order, on id
    lists materialised from the same searches (golden G4 pins that path)."""
    hard_negs_txt_all, hard_negs_img_all = [], []
    for loader in train_dataloaders_hn:
        n_top = num_hard_sampled(args.num_hard_negatives)
        # (the device route consumes the top-n_top SETS: ids-only searches, no recall; the host route keeps the reference's ranked lists)
        loss_hard, correct_ratio_hard, indexer_hard, recall_hard, (hard_neg_img, hard_neg_txt) = \
            eval_model_on_dataloader(bi_encoder, loader, args, train_img2txt, n_top, rank_sets_only=sample is None)
        if sample is None:
            hn_txt, hn_img = device_hard_negatives(hard_neg_img, hard_neg_txt, train_img2txt, train_txt2img,
                                                   args.num_hard_negatives, generator=generator)
        else:
            hn_txt, hn_img = postprocess_hard_negatives(dict(hard_neg_img), dict(hard_neg_txt), train_img2txt, train_txt2img,
                                                        args.num_hard_negatives, sample=sample)
        hard_negs_txt_all.append(hn_txt)
        hard_negs_img_all.append(hn_img)
    def merged(maps):
        # dict(collections.ChainMap(*maps)) of :65-66 — the FIRST map that holds a key wins — without ChainMap's per-key Python lookups
        if len(maps) == 1:
            return maps[0] if type(maps[0]) is dict else dict(maps[0])
        out = {}
        for m in reversed(maps):
            out.update(m)
        return out
    return merged(hard_negs_txt_all), merged(hard_negs_img_all)
