"""Hard-negative mining — host-side mirror of dvl/hn.py:45-66 on the MI355X retrieval path.

    sampled_hard_negatives   <- dvl/hn.py:45-66   (num_tops = min(max(2*nh+10, 50), 1000), :53)

The mining search (the largest retrieval in the reference: every train caption x every train image and back, top-k
up to 1000, every epoch) runs through ``eval_model_on_dataloader`` -> DenseFlatIndexer -> fused HIP search.
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


def sampled_hard_negatives(train_dataloaders_hn: Iterable, args, bi_encoder, train_img2txt, train_txt2img):
    """dvl/hn.py:45-66.  ``train_dataloaders_hn`` yields one evaluation-style dataloader per training set (the
    reference builds them itself from LMDB paths at :46-50; dataset construction is outside the hot path).
    Returns ({img_fname: [txt_id]*nh}, {txt_id: [img_fname]*nh}) exactly as consumed by ItmFastDataset.new_epoch
    (dvl/data/itm.py:60-62)."""
    hard_negs_txt_all, hard_negs_img_all = [], []
    for loader in train_dataloaders_hn:
        n_top = num_hard_sampled(args.num_hard_negatives)
        loss_hard, correct_ratio_hard, indexer_hard, recall_hard, (hard_neg_img, hard_neg_txt) = \
            eval_model_on_dataloader(bi_encoder, loader, args, train_img2txt, n_top)
        hn_txt, hn_img = postprocess_hard_negatives(hard_neg_img, hard_neg_txt, train_img2txt, train_txt2img,
                                                    args.num_hard_negatives)
        hard_negs_txt_all.append(hn_txt)
        hard_negs_img_all.append(hn_img)
    hard_negs_txt_all = dict(collections.ChainMap(*hard_negs_txt_all))
    hard_negs_img_all = dict(collections.ChainMap(*hard_negs_img_all))
    return hard_negs_txt_all, hard_negs_img_all
