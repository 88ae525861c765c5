"""Synthetic batches in the reference's collate layout (dvl/data/itm.py:254-287, itm_fast_collate) — the LMDB feature /
text databases are not available and their readers need lmdb / lz4 / msgpack_numpy, so smoke runs, tests
and benchmarks of the full tower -> index -> search -> recall flow use these."""
import torch


def synthetic_itm_batches(n_img: int, caps_per_img: int = 5, batch_size: int = 80, txt_len: int = 20, num_bb: int = 36,
                          vocab_size: int = 28996, img_dim: int = 2048, seed: int = 0, device='cpu',
                          num_hard_negatives: int = 0):
    """Yields batches (one item per caption, like ItmFastDataset) plus the img2txt mapping.  With
    ``num_hard_negatives`` the negatives are appended after the ``sample_size`` positives exactly as the reference
    collate does (dvl/data/itm.py:283-284)."""
    g = torch.Generator().manual_seed(seed)
    img_feats = torch.randn(n_img, num_bb, img_dim, generator=g)
    img_pos = torch.rand(n_img, num_bb, 7, generator=g)
    items, img2txt = [], {}
    for i in range(n_img):
        for c in range(caps_per_img):
            tid, iid = f'txt{i:05d}_{c}', f'img{i:05d}.npz'
            ids = torch.randint(1000, vocab_size, (txt_len,), generator=g)
            ids[0] = 101
            items.append((tid, iid, ids, i))
            img2txt.setdefault(iid, []).append(tid)
    batches = []
    for b0 in range(0, len(items), batch_size):
        chunk = items[b0:b0 + batch_size]
        bs = len(chunk)
        extra = []
        if num_hard_negatives > 0:
            for _ in range(bs * num_hard_negatives):
                extra.append(items[int(torch.randint(0, len(items), (1,), generator=g))])
        allc = chunk + extra
        n = len(allc)
        input_ids = torch.stack([c[2] for c in allc])
        feat = torch.stack([img_feats[c[3]] for c in allc])
        pos = torch.stack([img_pos[c[3]] for c in allc])
        mk = lambda t: t.to(device)
        batches.append({
            'txts': dict(input_ids=mk(input_ids), position_ids=mk(torch.arange(txt_len).unsqueeze(0)),
                         attention_mask=mk(torch.ones(n, txt_len, dtype=torch.long)), img_feat=None, img_pos_feat=None,
                         img_masks=None, gather_index=None),
            'imgs': dict(input_ids=mk(torch.full((n, 1), 101)), position_ids=mk(torch.zeros(1, 1, dtype=torch.long)),
                         attention_mask=mk(torch.ones(n, 1 + num_bb, dtype=torch.long)), img_feat=mk(feat),
                         img_pos_feat=mk(pos), img_masks=None,
                         gather_index=mk(torch.arange(1 + num_bb).unsqueeze(0).repeat(n, 1))),
            'caps': dict(input_ids=None, position_ids=None, attention_mask=None, img_feat=None, img_pos_feat=None,
                         img_masks=None, gather_index=None),
            'sample_size': bs, 'pos_ctx_indices': list(range(bs)), 'neg_ctx_indices': list(range(bs, n)),
            'txt_index': [c[0] for c in allc], 'img_fname': [c[1] for c in allc]})
    return batches, img2txt


def s2_embeddings(n_img: int, d: int = 768, caps_per_img: int = 5, noise: float = 0.9, seed: int = 7, device='cpu'):
    """SURVEY §8d S2 stand-in for the Flickr-1k / COCO-5k embedding sets (the real checkpoints and LMDBs are absent):
    ``n_img`` image rows N(0,1) and ``caps_per_img`` captions per image, caption = image + noise * N(0,1).
    -> (img [n_img, d], txt [n_img * caps_per_img, d]) fp32; caption j belongs to image j // caps_per_img."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(n_img, d, generator=g)
    txt = img.repeat_interleave(caps_per_img, 0) + noise * torch.randn(n_img * caps_per_img, d, generator=g)
    return img.to(device), txt.to(device)
