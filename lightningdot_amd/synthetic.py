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


class SyntheticItmDataset(torch.utils.data.Dataset):
    """Stand-in for data.ItmFastDataset when the LMDB-derived databases are absent: the same item tuple (so data.itm_fast_collate
    and the training loop run unchanged), the same ``new_epoch(hard_negatives_img, hard_negatives_txt)`` contract
    (dvl/data/itm.py:51-68), random region features / token ids.  Caption j belongs to image j // caps_per_img and — so that a
    model can actually learn the pairing — shares its first tokens with the other captions of that image and its regions carry an
    image-specific offset."""

    def __init__(self, n_img: int, caps_per_img: int = 2, txt_len: int = 12, num_bb: int = 10, vocab_size: int = 28996,
                 img_dim: int = 2048, num_hard_negatives: int = 0, seed: int = 0):
        g = torch.Generator().manual_seed(seed)
        self.num_hard_negatives = num_hard_negatives
        self.img_names = [f'img{i:05d}.npz' for i in range(n_img)]
        self.ids = [f'txt{i:05d}_{c}' for i in range(n_img) for c in range(caps_per_img)]
        self.ids_2_idx = {t: j for j, t in enumerate(self.ids)}
        self._img_of = [self.img_names[j // caps_per_img] for j in range(len(self.ids))]
        base = torch.randn(n_img, 1, img_dim, generator=g)
        self._feat = {n: (base[i] + 0.5 * torch.randn(num_bb, img_dim, generator=g)) for i, n in enumerate(self.img_names)}
        self._pos = {n: torch.rand(num_bb, 7, generator=g) for n in self.img_names}
        key = torch.randint(1000, vocab_size, (n_img, txt_len // 2), generator=g)
        self._tok = {}
        for j, t in enumerate(self.ids):
            tail = torch.randint(1000, vocab_size, (txt_len - 1 - key.shape[1],), generator=g)
            self._tok[t] = torch.cat([torch.tensor([101]), key[j // caps_per_img], tail])
        self.img2txts = {n: [t for t, f in zip(self.ids, self._img_of) if f == n] for n in self.img_names}
        self.txt2img = dict(zip(self.ids, self._img_of))
        self.datasets = [self]                    # the reference iterates train_dataset.datasets (train_itm.py:188-190)
        self.lens = [txt_len + num_bb] * len(self.ids)
        self.new_epoch()

    def __len__(self):
        return len(self.ids)

    def new_epoch(self, hard_negatives_img=None, hard_negatives_txt=None):
        nh = self.num_hard_negatives
        self.neg_imgs, self.neg_txts = [], []
        for t, f in zip(self.ids, self._img_of):
            if hard_negatives_img is not None and nh > 0:
                self.neg_imgs.append(hard_negatives_img[t][:nh])
                self.neg_txts.append(hard_negatives_txt[f][:nh])
            else:
                self.neg_imgs.append(None)
                self.neg_txts.append(None)

    def __getitem__(self, i):
        t, f = self.ids[i], self._img_of[i]
        input_ids = self._tok[t]
        nb = self._feat[f].shape[0]
        neg_imgs = neg_txts = None
        if self.neg_imgs[i] is not None:
            neg_imgs = {'img_input_ids': [], 'img_feat': [], 'img_pos_feat': [], 'num_bb': [], 'attn_masks_img': [],
                        'caption_ids': [], 'attn_masks_captions': []}
            for n in self.neg_imgs[i]:
                neg_imgs['img_input_ids'].append(torch.tensor([101]))
                neg_imgs['img_feat'].append(self._feat[n])
                neg_imgs['img_pos_feat'].append(self._pos[n])
                neg_imgs['num_bb'].append(self._feat[n].shape[0])
                neg_imgs['attn_masks_img'].append(torch.ones(self._feat[n].shape[0] + 1, dtype=torch.long))
            neg_txts = {'input_ids': [], 'position_ids': [], 'attention_mask': []}
            for n in self.neg_txts[i]:
                neg_txts['input_ids'].append(self._tok[n])
                neg_txts['attention_mask'].append(torch.ones(len(self._tok[n]), dtype=torch.long))
        return (input_ids, self._feat[f], self._pos[f], torch.tensor([101]), torch.ones(len(input_ids), dtype=torch.long),
                torch.ones(nb + 1, dtype=torch.long), t, f, neg_imgs, neg_txts, None, None)
