#!/usr/bin/env python3
"""Writes a tiny text DB + image-feature DB in the layout lightningdot_amd.data reads (FlatDb container, the reference's VALUE
formats: lz4-frame(msgpack) text records as TxtLmdb.__setitem__ produces them, data.py:163-166; fp16 region features as
np.savez_compressed archives ('_compressed' DBs) and as msgpack + msgpack_numpy dicts).

    python tools/make_db_fixture.py OUT_DIR [n_images] [captions_per_image]

Deterministic (seeded); used by tests/test_data_readers.py (which builds it into tmp_path) and as a template for converted DBs."""
import io
import json
import os
import sys

import msgpack
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.data import FlatDbWriter, lz4_frame_compress, msgpack_numpy_encode   # noqa: E402


def make(out_dir: str, n_img: int = 6, cpi: int = 2, seed: int = 0, feat_dim: int = 2048):
    rng = np.random.default_rng(seed)
    txt_dir, img_dir = os.path.join(out_dir, 'txt_db'), os.path.join(out_dir, 'img_db')
    os.makedirs(txt_dir, exist_ok=True)
    os.makedirs(img_dir, exist_ok=True)
    # ---- text DB (uniter_model prepro layout: id2len / txt2img / img2txts / meta + one record per caption)
    w = FlatDbWriter(os.path.join(txt_dir, 'data'))
    id2len, txt2img, img2txts, examples = {}, {}, {}, {}
    for i in range(n_img):
        fname = f'flickr30k_{i:012d}.npz'
        for c in range(cpi):
            tid = str(i * cpi + c)
            n_tok = int(rng.integers(4, 70 if (i == 1 and c == 0) else 20))       # one caption longer than max_txt_len = 60
            if i == 1 and c == 0:
                n_tok = 65
            ex = {'id': tid, 'dataset': 'flickr30k', 'split': 'test', 'sent': f'caption {c} of image {i} ' + 'word ' * 6,
                  'sent_id': int(tid), 'img_fname': fname, 'image_id': i,
                  'input_ids': [int(t) for t in rng.integers(1000, 28000, n_tok)]}
            w.put(tid, lz4_frame_compress(msgpack.dumps(ex, use_bin_type=True)))
            id2len[tid], txt2img[tid] = n_tok, fname
            img2txts.setdefault(fname, []).append(tid)
            examples[tid] = ex
    w.close()
    json.dump(id2len, open(os.path.join(txt_dir, 'id2len.json'), 'w'))
    json.dump(txt2img, open(os.path.join(txt_dir, 'txt2img.json'), 'w'))
    json.dump(img2txts, open(os.path.join(txt_dir, 'img2txts.json'), 'w'))
    json.dump({'CLS': 101, 'SEP': 102, 'MASK': 103, 'v_range': [106, 28996], 'UNK': 100, 'bert': 'bert-base-cased'},
              open(os.path.join(txt_dir, 'meta.json'), 'w'))
    # ---- image DB: the thresholded, compressed flavour (feat_th0.2_max100_min10_compressed + nbb json) and the raw 'all' one
    feats, name2nbb = {}, {}
    wc = FlatDbWriter(os.path.join(img_dir, 'feat_th0.2_max100_min10_compressed'))
    wa = FlatDbWriter(os.path.join(img_dir, 'all'))
    for i in range(n_img):
        fname = f'flickr30k_{i:012d}.npz'
        n = int(rng.integers(12, 40))
        conf = np.sort(rng.uniform(0.05, 0.95, n).astype(np.float16))[::-1].copy()
        d = {'features': rng.standard_normal((n, feat_dim)).astype(np.float16),
             'norm_bb': rng.uniform(0, 1, (n, 6)).astype(np.float16), 'conf': conf}
        buf = io.BytesIO()
        np.savez_compressed(buf, **d)
        wc.put(fname, buf.getvalue())
        wa.put(fname, msgpack.dumps(d, default=msgpack_numpy_encode, use_bin_type=True))
        feats[fname] = d
        name2nbb[fname] = int(min(100, max(10, int((conf > 0.2).sum()))))
    wa.put('__keys__', json.dumps(list(feats)).encode('utf-8'))
    wc.close()
    wa.close()
    json.dump(name2nbb, open(os.path.join(img_dir, 'nbb_th0.2_max100_min10.json'), 'w'))
    return examples, feats, name2nbb


if __name__ == '__main__':
    make(sys.argv[1], *(int(a) for a in sys.argv[2:4]))
    print('wrote', sys.argv[1])
