"""On-disk readers, ItmFastDataset and itm_fast_collate pinned against the REFERENCE'S OWN classes (golden G8, oracle/gen_golden.py:
uniter_model.data.data.TxtTokLmdb / DetectFeatLmdb and dvl.data.itm.ItmFastDataset / itm_fast_collate run over an in-memory
stand-in for the lmdb container).  The golden holds the DB content (records, json side files, fp16 region features) and what the
reference returned; here the same content is written as FlatDb files — text records as REAL LZ4 frames of the same msgpack bytes —
and read back through lightningdot_amd.data.  Everything is compared exactly (integer ids / masks, fp16 -> fp32 features)."""
import io
import json
import os

import msgpack
import numpy as np
import pytest
import torch

from lightningdot_amd import data as D


class FakeTokenizer:
    """the tokenizer stand-in the golden was generated with (only encode / cls / sep are used by the caption branch)"""
    cls_token_id, sep_token_id = 101, 102

    def encode(self, text, add_special_tokens=False):
        return [200 + (sum(map(ord, w)) % 300) for w in text.split()]


def _write_db(tmp, inp, arr, flavour):
    txt_dir, img_dir = os.path.join(tmp, 'txt.db'), os.path.join(tmp, 'img')
    os.makedirs(txt_dir, exist_ok=True)
    os.makedirs(img_dir, exist_ok=True)
    for name in ('id2len', 'txt2img', 'img2txts', 'meta'):
        json.dump(inp[name], open(os.path.join(txt_dir, name + '.json'), 'w'))
    w = D.FlatDbWriter(os.path.join(txt_dir, 'data'))
    for tid, ex in inp['examples'].items():
        w.put(tid, D.lz4_frame_compress(msgpack.dumps(ex, use_bin_type=True)))      # TxtLmdb.__setitem__, data.py:163-166
    w.close()
    th, mx, mn = inp['conf_th'], inp['max_bb'], inp['min_bb']
    json.dump(inp['nbb'], open(os.path.join(img_dir, f'nbb_th{th}_max{mx}_min{mn}.json'), 'w'))
    name = f'feat_th{th}_max{mx}_min{mn}' + ('_compressed' if flavour == 'npz' else '')
    w = D.FlatDbWriter(os.path.join(img_dir, name))
    for f in inp['nbb']:
        d = {k: arr[f'img.{f}.{k}'] for k in ('features', 'norm_bb', 'conf')}
        if flavour == 'npz':
            buf = io.BytesIO()
            np.savez_compressed(buf, **d)
            w.put(f, buf.getvalue())
        else:
            w.put(f, msgpack.dumps(d, default=D.msgpack_numpy_encode, use_bin_type=True))
    w.close()
    return txt_dir, img_dir


def _check_batch(batch, arr, exp, prefix):
    for k, v in batch.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                key = f'{prefix}.{k}.{kk}'
                if vv is None:
                    assert key not in arr and exp[f'{k}.{kk}'] is None, key
                else:
                    want = arr[key]
                    assert tuple(vv.shape) == want.shape and str(vv.numpy().dtype) == str(want.dtype), key
                    np.testing.assert_array_equal(vv.numpy(), want, err_msg=key)
        else:
            assert v == exp[k], k
    assert {f'{k}.{kk}' for k, v in batch.items() if isinstance(v, dict) for kk in v} == \
        {k[len(prefix) + 1:] for k in arr.files if k.startswith(prefix + '.')} | {k for k, v in exp.items() if '.' in k and v is None}


@pytest.mark.parametrize('flavour', ['npz', 'msgpack'])
def test_readers_dataset_and_collate_equal_the_reference(tmp_path, golden_dir, flavour):
    g = json.load(open(os.path.join(golden_dir, 'g8_itm_data.json')))
    arr = np.load(os.path.join(golden_dir, 'g8_itm_data.npz'))
    inp, exp = g['inputs'], g['expected']
    txt_dir, img_dir = _write_db(str(tmp_path), inp, arr, flavour)
    txt_db = D.TxtTokDb(txt_dir, inp['max_txt_len'])
    img_db = D.DetectFeatDb(img_dir, inp['conf_th'], inp['max_bb'], inp['min_bb'], 36, flavour == 'npz')
    assert txt_db.ids == inp['kept']                                    # id2len filter (data.py:181-183), order kept
    # DetectFeatLmdb.__getitem__ / get_dump (data.py:98-121)
    f0 = exp[f'{flavour}.getitem.fname']
    feat, bb = img_db[f0]
    np.testing.assert_array_equal(feat.numpy(), arr[f'{flavour}.getitem.feat'])
    np.testing.assert_array_equal(bb.numpy(), arr[f'{flavour}.getitem.bb'])
    dump = img_db.get_dump(f0)
    for k in ('features', 'norm_bb', 'conf'):
        want = arr[f'{flavour}.get_dump.{k}']
        assert dump[k].dtype == want.dtype
        np.testing.assert_array_equal(dump[k], want)
    cases = {
        'eval': dict(nh=2, hn=False, caps=False),
        'train': dict(nh=2, hn=True, caps=False),
        'caps': dict(nh=1, hn=True, caps=True),
    }
    for cname, c in cases.items():
        e = exp[f'{flavour}.{cname}']
        ds = D.ItmFastDataset(txt_db, img_db, num_hard_negatives=c['nh'], img_meta=inp['img_meta'] if c['caps'] else None,
                              tokenizer=FakeTokenizer() if c['caps'] else None)
        if c['hn']:
            ds.new_epoch(inp['hn_img'], inp['hn_txt'])
        else:
            ds.new_epoch()
        assert list(ds.ids) == e['ids'] and [int(x) for x in ds.lens] == e['lens'] and list(ds.train_imgs) == e['train_imgs']
        assert sorted(ds.all_imgs) == e['all_imgs']
        batch = D.itm_fast_collate([ds[i] for i in e['items']])
        _check_batch(batch, arr, e, f'{flavour}.{cname}')
