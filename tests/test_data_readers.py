"""CPU tests of the on-disk readers and the ITM collate: LZ4 frame codec, the reference's value formats
(lz4+msgpack text records, npz / msgpack_numpy fp16 region features), db naming and nbb thresholding, new_epoch with hard
negatives, batch layout.  The DB is written by tools/make_db_fixture.py (committed) into tmp_path."""
import importlib.util
import os
from struct import error as struct_error

import numpy as np
import pytest
import torch

from lightningdot_amd import data as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture(tmp_path, **kw):
    spec = importlib.util.spec_from_file_location('make_db_fixture', os.path.join(ROOT, 'tools', 'make_db_fixture.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.make(str(tmp_path), **kw)


def test_lz4_frame_roundtrip_and_known_frames():
    rng = np.random.default_rng(0)
    cases = [b'', b'a', b'abc' * 1000, bytes(rng.integers(0, 256, 5000, dtype=np.uint8)),
             (b'the quick brown fox ' * 40 + bytes(rng.integers(0, 4, 3000, dtype=np.uint8))) * 3, b'\x00' * 100000]
    for c in cases:
        enc = D.lz4_frame_compress(c)
        assert D.lz4_frame_decompress(enc) == c
        if len(c) > 2000 and len(set(c)) < 200:
            assert len(enc) < len(c)                     # the matcher really compresses repetitive input
    # a frame assembled by hand from the LZ4 frame / block specification (not by this module's encoder): FLG 0x60, BD 0x70, header
    # checksum 0x73; one 16-byte block = [token 0x6F: 6 literals "hello ", match offset 6, length 4 + 15 + 5 = 24] + [token 0x50: the
    # 5 last literals "hello"]; end mark
    ref = bytes.fromhex('04224d18' '607073' '10000000' '6f' '68656c6c6f20' '0600' '05' '50' '68656c6c6f' '00000000')
    assert D.lz4_frame_decompress(ref) == b'hello ' * 5 + b'hello'
    # the same payload as a STORED block (high bit of the block size) and with the content-size field present (FLG 0x68)
    payload = b'hello ' * 5 + b'hello'
    desc = bytes.fromhex('6870') + (35).to_bytes(8, 'little')
    hc = bytes([(D.xxh32(desc) >> 8) & 0xff])                  # header checksum = second byte of XXH32(descriptor) (xxh32 is pinned below)
    stored = bytes.fromhex('04224d18') + desc + hc + (35 | 0x80000000).to_bytes(4, 'little') + payload + bytes(4)
    assert D.lz4_frame_decompress(stored) == payload
    with pytest.raises(ValueError):
        D.lz4_frame_decompress(b'not a frame at all')


def test_lz4_frames_written_by_liblz4(golden_dir):
    """Frames produced by the REAL liblz4 (oracle/gen_lz4_vectors.py: LZ4F_compressFrame with explicit preferences — python-lz4's
    defaults, linked 64 KiB blocks with matches across block borders, independent blocks, block + content checksums, HC level, stored
    blocks, empty input) decode to their payloads; a flipped bit anywhere in a checksummed frame is detected."""
    import base64
    import hashlib
    import json
    g = json.load(open(os.path.join(golden_dir, 'lz4_frames.json')))
    assert len(g['vectors']) >= 8
    for v in g['vectors']:
        frame = base64.b64decode(v['frame_b64'])
        out = D.lz4_frame_decompress(frame)
        assert len(out) == v['length'] and hashlib.sha256(out).hexdigest() == v['sha256'], v['name']
        if 'payload_b64' in v:
            assert out == base64.b64decode(v['payload_b64'])
        if v['prefs'].get('content_checksum') and v['length']:
            bad = bytearray(frame)
            bad[len(bad) // 2] ^= 0x10
            with pytest.raises((ValueError, IndexError, struct_error)):
                D.lz4_frame_decompress(bytes(bad))
    rec = next(v for v in g['vectors'] if v['name'] == 'record_default')
    import msgpack
    ex = msgpack.loads(D.lz4_frame_decompress(base64.b64decode(rec['frame_b64'])), raw=False)
    assert ex['img_fname'] == 'flickr30k_000000001234.npz' and len(ex['input_ids']) == 17


def test_xxh32_known_answers_and_library():
    # published XXH32 test values (xxHash repository's sanity checks): empty input, seed 0 / seed 0x9E3779B1
    assert D.xxh32(b'') == 0x02CC5D05 and D.xxh32(b'', 0x9E3779B1) == 0x36B78AE7
    xxhash = pytest.importorskip('xxhash')
    rng = np.random.default_rng(1)
    for n in (1, 3, 4, 15, 16, 17, 31, 32, 33, 100, 4099):
        b = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        assert D.xxh32(b) == xxhash.xxh32(b).intdigest() and D.xxh32(b, 77) == xxhash.xxh32(b, seed=77).intdigest()


def test_frames_written_here_are_read_by_liblz4():
    """the other direction, live, where the system's liblz4 is present (it is in this image): what lz4_frame_compress writes
    (fixtures, converted DBs) is a valid frame for the real library"""
    import ctypes.util
    if ctypes.util.find_library('lz4') is None:
        pytest.skip('no system liblz4')
    import importlib.util
    spec = importlib.util.spec_from_file_location('gen_lz4_vectors', os.path.join(ROOT, 'oracle', 'gen_lz4_vectors.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = mod.liblz4()
    for name, payload in mod.payloads().items():
        assert mod.decompress(lib, D.lz4_frame_compress(payload), len(payload) + 1) == payload, name


def test_text_db_reader(tmp_path):
    examples, feats, nbb = _fixture(tmp_path)
    db = D.TxtTokDb(str(tmp_path / 'txt_db'), max_txt_len=60)
    assert '2' not in db.ids and len(db.ids) == len(examples) - 1       # the 65-token caption is filtered (data.py:181-183)
    assert len(D.TxtTokDb(str(tmp_path / 'txt_db'), max_txt_len=-1).ids) == len(examples)
    for tid in db.ids:
        assert db[tid] == examples[tid]
    assert (db.cls_, db.sep, db.mask) == (101, 102, 103)
    assert db.combine_inputs([5, 6], [7]).tolist() == [101, 5, 6, 102, 7, 102]
    assert db.txt2img['3'] == 'flickr30k_000000000001.npz' and db.img2txts['flickr30k_000000000001.npz'] == ['2', '3']
    with pytest.raises(KeyError):
        db['nope']


@pytest.mark.parametrize('flavour', ['thresholded_npz', 'all_msgpack'])
def test_feature_db_reader(tmp_path, flavour):
    examples, feats, nbb = _fixture(tmp_path)
    if flavour == 'thresholded_npz':
        db = D.DetectFeatDb(str(tmp_path / 'img_db'), conf_th=0.2, max_bb=100, min_bb=10, num_bb=36, compress=True)
    else:                                                               # no precomputed nbb -> 'all' DB, nbb computed from conf
        os.remove(tmp_path / 'img_db' / 'nbb_th0.2_max100_min10.json')
        db = D.DetectFeatDb(str(tmp_path / 'img_db'), conf_th=0.2, max_bb=100, min_bb=10, num_bb=36, compress=False)
    assert dict(db.name2nbb) == nbb
    for f, d in feats.items():
        feat, bb = db[f]
        n = nbb[f]
        assert feat.dtype == torch.float32 and feat.shape == (n, 2048) and bb.shape == (n, 6)
        np.testing.assert_array_equal(feat.numpy(), d['features'][:n].astype(np.float32))      # fp16 -> fp32 exactly
        np.testing.assert_array_equal(bb.numpy(), d['norm_bb'][:n].astype(np.float32))
        dump = db.get_dump(f)
        assert set(dump) == {'features', 'norm_bb', 'conf'} and dump['conf'].dtype == np.float32 and len(dump['conf']) == n
        assert f in db
    assert 'missing.npz' not in db
    assert D.compute_num_bb(np.array([0.9, 0.5, 0.1]), 0.2, 10, 100) == 10 and D.compute_num_bb(np.ones(300), 0.2, 10, 100) == 100


def test_dataset_collate_and_hard_negatives(tmp_path):
    examples, feats, nbb = _fixture(tmp_path)
    txt = D.TxtTokDb(str(tmp_path / 'txt_db'), max_txt_len=60)
    img = D.DetectFeatDb(str(tmp_path / 'img_db'))
    ds = D.ItmFastDataset(txt, img, num_hard_negatives=2)
    ds.new_epoch()
    b = D.itm_fast_collate([ds[i] for i in range(4)])
    assert b['sample_size'] == 4 and b['neg_ctx_indices'] == [] and b['pos_ctx_indices'] == [0, 1, 2, 3]
    assert b['txt_index'] == ds.ids[:4] and b['img_fname'] == [examples[t]['img_fname'] for t in ds.ids[:4]]
    lens = [len(examples[t]['input_ids']) + 2 for t in ds.ids[:4]]
    assert b['txts']['input_ids'].shape == (4, max(lens)) and b['txts']['attention_mask'].sum(1).tolist() == lens
    assert (b['txts']['input_ids'][:, 0] == 101).all()
    nb = [nbb[f] for f in b['img_fname']]
    assert b['imgs']['img_feat'].shape == (4, max(nb), 2048) and b['imgs']['img_pos_feat'].shape == (4, max(nb), 7)
    assert b['imgs']['attention_mask'].sum(1).tolist() == [n + 1 for n in nb]
    assert b['imgs']['gather_index'].shape == (4, max(nb) + 1) and b['caps']['input_ids'] is None
    # 7th position feature = w * h of the normalised box (data.py:243)
    f0 = feats[b['img_fname'][0]]['norm_bb'][:nb[0]].astype(np.float32)
    np.testing.assert_allclose(b['imgs']['img_pos_feat'][0, :nb[0], 6].numpy(), f0[:, 4] * f0[:, 5], rtol=1e-6)
    # with mining results: nh negatives of each kind per item, appended AFTER the positives (itm.py:283-284)
    imgs, ids = list(feats), ds.ids
    hn_img = {t: [f for f in imgs if f != examples[t]['img_fname']][:3] for t in ids}
    hn_txt = {f: [t for t in ids if examples[t]['img_fname'] != f][:3] for f in imgs}
    ds.new_epoch(hn_img, hn_txt)
    b = D.itm_fast_collate([ds[i] for i in range(3)])
    assert b['sample_size'] == 3 and b['neg_ctx_indices'] == list(range(3, 9))
    assert b['txts']['input_ids'].shape[0] == 9 and b['imgs']['img_feat'].shape[0] == 9
    first_neg_txt = hn_txt[examples[ids[0]]['img_fname']][0]
    want = [101] + examples[first_neg_txt]['input_ids'] + [102]
    assert b['txts']['input_ids'][3, :len(want)].tolist() == want
    first_neg_img = hn_img[ids[0]][0]
    np.testing.assert_array_equal(b['imgs']['img_feat'][3, :nbb[first_neg_img]].numpy(),
                                  feats[first_neg_img]['features'][:nbb[first_neg_img]].astype(np.float32))
    # the batch drives the towers (layout identical to the synthetic generator's)
    from lightningdot_amd.synthetic import synthetic_itm_batches
    syn, _ = synthetic_itm_batches(4, caps_per_img=1, batch_size=4, num_hard_negatives=2)
    assert set(syn[0]) == set(b) and all(set(syn[0][k]) == set(b[k]) for k in ('txts', 'imgs', 'caps'))

