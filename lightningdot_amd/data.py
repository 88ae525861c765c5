"""On-disk formats of the reference's text / image-feature databases and the ITM batch collate — the step BEFORE
the hot path.

    TxtTokDb          <- uniter_model/data/data.py:177-214  (TxtTokLmdb: id2len filter, rank-strided ids :185-186, meta.json, mappings)
    DetectFeatDb      <- uniter_model/data/data.py:44-125   (DetectFeatLmdb: db naming, nbb thresholding :30-33, npz / msgpack values,
                                                             fp16 -> fp32, first nbb regions)
    ItmFastDataset    <- dvl/data/itm.py:30-122             (new_epoch with per-item hard negatives, item layout)
    itm_fast_collate  <- dvl/data/itm.py:203-288            (batch dict consumed by BiEncoder.forward / train_step)
    EvalLoader        <- dvl/trainer.py:29-41               (build_dataloader(is_train=False): consecutive batches, moved to the device)

The reference keeps its records in LMDB environments.  The `lmdb`, `lz4` and `msgpack_numpy` packages are not part of this image, so
the CONTAINER is replaced and the VALUES are read as they are:
  * container: ``FlatDb`` = one ``<name>.bin`` with the raw LMDB values back to back + ``<name>.idx.json`` {key: [offset, length]};
    ``convert_lmdb`` (run once where the data and the `lmdb` package exist) copies every key/value of an LMDB environment into
    it byte for byte;
  * values: text records are ``lz4.frame.compress(msgpack.dumps(example))`` (data.py:160,165-166) -> decoded by the LZ4 frame
    decoder below (pure Python; a record is a few hundred bytes) + msgpack; feature records are ``np.savez_compressed`` archives
    (``compress=True``) or msgpack with msgpack_numpy's ndarray extension dicts -> decoded by numpy / the object hook below.
"""
import io
import json
import os
import struct
from collections import defaultdict
from typing import Dict, Iterable, List, Optional

import msgpack
import numpy as np
import torch
from torch.nn.utils.rnn import pad_sequence

# ---------------------------------------------------------------------------------------------------------------------------
# LZ4 frame format (the text DB's value compression).  Decoder: complete for what lz4.frame.compress emits (independent or
# linked blocks, optional content size / checksums, stored blocks).  Encoder: a small greedy one — enough to write fixtures and
# converted DBs that any LZ4 implementation reads.
# ---------------------------------------------------------------------------------------------------------------------------
_LZ4_MAGIC = 0x184D2204


def _lz4_block_decode(src: bytes, out: bytearray) -> None:
    """one LZ4 block appended to ``out`` (which already holds the history that linked blocks may reference)"""
    i, n = 0, len(src)
    while i < n:
        token = src[i]
        i += 1
        lit = token >> 4
        if lit == 15:
            while True:
                b = src[i]
                i += 1
                lit += b
                if b != 255:
                    break
        out += src[i:i + lit]
        i += lit
        if i >= n:
            break                                   # the last sequence has literals only
        offset = src[i] | (src[i + 1] << 8)
        i += 2
        if offset == 0 or offset > len(out):
            raise ValueError('corrupt LZ4 block: bad match offset')
        mlen = (token & 15) + 4
        if (token & 15) == 15:
            while True:
                b = src[i]
                i += 1
                mlen += b
                if b != 255:
                    break
        start = len(out) - offset
        if offset >= mlen:
            out += out[start:start + mlen]
        else:                                       # overlapping match: the pattern repeats
            for k in range(mlen):
                out.append(out[start + k])


_P1, _P2, _P3, _P4, _P5, _M32 = 2654435761, 2246822519, 3266489917, 668265263, 374761393, 0xffffffff


def xxh32(data: bytes, seed: int = 0) -> int:
    """XXH32 (the checksum of the LZ4 frame format: header, optional per-block and content checksums)"""
    n, i = len(data), 0
    rotl = lambda x, r: ((x << r) | (x >> (32 - r))) & _M32
    if n >= 16:
        v1, v2, v3, v4 = (seed + _P1 + _P2) & _M32, (seed + _P2) & _M32, seed & _M32, (seed - _P1) & _M32
        for a, b, c, d in struct.iter_unpack('<4I', data[:n - n % 16]):
            v1 = rotl((v1 + a * _P2) & _M32, 13) * _P1 & _M32
            v2 = rotl((v2 + b * _P2) & _M32, 13) * _P1 & _M32
            v3 = rotl((v3 + c * _P2) & _M32, 13) * _P1 & _M32
            v4 = rotl((v4 + d * _P2) & _M32, 13) * _P1 & _M32
        i = n - n % 16
        h = (rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18)) & _M32
    else:
        h = (seed + _P5) & _M32
    h = (h + n) & _M32
    while i + 4 <= n:
        h = rotl((h + struct.unpack_from('<I', data, i)[0] * _P3) & _M32, 17) * _P4 & _M32
        i += 4
    while i < n:
        h = rotl((h + data[i] * _P5) & _M32, 11) * _P1 & _M32
        i += 1
    h ^= h >> 15
    h = h * _P2 & _M32
    h ^= h >> 13
    h = h * _P3 & _M32
    return h ^ (h >> 16)


def lz4_frame_decompress(data: bytes) -> bytes:
    """One LZ4 frame -> its content.  Header, block and content checksums are verified when the frame carries them, the content size
    when it is stored (what ``lz4.frame.decompress`` does: a damaged record raises instead of decoding to garbage)."""
    data = bytes(data)
    if len(data) < 7 or struct.unpack_from('<I', data, 0)[0] != _LZ4_MAGIC:
        raise ValueError('not an LZ4 frame')
    flg, pos = data[4], 6
    if (flg >> 6) != 1:
        raise ValueError('unsupported LZ4 frame version')
    block_checksum, content_size, content_checksum, dict_id = flg & 0x10, flg & 0x08, flg & 0x04, flg & 0x01
    want_size = None
    if content_size:
        want_size = struct.unpack_from('<Q', data, pos)[0]
        pos += 8
    if dict_id:
        pos += 4
    if data[pos] != (xxh32(data[4:pos]) >> 8) & 0xff:
        raise ValueError('corrupt LZ4 frame: header checksum mismatch')
    pos += 1
    out = bytearray()
    while True:
        size = struct.unpack_from('<I', data, pos)[0]
        pos += 4
        if size == 0:
            break
        stored = size & 0x80000000
        size &= 0x7fffffff
        blk = data[pos:pos + size]
        if len(blk) != size:
            raise ValueError('truncated LZ4 frame')
        pos += size
        if block_checksum:
            if struct.unpack_from('<I', data, pos)[0] != xxh32(blk):
                raise ValueError('corrupt LZ4 frame: block checksum mismatch')
            pos += 4
        if stored:
            out += blk
        else:
            _lz4_block_decode(blk, out)
    if content_checksum:
        if struct.unpack_from('<I', data, pos)[0] != xxh32(bytes(out)):
            raise ValueError('corrupt LZ4 frame: content checksum mismatch')
        pos += 4
    if want_size is not None and want_size != len(out):
        raise ValueError('corrupt LZ4 frame: content size mismatch')
    return bytes(out)


def _lz4_block_encode(src: bytes) -> bytes:
    """greedy single-probe hash matcher (4-byte minimum match, the format's end-of-block rules respected)"""
    n = len(src)
    out = bytearray()
    table = {}
    anchor = i = 0
    limit = n - 12                                  # the last match must start >= 12 bytes before the end
    while i < limit:
        key = src[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 65535:
            m = 4
            while i + m < n - 5 and src[cand + m] == src[i + m]:
                m += 1
            lit = i - anchor
            token = (min(lit, 15) << 4) | min(m - 4, 15)
            out.append(token)
            if lit >= 15:
                r = lit - 15
                while r >= 255:
                    out.append(255)
                    r -= 255
                out.append(r)
            out += src[anchor:i]
            out += struct.pack('<H', i - cand)
            if m - 4 >= 15:
                r = m - 4 - 15
                while r >= 255:
                    out.append(255)
                    r -= 255
                out.append(r)
            i += m
            anchor = i
        else:
            i += 1
    lit = n - anchor
    out.append(min(lit, 15) << 4)
    if lit >= 15:
        r = lit - 15
        while r >= 255:
            out.append(255)
            r -= 255
        out.append(r)
    out += src[anchor:]
    return bytes(out)


def lz4_frame_compress(data: bytes) -> bytes:
    """one frame, independent 4 MiB-max blocks, no checksums (FLG 0x60, BD 0x70; the header checksum byte is xxh32(desc)>>8 — the
    two-byte descriptor used here is constant, so is its checksum 0x73)"""
    out = bytearray(struct.pack('<I', _LZ4_MAGIC) + b'\x60\x70\x73')
    for b0 in range(0, max(len(data), 1), 4 << 20):
        chunk = data[b0:b0 + (4 << 20)]
        if not chunk:
            break
        enc = _lz4_block_encode(chunk)
        if len(enc) < len(chunk):
            out += struct.pack('<I', len(enc)) + enc
        else:
            out += struct.pack('<I', len(chunk) | 0x80000000) + chunk
    out += struct.pack('<I', 0)
    return bytes(out)


# ---------------------------------------------------------------------------------------------------------------------------
# msgpack_numpy's ndarray encoding ({'nd': True, 'type': dtype.str, 'kind': '', 'shape': [...], 'data': raw bytes}; keys bytes or str)
# ---------------------------------------------------------------------------------------------------------------------------
def _np_object_hook(obj):
    nd = obj.get('nd', obj.get(b'nd'))
    if nd is True:
        dtype = obj.get('type', obj.get(b'type'))
        shape = obj.get('shape', obj.get(b'shape'))
        data = obj.get('data', obj.get(b'data'))
        if isinstance(dtype, bytes):
            dtype = dtype.decode()
        return np.frombuffer(data, dtype=np.dtype(dtype)).reshape(shape)
    return obj


def msgpack_numpy_encode(obj):
    """``default=`` hook producing msgpack_numpy's layout (fixtures / converted DBs)"""
    if isinstance(obj, np.ndarray):
        return {'nd': True, 'type': obj.dtype.str, 'kind': '', 'shape': list(obj.shape), 'data': obj.tobytes()}
    raise TypeError(type(obj))


# ---------------------------------------------------------------------------------------------------------------------------
# container
# ---------------------------------------------------------------------------------------------------------------------------
class FlatDb:
    """read-only key -> raw value bytes (stands where the reference holds ``lmdb.Environment.begin().get``)"""

    def __init__(self, prefix: str):
        with open(prefix + '.idx.json') as f:
            self.index = json.load(f)
        self._f = open(prefix + '.bin', 'rb')

    def get(self, key: str) -> Optional[bytes]:
        loc = self.index.get(key)
        if loc is None:
            return None
        self._f.seek(loc[0])
        return self._f.read(loc[1])

    def keys(self):
        return self.index.keys()

    def __contains__(self, key):
        return key in self.index

    def close(self):
        self._f.close()


class FlatDbWriter:
    def __init__(self, prefix: str):
        os.makedirs(os.path.dirname(prefix) or '.', exist_ok=True)
        self.prefix, self.index, self._f = prefix, {}, open(prefix + '.bin', 'wb')

    def put(self, key: str, value: bytes):
        self.index[key] = [self._f.tell(), len(value)]
        self._f.write(value)

    def close(self):
        self._f.close()
        with open(self.prefix + '.idx.json', 'w') as f:
            json.dump(self.index, f)


def convert_lmdb(lmdb_dir: str, out_prefix: str) -> int:
    """Offline, where the data lives: copy every key / value of an LMDB environment into a FlatDb, byte for byte.  Needs the
    ``lmdb`` package (not part of this image)."""
    import lmdb
    env = lmdb.open(lmdb_dir, readonly=True, create=False, lock=False, readahead=False)
    w = FlatDbWriter(out_prefix)
    n = 0
    with env.begin(buffers=True) as txn:
        for k, v in txn.cursor():
            w.put(bytes(k).decode('utf-8'), bytes(v))
            n += 1
    w.close()
    env.close()
    return n


# ---------------------------------------------------------------------------------------------------------------------------
# text side
# ---------------------------------------------------------------------------------------------------------------------------
class TxtTokDb:
    """uniter_model/data/data.py:177-214.  ``db_dir`` holds id2len.json, meta.json, txt2img.json, img2txts.json (as the reference's
    DB folders do) and the converted record store ``data.bin`` / ``data.idx.json``."""

    def __init__(self, db_dir: str, max_txt_len: int = 60, shard=None):
        """``shard=(rank, world)`` keeps ids[rank::world] like the reference does under horovod (data.py:185-186; its guard
        ``hvd.size() != hvd.local_size`` compares with the function object and is therefore always true, :36-41).  Default: all
        ids — here the training loader hands every rank its share of each batch (train_itm.default_train_loader), and striding
        the ids as well would shard twice."""
        self.id2len = json.load(open(f'{db_dir}/id2len.json'))
        if max_txt_len == -1:
            ids = list(self.id2len.keys())
        else:
            ids = [id_ for id_, len_ in self.id2len.items() if len_ <= max_txt_len]
        if shard is not None:
            rank, world = shard
            ids = ids[rank::world]
        self.ids = ids
        self.db_dir = db_dir
        self.db = FlatDb(os.path.join(db_dir, 'data'))
        meta = json.load(open(f'{db_dir}/meta.json'))
        self.cls_, self.sep, self.mask, self.v_range = meta['CLS'], meta['SEP'], meta['MASK'], meta['v_range']

    def __getitem__(self, id_):
        raw = self.db.get(id_)
        if raw is None:
            raise KeyError(id_)
        return msgpack.loads(lz4_frame_decompress(raw), raw=False)          # data.py:160

    def combine_inputs(self, *inputs):
        input_ids = [self.cls_]
        for ids in inputs:
            input_ids.extend(ids + [self.sep])
        return torch.tensor(input_ids)

    @property
    def txt2img(self):
        return json.load(open(f'{self.db_dir}/txt2img.json'))

    @property
    def img2txts(self):
        return json.load(open(f'{self.db_dir}/img2txts.json'))


# ---------------------------------------------------------------------------------------------------------------------------
# image side
# ---------------------------------------------------------------------------------------------------------------------------
def compute_num_bb(confs, conf_th, min_bb, max_bb):
    """data.py:30-33"""
    num_bb = max(min_bb, int((confs > conf_th).sum()))
    return min(max_bb, num_bb)


class DetectFeatDb:
    """uniter_model/data/data.py:44-125 — region features ('features' [n, 2048], 'norm_bb' [n, 6], 'conf' [n]), stored as fp16,
    served as fp32, truncated to the image's nbb regions."""

    def __init__(self, img_dir: str, conf_th=0.2, max_bb=100, min_bb=10, num_bb=36, compress=True):
        self.img_dir = img_dir
        self.conf_th, self.max_bb, self.min_bb, self.num_bb = conf_th, max_bb, min_bb, num_bb
        if conf_th == -1:
            db_name = f'feat_numbb{num_bb}'
            self.name2nbb = defaultdict(lambda: num_bb)
        else:
            db_name = f'feat_th{conf_th}_max{max_bb}_min{min_bb}'
            nbb = f'nbb_th{conf_th}_max{max_bb}_min{min_bb}.json'
            self.name2nbb = json.load(open(f'{img_dir}/{nbb}')) if os.path.exists(f'{img_dir}/{nbb}') else None
        self.compress = compress
        if compress:
            db_name += '_compressed'
        if self.name2nbb is None:
            db_name = 'all_compressed' if compress else 'all'
        self.db = FlatDb(os.path.join(img_dir, db_name))
        if self.name2nbb is None:
            self.name2nbb = self._compute_nbb()

    def _load(self, dump: bytes) -> Dict[str, np.ndarray]:
        if self.compress:
            with io.BytesIO(dump) as reader:
                z = np.load(reader, allow_pickle=True)
                return {k: z[k] for k in z.files}
        return msgpack.loads(dump, raw=False, object_hook=_np_object_hook)

    def _compute_nbb(self):
        fnames = json.loads(self.db.get('__keys__').decode('utf-8'))
        return {f: compute_num_bb(self._load(self.db.get(f))['conf'], self.conf_th, self.min_bb, self.max_bb) for f in fnames}

    def get_dump(self, file_name):
        d = self._load(self.db.get(file_name))
        nbb = self.name2nbb[file_name]
        return {k: (a.astype(np.float32) if a.dtype == np.float16 else a)[:nbb, ...] for k, a in d.items()}

    def __getitem__(self, file_name):
        d = self._load(self.db.get(file_name))
        nbb = self.name2nbb[file_name]
        img_feat = torch.tensor(np.asarray(d['features'][:nbb, :])).float()
        img_bb = torch.tensor(np.asarray(d['norm_bb'][:nbb, :])).float()
        return img_feat, img_bb

    def __contains__(self, file_name):
        return file_name in self.db


# ---------------------------------------------------------------------------------------------------------------------------
# dataset + collate
# ---------------------------------------------------------------------------------------------------------------------------
class ItmFastDataset(torch.utils.data.Dataset):
    """dvl/data/itm.py:30-122 (+ DetectFeatTxtTokDataset, data.py:216-246)."""

    def __init__(self, txt_db: TxtTokDb, img_db: DetectFeatDb, num_hard_negatives=0, img_meta=None, tokenizer=None):
        self.txt_db, self.img_db = txt_db, img_db
        self.ids = list(txt_db.ids)
        self.txt_lens = [txt_db.id2len[i] for i in self.ids]
        self.ids_2_idx = {idx: i for i, idx in enumerate(self.ids)}
        self._img_of = [self.txt_db[id_]['img_fname'] for id_ in self.ids]
        self.all_imgs = list(set(self._img_of))
        self.num_hard_negatives, self.img_meta, self.tokenizer = num_hard_negatives, img_meta, tokenizer
        self.train_imgs = self.neg_imgs = self.train_txts = self.neg_txts = None      # bound by new_epoch
        self.lens = [tl + self.img_db.name2nbb[f] for tl, f in zip(self.txt_lens, self._img_of)]

    def __len__(self):
        return len(self.ids)

    def new_epoch(self, hard_negatives_img=None, hard_negatives_txt=None):
        """itm.py:51-68: (re)bind every item to its image and, with mining results, to its nh hard negatives of each kind"""
        self.lens, self.train_imgs, self.neg_imgs, self.train_txts, self.neg_txts = [], [], [], [], []
        nh = self.num_hard_negatives
        for id_, tl, img_fname in zip(self.ids, self.txt_lens, self._img_of):
            self.train_imgs.append(img_fname)
            self.train_txts.append(id_)
            if hard_negatives_img is not None and nh > 0:
                self.neg_imgs.append(hard_negatives_img[id_][:nh])
                self.neg_txts.append(hard_negatives_txt[img_fname][:nh])
            else:
                self.neg_imgs.append(None)
                self.neg_txts.append(None)
            self.lens.append(tl + self.img_db.name2nbb[img_fname])

    def _get_img_feat(self, fname):
        img_feat, bb = self.img_db[fname]
        img_bb = torch.cat([bb, bb[:, 4:5] * bb[:, 5:]], dim=-1)       # data.py:243: [x1,y1,x2,y2,w,h] + w*h
        return img_feat, img_bb, img_feat.size(0)

    def _captions(self, fname, like):
        tok = self.tokenizer
        pieces = [tok.encode(c, add_special_tokens=False) + [tok.sep_token_id] for c in self.img_meta[fname]['caption_multiple']]
        ids = torch.tensor([tok.cls_token_id] + sum(pieces, []), dtype=like.dtype)
        return ids, torch.ones(len(ids), dtype=torch.long)

    def __getitem__(self, i):
        if self.train_imgs is None:
            self.new_epoch()
        example = self.txt_db[self.ids[i]]
        img_fname, hard_neg_imgs, hard_neg_txts = self.train_imgs[i], self.neg_imgs[i], self.neg_txts[i]
        img_input_ids = torch.tensor([101]).long()
        img_feat, img_pos_feat, num_bb = self._get_img_feat(img_fname)
        attn_masks_img = torch.ones(num_bb + 1, dtype=torch.long)
        input_ids = self.txt_db.combine_inputs(example['input_ids'])
        attn_masks = torch.ones(len(input_ids), dtype=torch.long)
        if hard_neg_imgs is not None:
            neg_imgs = {'img_input_ids': [], 'img_feat': [], 'img_pos_feat': [], 'num_bb': [], 'attn_masks_img': [],
                        'caption_ids': [], 'attn_masks_captions': []}
            for neg_id in hard_neg_imgs:
                neg_imgs['img_input_ids'].append(torch.tensor([101]).long())
                f, p, nb = self._get_img_feat(neg_id)
                neg_imgs['img_feat'].append(f)
                neg_imgs['img_pos_feat'].append(p)
                neg_imgs['num_bb'].append(nb)
                neg_imgs['attn_masks_img'].append(torch.ones(nb + 1, dtype=torch.long))
                if self.img_meta is not None:
                    c, m = self._captions(neg_id, input_ids)
                    neg_imgs['caption_ids'].append(c)
                    neg_imgs['attn_masks_captions'].append(m)
            neg_txts = {'input_ids': [], 'position_ids': [], 'attention_mask': []}
            for neg_id in hard_neg_txts:
                ids_ei = self.txt_db.combine_inputs(self.txt_db[neg_id]['input_ids'])
                neg_txts['input_ids'].append(ids_ei)
                neg_txts['attention_mask'].append(torch.ones(len(ids_ei), dtype=torch.long))
        else:
            neg_imgs = neg_txts = None
        if self.img_meta is not None:
            caption_ids, attn_masks_captions = self._captions(img_fname, input_ids)
        else:
            caption_ids = attn_masks_captions = None
        return (input_ids, img_feat, img_pos_feat, img_input_ids, attn_masks, attn_masks_img, self.ids[i], img_fname, neg_imgs,
                neg_txts, caption_ids, attn_masks_captions)


def pad_tensors(tensors: List[torch.Tensor], lens=None, pad=0):
    """itm.py:13-26 — B x [T, ...] -> [B, max T, ...]"""
    if lens is None:
        lens = [t.size(0) for t in tensors]
    out = torch.zeros(len(tensors), max(lens), tensors[0].size(-1), dtype=tensors[0].dtype)
    if pad:
        out.fill_(pad)
    for i, (t, l) in enumerate(zip(tensors, lens)):
        out[i, :l, ...] = t
    return out


def itm_fast_collate(inputs: Iterable):
    """dvl/data/itm.py:203-288: the hard negatives of every item are appended AFTER the ``sample_size`` positives (:283-284)."""
    cols = list(map(list, zip(*inputs)))
    (input_ids, img_feats, img_pos_feats, img_input_ids, attn_masks_text, attn_masks_img, idx, img_fname, neg_imgs, neg_txts,
     caption_ids, attn_masks_captions) = cols
    bs = len(input_ids)
    chain = lambda key, src: [x for n in src for x in n[key]]
    if None not in neg_imgs:
        num_bbs_neg, img_feats_neg = chain('num_bb', neg_imgs), chain('img_feat', neg_imgs)
        img_input_ids_neg, img_pos_feat_neg = chain('img_input_ids', neg_imgs), chain('img_pos_feat', neg_imgs)
        attn_masks_img_neg = chain('attn_masks_img', neg_imgs)
        caption_ids_neg, attn_masks_captions_neg = chain('caption_ids', neg_imgs), chain('attn_masks_captions', neg_imgs)
        input_ids_neg, attn_masks_text_neg = chain('input_ids', neg_txts), chain('attention_mask', neg_txts)
    else:
        num_bbs_neg, img_feats_neg, img_input_ids_neg, img_pos_feat_neg, attn_masks_img_neg = [], [], [], [], []
        caption_ids_neg, attn_masks_captions_neg, input_ids_neg, attn_masks_text_neg = [], [], [], []
    input_ids = pad_sequence(input_ids + input_ids_neg, batch_first=True, padding_value=0)
    position_ids = torch.arange(0, input_ids.size(1), dtype=torch.long).unsqueeze(0)
    has_caps = caption_ids[0] is not None
    captions_ids = pad_sequence(caption_ids + caption_ids_neg, batch_first=True, padding_value=0) if has_caps else None
    position_ids_captions = torch.arange(0, captions_ids.size(1), dtype=torch.long).unsqueeze(0) if has_caps else None
    num_bbs = [f.size(0) for f in img_feats] + num_bbs_neg
    img_feat = pad_tensors(img_feats + img_feats_neg, num_bbs)
    img_pos_feat = pad_tensors(img_pos_feats + img_pos_feat_neg, num_bbs)
    img_input_ids = pad_sequence(img_input_ids + img_input_ids_neg, batch_first=True, padding_value=0)
    img_position_ids = torch.arange(0, img_input_ids.size(1), dtype=torch.long).unsqueeze(0)
    attn_masks_text = pad_sequence(attn_masks_text + attn_masks_text_neg, batch_first=True, padding_value=0)
    attn_masks_captions = (pad_sequence(attn_masks_captions + attn_masks_captions_neg, batch_first=True, padding_value=0)
                           if has_caps else None)
    attn_masks_img = pad_sequence(attn_masks_img + attn_masks_img_neg, batch_first=True, padding_value=0)
    out_size = attn_masks_img.size(1)
    gather_index = torch.arange(0, out_size, dtype=torch.long).unsqueeze(0).repeat(len(num_bbs), 1)   # data.py:280-287
    return {
        'txts': {'input_ids': input_ids, 'position_ids': position_ids, 'attention_mask': attn_masks_text, 'img_feat': None,
                 'img_pos_feat': None, 'img_masks': None, 'gather_index': None},
        'imgs': {'input_ids': img_input_ids, 'position_ids': img_position_ids, 'attention_mask': attn_masks_img,
                 'img_feat': img_feat, 'img_pos_feat': img_pos_feat, 'img_masks': None, 'gather_index': gather_index},
        'caps': {'input_ids': captions_ids, 'position_ids': position_ids_captions, 'attention_mask': attn_masks_captions,
                 'img_feat': None, 'img_pos_feat': None, 'img_masks': None, 'gather_index': None},
        'sample_size': bs, 'pos_ctx_indices': list(range(bs)), 'neg_ctx_indices': list(range(bs, len(num_bbs))),
        'txt_index': idx, 'img_fname': img_fname}


def batch_to_device(batch: dict, device):
    """moves the tensors of a collated batch (the reference's PrefetchLoader does this on a side stream, loader.py:90-129)"""
    out = {}
    for k, v in batch.items():
        if isinstance(v, dict):
            out[k] = {kk: (vv.to(device, non_blocking=True) if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
        else:
            out[k] = v.to(device, non_blocking=True) if torch.is_tensor(v) else v
    return out


class EvalLoader:
    """Re-iterable evaluation-style loader (dvl/trainer.py:29-41 without the worker processes): consecutive items, collated and moved
    to the device ONE batch at a time — the towers consume a batch before the next one is read, so a pass over the Flickr / COCO
    training sets for hard-negative mining never holds more than one batch of region features."""

    def __init__(self, dataset, batch_size: int, device):
        self.dataset, self.batch_size, self.device = dataset, int(batch_size), device

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        ds, bs = self.dataset, self.batch_size
        for b0 in range(0, len(ds), bs):
            yield batch_to_device(itm_fast_collate([ds[i] for i in range(b0, min(b0 + bs, len(ds)))]), self.device)
