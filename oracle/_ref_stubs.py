"""Test infrastructure only.  sys.modules stubs that let the *reference* Python package at
/root/reference import inside THIS container (faiss / apex / horovod / lmdb / ... are not installed).
Used only by oracle/gen_golden.py to emit golden vectors; nothing here ships to the GPU box as product.

The numpy ``IndexFlatIP`` stand-in below restates the documented semantics of faiss-cpu==1.6.3
``IndexFlatIP`` (exact fp32 inner product, top-k descending, label -1 / score -FLT_MAX padding) because the
real library is third-party and un-vendored -> "parity unpinned" at that boundary.
"""
import sys
import types

import numpy as np
import torch

REF_ROOT = '/root/reference'


class _NumpyIndexFlatIP:
    def __init__(self, d):
        self.d = d
        self.x = np.zeros((0, d), dtype=np.float32)

    @property
    def ntotal(self):
        return self.x.shape[0]

    def add(self, v):
        v = np.ascontiguousarray(v, dtype=np.float32)
        assert v.ndim == 2 and v.shape[1] == self.d
        self.x = np.concatenate([self.x, v], axis=0)

    def search(self, q, k):
        q = np.ascontiguousarray(q, dtype=np.float32)
        s = q @ self.x.T
        nq, n = s.shape
        scores = np.full((nq, k), -np.finfo(np.float32).max, dtype=np.float32)
        labels = np.full((nq, k), -1, dtype=np.int64)
        kk = min(k, n)
        # descending by score, ties -> lowest index (stable sort on -s)
        order = np.argsort(-s, axis=1, kind='stable')[:, :kk]
        labels[:, :kk] = order
        scores[:, :kk] = np.take_along_axis(s, order, axis=1)
        return scores, labels


class FakeLmdb:
    """In-memory stand-in for the `lmdb` package (lmdb==0.97, DVL.yml:101; not installed): ``open(path)`` serves the key/value dict
    registered for that path — exactly the calls the reference makes (uniter_model/data/data.py:73-76,93,105,119,124,143-146,160:
    ``lmdb.open(...)``, ``env.begin(buffers=True)``, ``txn.get(key_bytes)``, ``env.close()``)."""
    registry = {}

    class _Txn:
        def __init__(self, kv):
            self.kv = kv

        def get(self, key=None, **kw):
            v = self.kv.get(bytes(key))
            return None if v is None else memoryview(v)          # buffers=True hands out buffer objects

    class _Env:
        def __init__(self, kv):
            self.kv = kv

        def begin(self, **kw):
            return FakeLmdb._Txn(self.kv)

        def close(self):
            pass

    @classmethod
    def open(cls, path, **kw):
        import os
        return cls._Env(cls.registry[os.path.normpath(path)])


def msgpack_numpy_patch():
    """Restatement of what ``msgpack_numpy.patch()`` (msgpack-numpy==0.4.6.post0, DVL.yml:109; not installed) does for the calls the
    reference makes: ``msgpack.loads`` decodes that package's ndarray dicts {nd: True, type: dtype.str, kind: '', shape, data} (keys
    bytes or str).  Third-party format -> "parity unpinned" at this boundary, like faiss."""
    import msgpack

    def hook(obj):
        nd = obj.get('nd', obj.get(b'nd'))
        if nd is True:
            t = obj.get('type', obj.get(b'type'))
            t = t.decode() if isinstance(t, bytes) else t
            return np.frombuffer(obj.get('data', obj.get(b'data')), dtype=np.dtype(t)).reshape(obj.get('shape', obj.get(b'shape')))
        return obj

    if getattr(msgpack.loads, '_ldot_np', False):
        return
    orig = msgpack.unpackb

    def loads(packed, **kw):
        kw.setdefault('object_hook', hook)
        return orig(bytes(packed), **kw)
    loads._ldot_np = True
    msgpack.loads = msgpack.unpackb = loads


def install():
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)

    # import the heavy third-party packages BEFORE any stub exists (accelerate probes find_spec('tensorboardX'))
    import transformers
    from transformers import BertModel, BertConfig, BertPreTrainedModel  # noqa: F401

    def mod(name, **attrs):
        import importlib.machinery
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    # apex
    mod('apex')
    mod('apex.normalization')
    mod('apex.normalization.fused_layer_norm', FusedLayerNorm=torch.nn.LayerNorm)
    mod('apex.amp')
    sys.modules['apex'].amp = sys.modules['apex.amp']
    # horovod
    hvd = mod('horovod.torch', rank=lambda: 0, size=lambda: 1, local_rank=lambda: 0,
              local_size=lambda: 1, init=lambda: None)
    mod('horovod', torch=hvd)
    # faiss
    mod('faiss', IndexFlatIP=_NumpyIndexFlatIP)
    # data deps
    mod('lmdb', open=FakeLmdb.open)
    mod('lz4')
    mod('lz4.frame', compress=lambda b: b, decompress=lambda b: b)
    sys.modules['lz4'].frame = sys.modules['lz4.frame']
    mod('msgpack_numpy', patch=msgpack_numpy_patch)
    mod('toolz')
    mod('toolz.sandbox', unzip=lambda seq: zip(*seq))
    def _partition_all(n, seq):
        seq = list(seq)
        return [tuple(seq[i:i + n]) for i in range(0, len(seq), n)]
    mod('cytoolz', concat=lambda seqs: [x for s in seqs for x in s], curry=lambda f: f,
        partition_all=_partition_all)
    mod('tensorboardX', SummaryWriter=object)
    # transformers API drift (reference pins 2.3.0)
    import transformers.optimization as topt
    if not hasattr(topt, 'AdamW'):
        topt.AdamW = torch.optim.AdamW
    if 'transformers.tokenization_bert' not in sys.modules:
        mod('transformers.tokenization_bert', BertTokenizer=getattr(transformers, 'BertTokenizer', object))


def patch_bert_encoder(ref_be):
    """Lets the reference's ``BertEncoder`` (dvl/models/bi_encoder.py:76-128, written against transformers==2.3.0) construct under the
    installed transformers: its ``self.init_weights()`` call is routed through ``post_init()`` (which newer PreTrainedModels need for
    their bookkeeping and which then runs the real ``init_weights``).  Configs must carry ``return_dict=False`` (the reference unpacks
    BertModel's output as a tuple, :110-119).  The class body — forward, [CLS] pooling, encode_proj — is the reference's own."""
    from transformers import PreTrainedModel
    orig = PreTrainedModel.init_weights

    def shim(self):
        if getattr(self, '_ldot_in_post', False):
            return orig(self)
        self._ldot_in_post = True
        try:
            self.post_init()
        finally:
            self._ldot_in_post = False
    ref_be.BertEncoder.init_weights = shim
