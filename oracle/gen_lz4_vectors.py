#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — LZ4 frame vectors from the REAL liblz4 (the C library behind the reference's ``lz4.frame``,
lz4==2.1.9 / lz4-c 1.9.2 in DVL.yml:20,102; the image ships the system's liblz4.so.1), driven through ctypes.

    python oracle/gen_lz4_vectors.py        # rewrites tests/golden/lz4_frames.json

Each vector = a frame produced by LZ4F_compressFrame with explicit preferences + the payload (or its sha256 when large).
tests/test_data_readers.py feeds the frames to lightningdot_amd.data.lz4_frame_decompress (pure Python) — and, where liblz4 is
present, also checks the other direction live (frames written by the build's encoder are read back by liblz4).
"""
import base64
import ctypes
import hashlib
import json
import os

import msgpack
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'lz4_frames.json')


class FrameInfo(ctypes.Structure):          # lz4frame.h: LZ4F_frameInfo_t
    _fields_ = [('blockSizeID', ctypes.c_int), ('blockMode', ctypes.c_int), ('contentChecksumFlag', ctypes.c_int),
                ('frameType', ctypes.c_int), ('contentSize', ctypes.c_ulonglong), ('dictID', ctypes.c_uint),
                ('blockChecksumFlag', ctypes.c_int)]


class Preferences(ctypes.Structure):        # lz4frame.h: LZ4F_preferences_t
    _fields_ = [('frameInfo', FrameInfo), ('compressionLevel', ctypes.c_int), ('autoFlush', ctypes.c_uint),
                ('favorDecSpeed', ctypes.c_uint), ('reserved', ctypes.c_uint * 3)]


def liblz4():
    lib = ctypes.CDLL('liblz4.so.1')
    lib.LZ4F_compressFrameBound.restype = ctypes.c_size_t
    lib.LZ4F_compressFrameBound.argtypes = [ctypes.c_size_t, ctypes.POINTER(Preferences)]
    lib.LZ4F_compressFrame.restype = ctypes.c_size_t
    lib.LZ4F_compressFrame.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Preferences)]
    lib.LZ4F_isError.restype = ctypes.c_uint
    lib.LZ4F_isError.argtypes = [ctypes.c_size_t]
    lib.LZ4F_createDecompressionContext.restype = ctypes.c_size_t
    lib.LZ4F_createDecompressionContext.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    lib.LZ4F_freeDecompressionContext.argtypes = [ctypes.c_void_p]
    lib.LZ4F_decompress.restype = ctypes.c_size_t
    lib.LZ4F_decompress.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p,
                                    ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    lib.LZ4_versionNumber.restype = ctypes.c_int
    return lib


def compress(lib, data: bytes, block_size_id=0, independent=False, content_checksum=False, block_checksum=False, store_size=False,
             level=0) -> bytes:
    p = Preferences()
    p.frameInfo.blockSizeID = block_size_id
    p.frameInfo.blockMode = 1 if independent else 0
    p.frameInfo.contentChecksumFlag = int(content_checksum)
    p.frameInfo.blockChecksumFlag = int(block_checksum)
    p.frameInfo.contentSize = len(data) if store_size else 0
    p.compressionLevel = level
    cap = lib.LZ4F_compressFrameBound(len(data), ctypes.byref(p))
    dst = ctypes.create_string_buffer(cap)
    n = lib.LZ4F_compressFrame(dst, cap, data, len(data), ctypes.byref(p))
    assert not lib.LZ4F_isError(n)
    return dst.raw[:n]


def decompress(lib, frame: bytes, max_out: int) -> bytes:
    ctx = ctypes.c_void_p()
    assert not lib.LZ4F_isError(lib.LZ4F_createDecompressionContext(ctypes.byref(ctx), 100))
    out = bytearray()
    src = ctypes.create_string_buffer(frame, len(frame))
    pos = 0
    buf = ctypes.create_string_buffer(1 << 16)
    try:
        while pos < len(frame):
            dn, sn = ctypes.c_size_t(len(buf)), ctypes.c_size_t(len(frame) - pos)
            r = lib.LZ4F_decompress(ctx, buf, ctypes.byref(dn), ctypes.byref(src, pos), ctypes.byref(sn), None)
            if lib.LZ4F_isError(r):
                raise ValueError('liblz4 rejects the frame')
            out += buf.raw[:dn.value]
            pos += sn.value
            if r == 0 and sn.value == 0 and dn.value == 0:
                break
            assert len(out) <= max_out
    finally:
        lib.LZ4F_freeDecompressionContext(ctx)
    return bytes(out)


def payloads():
    rng = np.random.default_rng(42)
    vocab = [bytes(rng.integers(97, 123, int(rng.integers(2, 9)), dtype=np.uint8)) for _ in range(400)]
    words = b' '.join(vocab[int(i)] for i in rng.integers(0, 400, 8000))[:40000]
    record = msgpack.dumps({'id': '1234', 'dataset': 'flickr30k', 'split': 'test', 'sent': 'two dogs play in the snow near a fence',
                            'sent_id': 1234, 'img_fname': 'flickr30k_000000001234.npz', 'image_id': 1234,
                            'input_ids': [int(t) for t in rng.integers(1000, 28000, 17)]}, use_bin_type=True)
    return {
        'record': record,                                                   # what TxtLmdb.__setitem__ compresses (data.py:163-166)
        'three_copies_120k': words * 3,                                     # matches 40 000 bytes back, across the 64 KiB block borders
        'random_9k': bytes(rng.integers(0, 256, 9000, dtype=np.uint8)),     # incompressible -> stored block
        'zeros_300k': b'\x00' * 300000,                                     # long overlapping matches, several blocks
        'empty': b'',
        'short': b'abc',
    }


def main():
    lib = liblz4()
    pay = payloads()
    spec = [
        # name, payload, kwargs — 'record_default' are python-lz4's own defaults: linked blocks + content size (lz4.frame.compress)
        ('record_default', 'record', dict(store_size=True)),
        ('linked_64k_blocks', 'three_copies_120k', dict(block_size_id=4)),
        ('linked_both_checksums_hc', 'three_copies_120k', dict(block_size_id=4, content_checksum=True, block_checksum=True, level=9,
                                                               store_size=True)),
        ('independent_both_checksums', 'three_copies_120k', dict(block_size_id=4, independent=True, content_checksum=True,
                                                                 block_checksum=True)),
        ('stored_block_content_checksum', 'random_9k', dict(content_checksum=True)),
        ('zeros_linked_block_checksum', 'zeros_300k', dict(block_size_id=4, block_checksum=True)),
        ('empty_content_checksum', 'empty', dict(content_checksum=True)),
        ('short_default', 'short', dict()),
    ]
    vectors = []
    for name, pname, kw in spec:
        data = pay[pname]
        frame = compress(lib, data, **kw)
        assert decompress(lib, frame, len(data) + 1) == data
        v = dict(name=name, prefs=kw, frame_b64=base64.b64encode(frame).decode(), length=len(data),
                 sha256=hashlib.sha256(data).hexdigest())
        if len(data) <= 4096:
            v['payload_b64'] = base64.b64encode(data).decode()
        vectors.append(v)
        print(name, len(data), '->', len(frame))
    json.dump(dict(liblz4_version=lib.LZ4_versionNumber(), vectors=vectors), open(OUT, 'w'))
    print(OUT, os.path.getsize(OUT))


if __name__ == '__main__':
    main()
