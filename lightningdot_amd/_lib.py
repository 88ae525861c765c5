"""ctypes binding of libldot.so (the C ABI declared in include/ldot.h).  Fails loudly: no fallbacks."""
import ctypes
import os
import threading

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = 'libldot.so'
_lock = threading.Lock()
_lib = None

# constants mirrored from include/ldot.h
LDOT_OK = 0
F32, BF16, F16 = 0, 1, 2
HOST, DEVICE = 0, 1
MODE_AUTO, MODE_DENSE, MODE_FUSED = 0, 1, 2
OPT_MODE, OPT_RESCORE, OPT_CHUNK_ROWS, OPT_MARGIN, OPT_PROFILE, OPT_WARM_ROWS, OPT_GROWTH_PCT, OPT_PRECISION, OPT_RESERVE_ROWS = 1, 2, 3, 4, 5, 6, 7, 8, 9
OPT_OPTIMISTIC = 11
OPT_SCAN_ORDER = 12
OPT_ROW_SHUFFLE = 13
OPT_DEFER_SYNC = 14
OPT_RESULT_SET = 15
OPT_VERIFY = 10
ABI_VERSION = 7
MAX_MARGIN = 1024
PAD_LABEL = -1
PAD_SCORE = -3.4028234663852886e+38
MAX_K = 2048

# every exported symbol of include/ldot.h: name -> (restype, argtypes)
_c = ctypes
_vp, _i, _i64, _f = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float
SYMBOLS = {
    'ldot_last_error': (_c.c_char_p, []),
    'ldot_abi_version': (_i, []),
    'ldot_device_count': (_i, []),
    'ldot_index_create': (_i, [_i, _c.POINTER(_vp)]),
    'ldot_index_destroy': (_i, [_vp]),
    'ldot_index_add': (_i, [_vp, _vp, _i64, _i, _i, _i, _vp]),
    'ldot_index_ntotal': (_i64, [_vp]),
    'ldot_index_dim': (_i, [_vp]),
    'ldot_index_reset': (_i, [_vp]),
    'ldot_index_set_option': (_i, [_vp, _i, _i64]),
    'ldot_index_search': (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'ldot_index_search_begin': (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    'ldot_index_search_finish': (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    'ldot_index_search_warmup': (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp]),
    'ldot_index_search_scan': (_i, [_vp, _vp, _vp, _vp]),
    'ldot_index_search_begin_shard': (_i, [_vp, _vp, _i64, _i, _i, _i, _i, _i, ctypes.c_double, _i64, _vp, _vp]),
    'ldot_index_shard_floor': (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    'ldot_index_search_finish_blocked': (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    'ldot_merge_topk_blocked': (_i, [_vp, _i, _i64, _i64, _i64, _i, _i, _vp, _vp, _vp]),
    'ldot_index_search_lists': (_i, [_vp, _vp, _i64, _i, _i, _vp, _i, _i64, _vp, _i, _i, _vp, _vp, _i, _vp]),
    'ldot_ivf_search': (_i, [_vp, _vp, _vp, _i64, _i, _i, _vp, _i64, _i, _i, _vp, _vp, _i, _vp]),
    'ldot_index_save': (_i, [_vp, _c.c_char_p]),
    'ldot_index_load': (_i, [_c.c_char_p, _c.POINTER(_vp)]),
    'ldot_index_get_rows': (_i, [_vp, _i64, _i64, _vp, _i, _vp]),
    'ldot_index_last_stats': (_i, [_vp, _c.POINTER(_i64)]),
    'ldot_index_last_regime': (_i, [_vp, _c.POINTER(_i64)]),
    'ldot_index_last_unproven': (_i, [_vp, _vp, _c.POINTER(_i64)]),
    'ldot_index_last_set_stats': (_i, [_vp, _c.POINTER(_i64)]),
    'ldot_index_last_profile': (_i, [_vp, _c.POINTER(_c.c_double)]),
    'ldot_merge_topk': (_i, [_vp, _vp, _i, _i64, _i, _i, _vp, _vp, _i, _vp]),
    'ldot_cls_pool': (_i, [_vp, _i, _i64, _i64, _i64, _i, _vp, _vp, _vp]),
    'ldot_inbatch_nll_fwd': (_i, [_vp, _vp, _vp, _f, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ldot_inbatch_nll_bwd': (_i, [_vp, _vp, _vp, _f, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ldot_dot_product_scores': (_i, [_vp, _vp, _i64, _i64, _i64, _vp, _vp]),
    'ldot_inbatch_nll_bidir_fwd': (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'ldot_inbatch_nll_bidir_bwd': (_i, [_vp, _vp, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}


class LdotError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f'libldot error {code}: {msg}')
        self.code = code


def lib_path() -> str:
    return os.environ.get('LDOT_LIBRARY', os.path.join(_PKG, _LIB_NAME))


def load_library():
    """Load libldot.so (torch is imported first so that the HIP runtime already mapped by torch — same soname
    libamdhip64.so.7 — is the one the library binds to).  Raises if the library is missing: the product path has
    no CPU fallback."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not os.path.exists(path):
            raise LdotError(-2, f'{path} not found: build it with `python -m lightningdot_amd.build` '
                                f'(or __graft_entry__.build()); there is no CPU fallback')
        try:
            import torch  # noqa: F401  (maps the ROCm runtime libraries)
        except Exception:  # pragma: no cover - torch is optional for pure C-ABI users
            pass
        lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)     # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if lib.ldot_abi_version() != ABI_VERSION:
            raise LdotError(-5, f'ABI version mismatch: library {lib.ldot_abi_version()}, binding {ABI_VERSION}')
        _lib = lib
        return lib


def check(code: int):
    if code != LDOT_OK:
        msg = load_library().ldot_last_error()
        raise LdotError(code, msg.decode(errors='replace') if msg else '')


def require_gpu():
    lib = load_library()
    n = lib.ldot_device_count()
    if n <= 0:
        raise LdotError(-2, 'no HIP device visible: the MI355X path has no CPU fallback')
    return n
