"""ctypes binding of the C ABI in include/svo_hip.h (libsvo_hip.so).

The product path has NO fallback: if the gfx950 library is missing or fails to
load, importing the symbols raises.  Nothing here touches oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

MAX_LEVELS = 8
STORE_TAIL_PAD = 256
MAX_PATCHES = 1024

HALFSAMPLE_SCALAR, HALFSAMPLE_SSE2, HALFSAMPLE_AUTO = 0, 1, 2
SIA_STOP = 1

_LIB_PATH = os.environ.get("SVO_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libsvo_hip.so")


class SvoHipError(RuntimeError):
    pass


class PyrLayout(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32),
        ("w", C.c_int32 * MAX_LEVELS),
        ("h", C.c_int32 * MAX_LEVELS),
        ("pitch", C.c_int32 * MAX_LEVELS),
        ("offset", C.c_int64 * MAX_LEVELS),
        ("slot_bytes", C.c_int64),
    ]


class SiaParams(C.Structure):
    _fields_ = [
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("max_level", C.c_int32), ("min_level", C.c_int32), ("n_iter", C.c_int32),
        ("reserved", C.c_int32), ("eps", C.c_double),
    ]


_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64

# name -> (restype, argtypes); the list every test checks against include/svo_hip.h
PROTOTYPES = {
    "svo_hip_strerror": (C.c_char_p, [_i]),
    "svo_hip_last_hip_error": (_i, []),
    "svo_hip_version": (C.c_char_p, []),
    "svo_hip_device_count": (_i, []),
    "svo_hip_set_device": (_i, [_i]),
    "svo_hip_malloc": (_i, [C.POINTER(_vp), C.c_size_t]),
    "svo_hip_free": (_i, [_vp]),
    "svo_hip_memcpy_h2d": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "svo_hip_memcpy_d2h": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "svo_hip_memset": (_i, [_vp, _i, C.c_size_t, _vp]),
    "svo_hip_stream_create": (_i, [C.POINTER(_vp)]),
    "svo_hip_stream_destroy": (_i, [_vp]),
    "svo_hip_stream_sync": (_i, [_vp]),
    "svo_hip_event_create": (_i, [C.POINTER(_vp)]),
    "svo_hip_event_destroy": (_i, [_vp]),
    "svo_hip_event_record": (_i, [_vp, _vp]),
    "svo_hip_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "svo_hip_pyr_layout_init": (_i, [_i, _i, _i, C.POINTER(PyrLayout)]),
    "svo_hip_pyr_store_bytes": (_i64, [C.POINTER(PyrLayout), _i]),
    "svo_hip_pyramid_load_level0": (_i, [C.POINTER(PyrLayout), _vp, _i, _i, _vp, _i64, _i, _vp]),
    "svo_hip_pyramid_upload_level0": (_i, [C.POINTER(PyrLayout), _vp, _i, _vp, _i, _vp]),
    "svo_hip_pyramid_build": (_i, [C.POINTER(PyrLayout), _vp, _i, _i, _i, _vp]),
    "svo_hip_pyramid_download_level": (_i, [C.POINTER(PyrLayout), _vp, _i, _i, _vp, _vp]),
    "svo_hip_sparse_align": (_i, [C.POINTER(PyrLayout), _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                  C.POINTER(SiaParams), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load libsvo_hip.so and bind every prototype.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise SvoHipError(
            f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code: int, what: str = "") -> int:
    if code < 0:
        lib = load()
        msg = lib.svo_hip_strerror(code).decode()
        raise SvoHipError(f"{what or 'svo_hip call'} failed: {msg} (code {code}, hip error {lib.svo_hip_last_hip_error()})")
    return code


def pyr_layout(width: int, height: int, n_levels: int) -> PyrLayout:
    L = PyrLayout()
    check(load().svo_hip_pyr_layout_init(width, height, n_levels, C.byref(L)), "svo_hip_pyr_layout_init")
    return L


def pyr_store_bytes(L: PyrLayout, n_slots: int) -> int:
    return check(load().svo_hip_pyr_store_bytes(C.byref(L), n_slots), "svo_hip_pyr_store_bytes")
