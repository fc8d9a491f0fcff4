"""ctypes wrapper of oracle/libsvo_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never from the product package rpg_svo_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsvo_oracle.so")
MAX_LEVELS = 8
HALFSAMPLE_SCALAR, HALFSAMPLE_SSE2, HALFSAMPLE_AUTO = 0, 1, 2


class Pyramid(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("w", C.c_int * MAX_LEVELS), ("h", C.c_int * MAX_LEVELS),
                ("data", C.c_void_p * MAX_LEVELS)]


class Pinhole(C.Structure):
    """orc_pinhole (orc_camera.h): intrinsics + model tag + distortion parameters."""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("width", C.c_int), ("height", C.c_int), ("model", C.c_int), ("pad_", C.c_int), ("d", C.c_double * 5)]


class SiaOptions(C.Structure):
    _fields_ = [("max_level", C.c_int), ("min_level", C.c_int), ("n_iter", C.c_int), ("eps", C.c_double)]


class SiaResult(C.Structure):
    _fields_ = [("n_tracked", C.c_int), ("stop", C.c_int), ("iters", C.c_int * MAX_LEVELS),
                ("chi2", C.c_double), ("H", C.c_double * 36), ("T_cur_from_ref", C.c_double * 12)]


_lib = None


def build(force: bool = False) -> None:
    args = ["make", "-C", HERE] + (["-B"] if force else [])
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)


def build_ref(force: bool = False) -> bool:
    """oracle/_ref/libsvo_ref.so: the reference's own translation units compiled in place
    (only where the reference checkout exists; see oracle/Makefile)."""
    from . import pytrack
    return pytrack.build_ref(force)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_sparse_img_align_run.restype = C.c_int
        _lib.orc_sparse_img_align_batch.restype = C.c_int
        _lib.orc_ldlt_solve_n.restype = C.c_int
    return _lib


_ref_lib = None


def ref_available() -> bool:
    from . import pytrack
    return pytrack.ref_available()


def ref_lib() -> C.CDLL:
    """oracle/_ref/libsvo_ref.so (prebuilt; built here only where the reference checkout exists)."""
    global _ref_lib
    if _ref_lib is None:
        from . import pytrack
        if not pytrack.ref_available():
            build_ref()
        _ref_lib = C.CDLL(pytrack.REF_LIB_PATH)
    return _ref_lib


_ref_release_lib = None


def ref_release_available() -> bool:
    from . import pytrack
    return os.path.exists(pytrack.REF_RELEASE_LIB_PATH)


def ref_release_lib() -> C.CDLL:
    """oracle/_ref/libsvo_ref_release.so: the reference's translation units with the reference's release flags
    (oracle/Makefile, target ref_release) -- timed by bench.py's cpu_baseline next to the bit-comparable build, never
    compared bit for bit.  RTLD_DEEPBIND: it exports the same symbols as libsvo_ref.so and must bind to its own."""
    global _ref_release_lib
    if _ref_release_lib is None:
        from . import pytrack
        _ref_release_lib = C.CDLL(pytrack.REF_RELEASE_LIB_PATH, mode=os.RTLD_LOCAL | os.RTLD_NOW | os.RTLD_DEEPBIND)
    return _ref_release_lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- SE(3) -----------------------------------------------------------------
def se3_exp(xi):
    out = np.zeros(12)
    lib().orc_se3_exp(_p(_f64(xi)), _p(out))
    return out


def se3_log(T):
    out = np.zeros(6)
    lib().orc_se3_log(_p(_f64(T)), _p(out))
    return out


def se3_mul(A, B):
    out = np.zeros(12)
    lib().orc_se3_mul(_p(_f64(A)), _p(_f64(B)), _p(out))
    return out


def se3_inv(A):
    out = np.zeros(12)
    lib().orc_se3_inv(_p(_f64(A)), _p(out))
    return out


def ldlt_solve(H, b):
    H = _f64(H)
    b = _f64(b)
    x = np.zeros(b.shape[0])
    lib().orc_ldlt_solve_n(C.c_int(b.shape[0]), _p(H), _p(b), _p(x))
    return x


# ---- pyramid ---------------------------------------------------------------
def half_sample(img: np.ndarray, mode: int = HALFSAMPLE_AUTO) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.zeros((h // 2, w // 2), dtype=np.uint8)
    lib().orc_half_sample(_p(img), C.c_int(w), C.c_int(h), C.c_int(w), _p(out), C.c_int(w // 2), C.c_int(mode))
    return out


def create_img_pyramid(img: np.ndarray, n_levels: int, mode: int = HALFSAMPLE_AUTO) -> list[np.ndarray]:
    """frame_utils::createImgPyramid (svo/src/frame.cpp:156-165)."""
    pyr = [np.ascontiguousarray(img, dtype=np.uint8)]
    for _ in range(1, n_levels):
        pyr.append(half_sample(pyr[-1], mode))
    return pyr


def make_pyramid_struct(levels: list[np.ndarray]) -> Pyramid:
    p = Pyramid()
    p.n_levels = len(levels)
    for i, l in enumerate(levels):
        assert l.dtype == np.uint8 and l.flags["C_CONTIGUOUS"]
        p.w[i] = l.shape[1]
        p.h[i] = l.shape[0]
        p.data[i] = l.ctypes.data
    p._keep = levels
    return p


def make_cam(cam) -> Pinhole:
    d = tuple(getattr(cam, "d", (0.0,) * 5))
    return Pinhole(cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height, int(getattr(cam, "model", 0)), 0, (C.c_double * 5)(*d))


# ---- SparseImgAlign --------------------------------------------------------
def sparse_img_align_run(ref_pyr, cur_pyr, cam, T_ref_w, T_cur_w, px, f, has_point, pos,
                         max_level, min_level, n_iter=30, eps=1e-6):
    """One SparseImgAlign::run.  Returns (T_cur_w_new [12], result dict)."""
    rp = make_pyramid_struct(ref_pyr)
    cp = make_pyramid_struct(cur_pyr)
    pc = make_cam(cam)
    px, f, pos = _f64(px), _f64(f), _f64(pos)
    n = px.shape[0]
    hp = np.ascontiguousarray(has_point, dtype=np.uint8)
    Tr = _f64(T_ref_w).copy()
    Tc = _f64(T_cur_w).copy()
    opt = SiaOptions(max_level, min_level, n_iter, eps)
    res = SiaResult()
    vis = np.zeros(max(n, 1), dtype=np.uint8)
    lib().orc_sparse_img_align_run(C.byref(rp), C.byref(cp), C.byref(pc), _p(Tr), _p(Tc), C.c_int(n),
                                   _p(px), _p(f), _p(hp), _p(pos), C.byref(opt), C.byref(res), _p(vis))
    return Tc, _res_dict(res, vis[:n])


def _res_dict(res: SiaResult, vis=None) -> dict:
    d = dict(n_tracked=res.n_tracked, stop=res.stop, iters=np.array(res.iters[:]), chi2=res.chi2,
             H=np.array(res.H[:]).reshape(6, 6), T_cur_from_ref=np.array(res.T_cur_from_ref[:]))
    if vis is not None:
        d["visible"] = vis.copy()
    return d


def sparse_img_align_batch(pyrs, ref_slot, cur_slot, cam, T_ref_w, T_cur_w, n, px, f, has_point, pos,
                           max_level, min_level, n_iter=30, eps=1e-6, n_threads=1, which="orc", timing=None):
    """pyrs: list of pyramids (list of level arrays).  Arrays are [B, n_stride, .].
    Returns (T_cur_w_new [B,12], list of result dicts)."""
    B = len(ref_slot)
    arr = (Pyramid * len(pyrs))()
    keep = []
    for i, lv in enumerate(pyrs):
        s = make_pyramid_struct(lv)
        keep.append(s)
        arr[i] = s
    pc = make_cam(cam)
    px, f, pos = _f64(px), _f64(f), _f64(pos)
    n_stride = px.shape[1]
    hp = np.ascontiguousarray(has_point, dtype=np.uint8)
    Tr = _f64(T_ref_w).copy()
    Tc = _f64(T_cur_w).copy()
    rs = np.ascontiguousarray(ref_slot, dtype=np.int32)
    cs = np.ascontiguousarray(cur_slot, dtype=np.int32)
    nn = np.ascontiguousarray(n, dtype=np.int32)
    opt = SiaOptions(max_level, min_level, n_iter, eps)
    res = (SiaResult * B)()
    if which in ("ref", "ref_release"):  # the reference's own SparseImgAlign (oracle/_ref/libsvo_ref[_release].so)
        secs = C.c_double(0)
        (ref_lib() if which == "ref" else ref_release_lib()).ref_sparse_img_align_batch(C.c_int(B), arr, _p(rs), _p(cs), C.byref(pc), _p(Tr), _p(Tc), _p(nn),
                                             C.c_int(n_stride), _p(px), _p(f), _p(hp), _p(pos), C.byref(opt), res,
                                             C.c_int(n_threads), C.byref(secs))
        if timing is not None:
            timing["run_seconds"] = secs.value
        return Tc, [_res_dict(r) for r in res]
    lib().orc_sparse_img_align_batch(C.c_int(B), arr, _p(rs), _p(cs), C.byref(pc), _p(Tr), _p(Tc), _p(nn),
                                     C.c_int(n_stride), _p(px), _p(f), _p(hp), _p(pos), C.byref(opt), res,
                                     C.c_int(n_threads))
    return Tc, [_res_dict(r) for r in res]
