"""ctypes wrappers for the a8-a13 checker entry points -- TEST INFRASTRUCTURE ONLY.

The same Python API serves two libraries with identical C signatures:
  Track("orc")  -> oracle/libsvo_oracle.so     (the C restatement, travels everywhere)
  Track("ref")  -> oracle/_ref/libsvo_ref.so   (the reference's own translation units
                                                compiled against oracle/shim/)
Never imported by the product package rpg_svo_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from . import pyoracle
from .pyoracle import MAX_LEVELS, Pinhole, Pyramid, SiaOptions, SiaResult, _f64, _p, make_cam, make_pyramid_struct

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB_PATH = os.path.join(HERE, "_ref", "libsvo_ref.so")
REF_RELEASE_LIB_PATH = os.path.join(HERE, "_ref", "libsvo_ref_release.so")  # the same units, the reference's release flags
REFERENCE_ROOT = os.environ.get("SVO_REFERENCE", "/root/reference")

SEED_ERASED_OLD, SEED_BEHIND, SEED_NOT_IN_FRAME, SEED_NO_MATCH, SEED_UPDATED, SEED_CONVERGED, SEED_NAN = range(1, 8)


class Frame(C.Structure):
    _fields_ = [("pyr", Pyramid), ("T_f_w", C.c_double * 12)]


class Feature(C.Structure):
    _fields_ = [("frame", C.c_int), ("level", C.c_int), ("type", C.c_int), ("pad_", C.c_int),
                ("px", C.c_double * 2), ("f", C.c_double * 3), ("grad", C.c_double * 2)]


class MatcherOptions(C.Structure):
    _fields_ = [("align_1d", C.c_int), ("align_max_iter", C.c_int), ("max_epi_length_optim", C.c_double),
                ("max_epi_search_steps", C.c_int), ("subpix_refinement", C.c_int),
                ("epi_search_edgelet_filtering", C.c_int), ("epi_search_edgelet_max_angle", C.c_double),
                ("n_pyr_levels", C.c_int), ("pad_", C.c_int)]


class MatchResult(C.Structure):
    _fields_ = [("success", C.c_int), ("ref_obs", C.c_int), ("search_level", C.c_int), ("reject", C.c_int),
                ("A_cur_ref", C.c_double * 4), ("px_cur", C.c_double * 2), ("h_inv", C.c_double),
                ("epi_length", C.c_double), ("depth", C.c_double), ("patch", C.c_uint8 * 64),
                ("patch_with_border", C.c_uint8 * 100)]


class PoseOptResult(C.Structure):
    _fields_ = [("T_f_w", C.c_double * 12), ("Cov", C.c_double * 36), ("estimated_scale", C.c_double),
                ("error_init", C.c_double), ("error_final", C.c_double), ("num_obs", C.c_int),
                ("n_iter_done", C.c_int), ("ran", C.c_int)]


class Seed(C.Structure):
    _fields_ = [("ftr", Feature), ("batch_id", C.c_int), ("a", C.c_float), ("b", C.c_float), ("mu", C.c_float),
                ("z_range", C.c_float), ("sigma2", C.c_float)]


class SeedUpdateInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("search_level", C.c_int), ("z", C.c_double), ("tau", C.c_double),
                ("px_cur", C.c_double * 2), ("xyz_world", C.c_double * 3)]


class DepthFilterOptions(C.Structure):
    _fields_ = [("max_n_kfs", C.c_int), ("batch_counter", C.c_int), ("seed_convergence_sigma2_thresh", C.c_double)]


def matcher_options(n_pyr_levels=3, **kw) -> MatcherOptions:
    o = MatcherOptions(0, 10, 2.0, 1000, 1, 1, 0.7, n_pyr_levels, 0)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def make_feature(frame, px, f, level=0, type_=0, grad=(1.0, 0.0)) -> Feature:
    ft = Feature()
    ft.frame, ft.level, ft.type = int(frame), int(level), int(type_)
    ft.px[:] = [float(px[0]), float(px[1])]
    ft.f[:] = [float(f[0]), float(f[1]), float(f[2])]
    ft.grad[:] = [float(grad[0]), float(grad[1])]
    return ft


def make_frames(pyrs, T_f_w):
    """pyrs: list of pyramids (lists of uint8 level arrays); T_f_w [n,12]."""
    arr = (Frame * len(pyrs))()
    keep = []
    for i, lv in enumerate(pyrs):
        s = make_pyramid_struct(lv)
        keep.append(s)
        arr[i].pyr = s
        arr[i].T_f_w[:] = [float(x) for x in np.asarray(T_f_w[i]).ravel()]
    arr._keep = (keep, pyrs)
    return arr


def ref_available() -> bool:
    return os.path.exists(REF_LIB_PATH)


def build_ref(force: bool = False) -> bool:
    """Compile the reference's translation units in place (needs the reference checkout).
    Returns True when oracle/_ref/libsvo_ref.so exists afterwards."""
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "svo", "src")):
        return ref_available()
    args = ["make", "-C", HERE, "ref", "ref_release", f"REF={REFERENCE_ROOT}"] + (["-B"] if force else [])
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)
    return ref_available()


class Track:
    def __init__(self, which: str = "orc"):
        assert which in ("orc", "ref")
        self.which = which
        if which == "orc":
            self.lib = pyoracle.lib()
        else:
            if not ref_available():
                raise FileNotFoundError(REF_LIB_PATH)
            self.lib = C.CDLL(REF_LIB_PATH)
        g = lambda name: getattr(self.lib, f"{which}_{name}")
        self._align2d = g("align2d")
        self._align1d = g("align1d")
        self._warp_matrix = g("get_warp_matrix_affine")
        self._best_level = g("get_best_search_level")
        self._warp_affine = g("warp_affine")
        self._find_match_direct = g("find_match_direct")
        self._find_epipolar = g("find_epipolar_match_direct")
        self._pose_optimize = g("pose_optimize")
        self._point_optimize = g("point_optimize")
        self._seed_init = g("seed_init")
        self._update_seed = g("update_seed")
        self._compute_tau = g("compute_tau")
        self._compute_tau.restype = C.c_double
        self._update_seeds = g("update_seeds")
        self._reproject_point = g("reproject_point")
        self._cam2world = g("cam2world")
        self._cam2world.restype = None
        self._sia_run = g("sparse_img_align_run")
        if which == "ref":
            self._create_pyr = self.lib.ref_create_img_pyramid

    # -- pyramid ------------------------------------------------------------------------
    def create_img_pyramid(self, img, n_levels, mode=pyoracle.HALFSAMPLE_AUTO):
        if self.which == "orc":
            return pyoracle.create_img_pyramid(img, n_levels, mode)
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        levels = [np.zeros((h >> l, w >> l), dtype=np.uint8) for l in range(n_levels)]
        hh, ww = h, w
        levels = []
        for _ in range(n_levels):
            levels.append(np.zeros((hh, ww), dtype=np.uint8))
            hh, ww = hh // 2, ww // 2
        ptrs = (C.c_void_p * n_levels)(*[l.ctypes.data for l in levels])
        self._create_pyr(_p(img), C.c_int(w), C.c_int(h), C.c_int(n_levels), C.c_int(mode), ptrs)
        return levels

    # -- sparse image alignment ---------------------------------------------------------
    def sparse_img_align_run(self, ref_pyr, cur_pyr, cam, T_ref_w, T_cur_w, px, f, has_point, pos, max_level,
                             min_level, n_iter=30, eps=1e-6):
        rp, cp, pc = make_pyramid_struct(ref_pyr), make_pyramid_struct(cur_pyr), make_cam(cam)
        px, f, pos = _f64(px), _f64(f), _f64(pos)
        n = px.shape[0]
        hp = np.ascontiguousarray(has_point, dtype=np.uint8)
        Tr, Tc = _f64(T_ref_w).copy(), _f64(T_cur_w).copy()
        opt = SiaOptions(max_level, min_level, n_iter, eps)
        res = SiaResult()
        vis = np.zeros(max(n, 1), dtype=np.uint8)
        self._sia_run(C.byref(rp), C.byref(cp), C.byref(pc), _p(Tr), _p(Tc), C.c_int(n), _p(px), _p(f), _p(hp),
                      _p(pos), C.byref(opt), C.byref(res), _p(vis))
        return Tc, pyoracle._res_dict(res, vis[:n])

    # -- feature alignment --------------------------------------------------------------
    def align2d(self, img, pwb, patch, n_iter, px):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        p = _f64(px).copy()
        ok = self._align2d(_p(img), C.c_int(img.shape[1]), C.c_int(img.shape[0]), C.c_int(img.shape[1]),
                           _p(np.ascontiguousarray(pwb, np.uint8)), _p(np.ascontiguousarray(patch, np.uint8)),
                           C.c_int(n_iter), _p(p))
        return bool(ok), p

    def align1d(self, img, dir_, pwb, patch, n_iter, px):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        p = _f64(px).copy()
        d = np.ascontiguousarray(dir_, dtype=np.float32)
        h_inv = C.c_double(0)
        ok = self._align1d(_p(img), C.c_int(img.shape[1]), C.c_int(img.shape[0]), C.c_int(img.shape[1]), _p(d),
                           _p(np.ascontiguousarray(pwb, np.uint8)), _p(np.ascontiguousarray(patch, np.uint8)),
                           C.c_int(n_iter), _p(p), C.byref(h_inv))
        return bool(ok), p, h_inv.value

    # -- warp ---------------------------------------------------------------------------
    def get_warp_matrix_affine(self, cam, px_ref, f_ref, depth, T_cur_ref, level_ref):
        pc = make_cam(cam)
        A = np.zeros(4)
        self._warp_matrix(C.byref(pc), C.byref(pc), _p(_f64(px_ref)), _p(_f64(f_ref)), C.c_double(depth),
                          _p(_f64(T_cur_ref)), C.c_int(level_ref), _p(A))
        return A.reshape(2, 2)

    def get_best_search_level(self, A, max_level):
        return int(self._best_level(_p(_f64(A).ravel().copy()), C.c_int(max_level)))

    def warp_affine(self, A, img, px_ref, level_ref, search_level, halfpatch_size):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        patch = np.zeros((2 * halfpatch_size) ** 2, dtype=np.uint8)
        ok = self._warp_affine(_p(_f64(A).ravel().copy()), _p(img), C.c_int(img.shape[1]), C.c_int(img.shape[0]),
                               C.c_int(img.shape[1]), _p(_f64(px_ref)), C.c_int(level_ref), C.c_int(search_level),
                               C.c_int(halfpatch_size), _p(patch))
        return bool(ok), patch

    # -- matcher ------------------------------------------------------------------------
    def find_match_direct(self, frames, cam, cur_frame, pt_pos, obs, px_cur, opt=None):
        opt = opt or matcher_options()
        pc = make_cam(cam)
        arr = (Feature * len(obs))(*obs)
        res = MatchResult()
        px = _f64(px_cur).copy()
        ok = self._find_match_direct(frames, C.byref(pc), C.c_int(cur_frame), _p(_f64(pt_pos)), C.c_int(len(obs)), arr,
                                     C.byref(opt), _p(px), C.byref(res))
        return bool(ok), px, _match_dict(res)

    def find_epipolar_match_direct(self, frames, cam, ref_frame, cur_frame, ftr, d_estimate, d_min, d_max, opt=None):
        opt = opt or matcher_options()
        pc = make_cam(cam)
        res = MatchResult()
        ok = self._find_epipolar(frames, C.byref(pc), C.c_int(ref_frame), C.c_int(cur_frame), C.byref(ftr),
                                 C.c_double(d_estimate), C.c_double(d_min), C.c_double(d_max), C.byref(opt),
                                 C.byref(res))
        return bool(ok), _match_dict(res)

    # -- pose optimizer -----------------------------------------------------------------
    def pose_optimize(self, cam, T_f_w, f, level, has_point, pos, reproj_thresh=2.0, n_iter=10):
        pc = make_cam(cam)
        f, pos = _f64(f), _f64(pos)
        n = f.shape[0]
        lv = np.ascontiguousarray(level, dtype=np.int32)
        hp = np.ascontiguousarray(has_point, dtype=np.uint8).copy()
        res = PoseOptResult()
        self._pose_optimize(C.c_double(reproj_thresh), C.c_int(n_iter), C.byref(pc), _p(_f64(T_f_w)), C.c_int(n),
                            _p(f), _p(lv), _p(hp), _p(pos), C.byref(res))
        return dict(T_f_w=np.array(res.T_f_w[:]), Cov=np.array(res.Cov[:]).reshape(6, 6),
                    estimated_scale=res.estimated_scale, error_init=res.error_init, error_final=res.error_final,
                    num_obs=res.num_obs, n_iter_done=res.n_iter_done, ran=res.ran, has_point=hp)

    def point_optimize(self, T_f_w, f, pos, n_iter=5):
        T, f = _f64(T_f_w), _f64(f)
        p = _f64(pos).copy()
        self._point_optimize(C.c_int(n_iter), C.c_int(T.shape[0]), _p(T), _p(f), _p(p))
        return p

    # -- depth filter -------------------------------------------------------------------
    def seed_init(self, depth_mean, depth_min) -> Seed:
        s = Seed()
        self._seed_init(C.byref(s), C.c_float(depth_mean), C.c_float(depth_min))
        return s

    def update_seed(self, x, tau2, seed: Seed) -> Seed:
        s = Seed.from_buffer_copy(seed)
        self._update_seed(C.c_float(x), C.c_float(tau2), C.byref(s))
        return s

    def compute_tau(self, T_ref_cur, f, z, px_error_angle):
        return float(self._compute_tau(_p(_f64(T_ref_cur)), _p(_f64(f)), C.c_double(z), C.c_double(px_error_angle)))

    def update_seeds(self, frames, cam, cur_frame, seeds, batch_counter, max_n_kfs=3, conv_thresh=200.0, opt=None):
        opt = opt or matcher_options()
        pc = make_cam(cam)
        n = len(seeds)
        arr = (Seed * n)(*[Seed.from_buffer_copy(s) for s in seeds])
        info = (SeedUpdateInfo * n)()
        dopt = DepthFilterOptions(max_n_kfs, batch_counter, conv_thresh)
        nu = self._update_seeds(frames, C.byref(pc), C.c_int(cur_frame), C.c_int(n), arr, info, C.byref(dopt),
                                C.byref(opt))
        return int(nu), list(arr), list(info)

    def cam2world(self, cam, px):
        """Unit bearings [n,3] of pixels px [n,2] (Frame::c2f)."""
        pc = make_cam(cam)
        px = _f64(px).reshape(-1, 2)
        f = np.zeros((px.shape[0], 3))
        self._cam2world(C.byref(pc), C.c_int(px.shape[0]), _p(px), _p(f))
        return f

    def reproject_point(self, cam, T_f_w, pos, cell_size, grid_n_cols):
        pc = make_cam(cam)
        px = np.zeros(2)
        k = self._reproject_point(C.byref(pc), _p(_f64(T_f_w)), _p(_f64(pos)), C.c_int(cell_size),
                                  C.c_int(grid_n_cols), _p(px))
        return int(k), px


def select_matches(cam, cell, ok, px, level, pos, max_fts):
    """Reprojector::reprojectMap's cell loop + reprojectCell (reprojector.cpp:131-139, 150-200) over trials with a
    known outcome (restatement only: the reference's own version is private and entangled with the map).
    Returns (sel, f, level_out, pos_out) of the new features in Frame::fts_ order."""
    lib = pyoracle.lib()
    pc = make_cam(cam)
    cell = np.ascontiguousarray(cell, dtype=np.int32)
    ok = np.ascontiguousarray(ok, dtype=np.int32)
    level = np.ascontiguousarray(level, dtype=np.int32)
    px, pos = _f64(px).reshape(-1, 2), _f64(pos).reshape(-1, 3)
    M = cell.shape[0]
    cap = max(1, min(M, max_fts + 1))
    sel = np.zeros(cap, dtype=np.int32)
    f, pos_out = np.zeros((cap, 3)), np.zeros((cap, 3))
    level_out = np.zeros(cap, dtype=np.int32)
    i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    n = lib.orc_select_matches(C.byref(pc), C.c_int(M), i32(cell), i32(ok), _p(px), i32(level), _p(pos), C.c_int(max_fts),
                               i32(sel), _p(f), i32(level_out), _p(pos_out))
    return sel[:n], f[:n], level_out[:n], pos_out[:n]


def reproject_map(cam, frames_T, cur_frame, kf_rank, pos, type, order, obs_begin, obs_count, obs_frame, obs_order, cell_size,
                  n_cols, n_cells, cell_rank, first_cell=0, max_cells_with_trials=1 << 30):
    """Reprojector::reprojectMap up to the first findMatchDirect on the plain-array form of the map (orc_reproject_map,
    svo_oracle_track.c): returns a dict of numpy arrays named like svo_hip_reprojection's fields."""
    lib = pyoracle.lib()
    pc = make_cam(cam)
    i32a = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    i32 = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    frames_T = _f64(frames_T).reshape(-1, 12)
    pos = _f64(pos).reshape(-1, 3)
    P = pos.shape[0]
    kf_rank, type, order, obs_begin, obs_count = map(i32a, (kf_rank, type, order, obs_begin, obs_count))
    obs_frame, obs_order, cell_rank = map(i32a, (obs_frame, obs_order, cell_rank))
    header = np.zeros(8, dtype=np.int32)
    cap = max(P, 1)
    out = {"point_cell": np.zeros(cap, dtype=np.int32), "point_px": np.zeros((cap, 2)), "kf_count": np.zeros(frames_T.shape[0], dtype=np.int32),
           "visit_point": np.zeros(cap, dtype=np.int32), "visit_cell": np.zeros(cap, dtype=np.int32),
           "visit_trial": np.zeros(cap, dtype=np.int32), "trial_obs": np.zeros(cap, dtype=np.int32),
           "trial_cell": np.zeros(cap, dtype=np.int32), "trial_px": np.zeros((cap, 2)), "trial_pos": np.zeros((cap, 3))}
    lib.orc_reproject_map.restype = C.c_int
    lib.orc_reproject_map(C.byref(pc), C.c_int(frames_T.shape[0]), _p(frames_T), C.c_int(cur_frame), i32(kf_rank), C.c_int(P), _p(pos),
                          i32(type), i32(order), i32(obs_begin), i32(obs_count), i32(obs_frame), i32(obs_order), C.c_int(cell_size),
                          C.c_int(n_cols), C.c_int(n_cells), i32(cell_rank), C.c_int(first_cell),
                          C.c_int(min(max_cells_with_trials, 1 << 30)), i32(header), i32(out["point_cell"]), _p(out["point_px"]),
                          i32(out["kf_count"]), i32(out["visit_point"]), i32(out["visit_cell"]), i32(out["visit_trial"]),
                          i32(out["trial_obs"]), i32(out["trial_cell"]), _p(out["trial_px"]), _p(out["trial_pos"]))
    V, M = int(header[2]), int(header[3])
    out["header"] = header
    for k in ("visit_point", "visit_cell", "visit_trial"):
        out[k] = out[k][:V]
    for k in ("trial_obs", "trial_cell", "trial_px", "trial_pos"):
        out[k] = out[k][:M]
    out["point_cell"], out["point_px"] = out["point_cell"][:P], out["point_px"][:P]
    return out


def fast_detect_grid(pyr_levels, n_levels, cell_size, cols, rows, occupancy=None, fast_threshold=20,
                     detection_threshold=20.0):
    """C restatement of FastDetector::detect: (xy [cells,2], level [cells], score [cells], n_features)."""
    lib = pyoracle.lib()
    ps = make_pyramid_struct(pyr_levels)
    n_cells = cols * rows
    xy = np.zeros((n_cells, 2), dtype=np.int32)
    lvl = np.zeros(n_cells, dtype=np.int32)
    sc = np.zeros(n_cells, dtype=np.float32)
    occ = None if occupancy is None else np.ascontiguousarray(occupancy, dtype=np.uint8)
    n = lib.orc_fast_detect_grid(C.byref(ps), C.c_int(n_levels), C.c_int(fast_threshold), C.c_int(cell_size), C.c_int(cols),
                                 C.c_int(rows), _p(occ) if occ is not None else None, C.c_double(detection_threshold),
                                 _p(xy), _p(lvl), _p(sc))
    return xy, lvl, sc, n


def ref_fast_detect(pyr_levels, cam, n_levels, cell_size, occupancy=None, detection_threshold=20.0, max_out=4096):
    """The reference's own FastDetector::detect (oracle/_ref): (px [n,2], level [n]) in emission order."""
    lib = C.CDLL(REF_LIB_PATH)
    ps = make_pyramid_struct(pyr_levels)
    pc = make_cam(cam)
    px = np.zeros((max_out, 2))
    lvl = np.zeros(max_out, dtype=np.int32)
    occ = None if occupancy is None else np.ascontiguousarray(occupancy, dtype=np.uint8)
    n = lib.ref_fast_detect(C.byref(ps), C.byref(pc), C.c_int(n_levels), C.c_int(cell_size),
                            _p(occ) if occ is not None else None, C.c_double(detection_threshold), C.c_int(max_out),
                            _p(px), _p(lvl))
    return px[:n], lvl[:n]


def _match_dict(r: MatchResult) -> dict:
    return dict(success=r.success, ref_obs=r.ref_obs, search_level=r.search_level, reject=r.reject,
                A_cur_ref=np.array(r.A_cur_ref[:]).reshape(2, 2), px_cur=np.array(r.px_cur[:]), h_inv=r.h_inv,
                epi_length=r.epi_length, depth=r.depth, patch=np.array(r.patch[:], dtype=np.uint8),
                patch_with_border=np.array(r.patch_with_border[:], dtype=np.uint8))
