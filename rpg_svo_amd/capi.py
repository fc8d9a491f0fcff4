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
        ("tile", C.c_int32),  # SVO_HIP_PYR_*: 1 = 16 x 8 tiles (one 128-byte line each), 0 = row-major (A/B builds)
        ("offset", C.c_int64 * MAX_LEVELS),
        ("slot_bytes", C.c_int64),
    ]


class SiaParams(C.Structure):
    _fields_ = [
        ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("max_level", C.c_int32), ("min_level", C.c_int32), ("n_iter", C.c_int32),
        ("cam_model", C.c_int32), ("eps", C.c_double), ("d", C.c_double * 5),
    ]


class Camera(C.Structure):
    """svo_hip_camera: a vk::AbstractCamera (pinhole, pinhole + radial-tangential, ATAN)."""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32), ("model", C.c_int32), ("reserved", C.c_int32),
                ("d", C.c_double * 5)]


class Frames(C.Structure):
    """svo_hip_frames: frame table (device pointers)."""
    _fields_ = [("n_frames", C.c_int32), ("reserved", C.c_int32), ("d_slot", C.c_void_p), ("d_T_f_w", C.c_void_p)]


class Features(C.Structure):
    """svo_hip_features: SoA svo::Feature records (device pointers)."""
    _fields_ = [("d_frame", C.c_void_p), ("d_level", C.c_void_p), ("d_type", C.c_void_p), ("d_px", C.c_void_p),
                ("d_f", C.c_void_p), ("d_grad", C.c_void_p)]


class Seeds(C.Structure):
    """svo_hip_seeds: SoA svo::Seed state (device pointers)."""
    _fields_ = [("d_a", C.c_void_p), ("d_b", C.c_void_p), ("d_mu", C.c_void_p), ("d_z_range", C.c_void_p),
                ("d_sigma2", C.c_void_p), ("d_batch_id", C.c_void_p)]


class SeedPatch(C.Structure):
    """svo_hip_seed_patch: n new seed records (SoA) and the slots of the resident store they go to."""
    _fields_ = [("n", C.c_int32), ("reserved", C.c_int32), ("d_slot", C.c_void_p), ("src_ftr", Features), ("src_seeds", Seeds)]


class DepthFilterOptions(C.Structure):
    _fields_ = [("max_n_kfs", C.c_int32), ("batch_counter", C.c_int32),
                ("seed_convergence_sigma2_thresh", C.c_double), ("align_1d", C.c_int32),
                ("align_max_iter", C.c_int32), ("max_epi_search_steps", C.c_int32),
                ("subpix_refinement", C.c_int32), ("epi_search_edgelet_filtering", C.c_int32),
                ("n_pyr_levels", C.c_int32), ("epi_search_edgelet_max_angle", C.c_double)]


class Map(C.Structure):
    """svo_hip_map: the device-resident mirror of the map's points and observations (row N2)."""
    _fields_ = [("n_points", C.c_int32), ("n_obs", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("d_pos", "d_type", "d_order", "d_obs_begin", "d_obs_count", "d_obs_frame",
                                          "d_obs_order", "d_obs_level", "d_obs_type", "d_obs_px", "d_obs_f", "d_obs_grad")]


class MapPatch(C.Structure):
    """svo_hip_map_patch: entries rewritten before the map is read."""
    _fields_ = [("n_points", C.c_int32), ("n_obs", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("d_index", "d_pos", "d_type", "d_order", "d_obs_begin", "d_obs_count",
                                          "d_obs_index", "d_obs_order")] + [("obs", Features)]


class Grid(C.Structure):
    """svo_hip_grid: Reprojector::Grid with the visiting rank of every cell."""
    _fields_ = [("cell_size", C.c_int32), ("n_cols", C.c_int32), ("n_rows", C.c_int32), ("n_cells", C.c_int32),
                ("d_cell_rank", C.c_void_p)]


class Reprojection(C.Structure):
    """svo_hip_reprojection: outputs of svo_hip_reproject_map."""
    _fields_ = [(n, C.c_void_p) for n in ("d_header", "d_point_cell", "d_point_px", "d_kf_count", "d_visit_point",
                                          "d_visit_cell", "d_visit_trial", "d_trial_cur", "d_trial_pos",
                                          "d_trial_obs_begin", "d_trial_obs_end", "d_trial_cell", "d_trial_px")]


REPROJ_MAX_IN_FRAME, REPROJ_MAX_CELLS, REPROJ_HEADER = 4096, 2048, 8
FTR_CORNER, FTR_EDGELET = 0, 1
SEED_ERASED_OLD, SEED_BEHIND, SEED_NOT_IN_FRAME, SEED_NO_MATCH, SEED_UPDATED, SEED_CONVERGED, SEED_NAN = range(1, 8)

_vp, _i, _i64 = C.c_void_p, C.c_int, C.c_int64
_LP = C.POINTER(PyrLayout)

# name -> (restype, argtypes); the list every test checks against include/svo_hip.h
PROTOTYPES = {
    "svo_hip_strerror": (C.c_char_p, [_i]),
    "svo_hip_last_hip_error": (_i, []),
    "svo_hip_version": (C.c_char_p, []),
    "svo_hip_device_count": (_i, []),
    "svo_hip_pin_calling_thread": (_i, []),
    "svo_hip_set_device": (_i, [_i]),
    "svo_hip_malloc": (_i, [C.POINTER(_vp), C.c_size_t]),
    "svo_hip_free": (_i, [_vp]),
    "svo_hip_host_alloc": (_i, [C.POINTER(_vp), C.c_size_t]),
    "svo_hip_host_free": (_i, [_vp]),
    "svo_hip_memcpy_h2d": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "svo_hip_memcpy_d2h": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "svo_hip_memcpy_d2d": (_i, [_vp, _vp, C.c_size_t, _vp]),
    "svo_hip_memset": (_i, [_vp, _i, C.c_size_t, _vp]),
    "svo_hip_stream_create": (_i, [C.POINTER(_vp)]),
    "svo_hip_stream_destroy": (_i, [_vp]),
    "svo_hip_stream_sync": (_i, [_vp]),
    "svo_hip_event_create": (_i, [C.POINTER(_vp)]),
    "svo_hip_event_destroy": (_i, [_vp]),
    "svo_hip_event_record": (_i, [_vp, _vp]),
    "svo_hip_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
    "svo_hip_event_sync": (_i, [_vp]),
    "svo_hip_event_query": (_i, [_vp]),
    "svo_hip_stream_wait_event": (_i, [_vp, _vp]),
    "svo_hip_graph_begin_capture": (_i, [_vp]),
    "svo_hip_graph_end_capture": (_i, [_vp, C.POINTER(_vp)]),
    "svo_hip_graph_launch": (_i, [_vp, _vp]),
    "svo_hip_graph_destroy": (_i, [_vp]),
    "svo_hip_camera_pinhole": (_i, [_i, _i] + [C.c_double] * 9 + [C.POINTER(Camera)]),
    "svo_hip_camera_atan": (_i, [_i, _i] + [C.c_double] * 5 + [C.POINTER(Camera)]),
    "svo_hip_pyr_layout_init": (_i, [_i, _i, _i, C.POINTER(PyrLayout)]),
    "svo_hip_pyr_store_bytes": (_i64, [C.POINTER(PyrLayout), _i]),
    "svo_hip_pyramid_load_level0": (_i, [C.POINTER(PyrLayout), _vp, _i, _i, _vp, _i64, _i, _vp]),
    "svo_hip_pyramid_upload_level0": (_i, [C.POINTER(PyrLayout), _vp, _i, _vp, _i, _vp, _vp]),
    "svo_hip_pyramid_upload_level": (_i, [C.POINTER(PyrLayout), _vp, _i, _i, _vp, _i, _vp, _vp]),
    "svo_hip_pyramid_upload_build": (_i, [C.POINTER(PyrLayout), _vp, _i, _vp, _i, _i, _vp, _vp]),
    "svo_hip_pyramid_build": (_i, [C.POINTER(PyrLayout), _vp, _i, _i, _i, _vp]),
    "svo_hip_pyramid_build_from_images": (_i, [C.POINTER(PyrLayout), _vp, _i, _i, _vp, _i64, _i, _i, _vp]),
    "svo_hip_pyramid_build_tiled": (_i, [C.POINTER(PyrLayout), _vp, _i, _i, _vp, _i64, _i, _i, _i, _vp]),
    "svo_hip_pyramid_build_per_level": (_i, [C.POINTER(PyrLayout), _vp, _i, _i, _i, _vp]),
    "svo_hip_pyramid_download_level": (_i, [C.POINTER(PyrLayout), _vp, _i, _i, _vp, _vp]),
    "svo_hip_sparse_align": (_i, [C.POINTER(PyrLayout), _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                  C.POINTER(SiaParams), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "svo_hip_sparse_align_workgroup": (_i, [C.POINTER(PyrLayout), _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp,
                                            C.POINTER(SiaParams), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "svo_hip_solve6_hipsolver_workspace_bytes": (C.c_size_t, [_i]),
    "svo_hip_solve6_hipsolver": (_i, [_i, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "svo_hip_align_batch": (_i, [_LP, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "svo_hip_align_batch_counted": (_i, [_LP, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "svo_hip_align_workspace_bytes": (C.c_size_t, [_i]),
    "svo_hip_align_batch_phased": (_i, [_LP, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "svo_hip_update_seeds_scan_steps": (_vp, [_vp]),
    "svo_hip_update_seeds_count_evaluations": (C.c_int, [C.c_int]),
    "svo_hip_update_seeds_align_evaluations": (_vp, [_vp, C.c_int]),
    "svo_hip_match_workspace_bytes": (C.c_size_t, [_i]),
    "svo_hip_find_match_direct": (_i, [_LP, _vp, C.POINTER(Camera), C.POINTER(Frames), _i, _vp, _vp, _vp,
                                       C.POINTER(Features), _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "svo_hip_reproject_map": (_i, [C.POINTER(Camera), C.POINTER(Frames), _i, _vp, C.POINTER(Map), C.POINTER(MapPatch),
                                   C.POINTER(Grid), _i, _i, _i, _i, C.POINTER(Reprojection), _vp]),
    "svo_hip_find_match_direct_indirect": (_i, [_LP, _vp, C.POINTER(Camera), C.POINTER(Frames), _i, _vp, _vp, _vp, _vp, _vp,
                                                C.POINTER(Features), _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "svo_hip_select_matches_indirect": (_i, [C.POINTER(Camera), _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                             _vp, _vp, _i, _vp]),
    "svo_hip_reproject_points": (_i, [C.POINTER(Camera), C.POINTER(Frames), _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "svo_hip_compose_poses": (_i, [_i, _vp, _vp, _vp, _vp, _vp]),
    "svo_hip_frame_pose_compose": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _vp]),
    "svo_hip_cam2world": (_i, [C.POINTER(Camera), _i, _vp, _vp, _vp]),
    "svo_hip_select_matches": (_i, [C.POINTER(Camera), _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "svo_hip_pose_optimize": (_i, [C.POINTER(Camera), _i, _vp, _i, _vp, _vp, _vp, _vp, C.c_double, _i, _vp, _vp, _vp,
                                   _vp, _vp]),
    "svo_hip_pose_optimize_ordered": (_i, [C.POINTER(Camera), _i, _vp, _i, _vp, _vp, _vp, _vp, C.c_double, _i, _vp, _vp,
                                           _vp, _vp, _vp]),
    "svo_hip_pose_optimize_deferred": (_i, [C.POINTER(Camera), _i, _vp, _i, _vp, _vp, _vp, _vp, C.c_double, _i, _vp, _vp,
                                           _vp, _vp, _vp]),
    "svo_hip_point_optimize": (_i, [C.POINTER(Frames), _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "svo_hip_update_seeds": (_i, [_LP, _vp, C.POINTER(Camera), C.POINTER(Frames), _i, _vp, C.POINTER(Features),
                                  C.POINTER(Seeds), C.POINTER(DepthFilterOptions), _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "svo_hip_seed_store_patch": (_i, [C.POINTER(SeedPatch), C.POINTER(Features), C.POINTER(Seeds), _vp]),
    "svo_hip_update_seeds_resident": (_i, [_LP, _vp, C.POINTER(Camera), C.POINTER(Frames), _i, _i, _vp, C.POINTER(Features),
                                           C.POINTER(Seeds), C.POINTER(DepthFilterOptions), _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "svo_hip_update_seeds_resident_pose": (_i, [_LP, _vp, C.POINTER(Camera), C.POINTER(Frames), _i, _vp, _i, _vp, C.POINTER(Features),
                                                C.POINTER(Seeds), C.POINTER(DepthFilterOptions), _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "svo_hip_find_epipolar_match_direct": (_i, [_LP, _vp, C.POINTER(Camera), C.POINTER(Frames), _i, _vp, C.POINTER(Features),
                                                _vp, _vp, _vp, C.POINTER(DepthFilterOptions), _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "svo_hip_update_seed_batch": (_i, [_i, _vp, _vp, C.POINTER(Seeds), _vp]),
    "svo_hip_fast_workspace_bytes": (C.c_size_t, [_LP, _i, _i]),
    "svo_hip_fast_detect": (_i, [_LP, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, C.c_double, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "svo_hip_compute_tau_batch": (_i, [_i, _vp, _vp, _vp, C.c_double, _vp, _vp]),
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
    # The Python mirror hands torch's streams and device pointers to the library, so both must live on ONE HIP runtime.
    # torch ships its own libamdhip64: imported first, the loader resolves libsvo_hip.so's dependency to that copy; in
    # the other order /opt/rocm's runtime is loaded first and a later `import torch` ends up without a device
    # (hipErrorNoDevice on the first launch; seen with build() and smoke() in one process).
    import torch  # noqa: F401
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


PYR_ROWMAJOR, PYR_TILED = 0, 1


def pyr_px_offset(L: PyrLayout, level: int, x, y):
    """Byte offset inside a slot of pixel (x, y) of `level` (ints or integer numpy arrays): the host mirror of
    csrc/pyr_addr.h for accounting (bench.py's cache-line floor) and tests.  The store itself is only ever
    filled and read through the library."""
    p = int(L.pitch[level])
    if L.tile == PYR_TILED:  # 16 x 8 pixel tiles of 128 bytes, the tiles of an 8-row band consecutive
        return L.offset[level] + (y >> 3) * (8 * p) + ((y & 7) << 4) + ((x >> 4) << 7) + (x & 15)
    return L.offset[level] + y * p + x


def pyr_store_bytes(L: PyrLayout, n_slots: int) -> int:
    return check(load().svo_hip_pyr_store_bytes(C.byref(L), n_slots), "svo_hip_pyr_store_bytes")


def camera(cam) -> Camera:
    """svo_hip_camera of a camera description with fx, fy, cx, cy, width, height and (optionally)
    model / d as svo_hip_camera defines them (rpg_svo_amd.synth.Camera)."""
    d = tuple(getattr(cam, "d", (0.0,) * 5))
    return Camera(cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height, int(getattr(cam, "model", 0)), 0,
                  (C.c_double * 5)(*d))
