"""Batched host mirrors of the reference classes that run after sparse alignment, over the
C ABI (include/svo_hip.h).  Names and argument meaning follow the reference:

  feature_alignment.align2D / align1D   svo/include/svo/feature_alignment.h:29-44
  Matcher.findMatchDirect               svo/include/svo/matcher.h:106-111
  Reprojector.reprojectPoint            svo/src/reprojector.cpp:206-217
  pose_optimizer.optimizeGaussNewton    svo/include/svo/pose_optimizer.h:37-45
  Point.optimize                        svo/include/svo/point.h:85
  DepthFilter.updateSeeds / updateSeed  svo/include/svo/depth_filter.h:120-135,158

Every call takes device-resident torch tensors (torch is the allocator / stream provider
only) and enqueues on the current stream.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch

from . import capi
from .pyramid import PyramidStore, _stream_ptr


def _ptr(t):
    return None if t is None else t.data_ptr()


def _chk(t, dtype):
    assert t.is_cuda and t.dtype == dtype and t.is_contiguous(), "device-resident contiguous tensor required"
    return t


class FrameTable:
    """Device mirror of the svo::Frame objects a batch refers to: pyramid slot + T_f_w_."""

    def __init__(self, slot: torch.Tensor, T_f_w: torch.Tensor):
        self.slot = _chk(slot, torch.int32)
        self.T_f_w = _chk(T_f_w, torch.float64)
        assert T_f_w.shape == (slot.shape[0], 12)

    def struct(self) -> capi.Frames:
        return capi.Frames(self.slot.shape[0], 0, self.slot.data_ptr(), self.T_f_w.data_ptr())


@dataclass
class FeatureSet:
    """SoA of svo::Feature (feature.h:26-71)."""
    frame: torch.Tensor            # [n] i32 index into a FrameTable
    level: torch.Tensor            # [n] i32
    px: torch.Tensor               # [n,2] f64
    f: torch.Tensor                # [n,3] f64
    type: torch.Tensor | None = None   # [n] u8
    grad: torch.Tensor | None = None   # [n,2] f64

    def struct(self) -> capi.Features:
        _chk(self.frame, torch.int32); _chk(self.level, torch.int32); _chk(self.px, torch.float64); _chk(self.f, torch.float64)
        if self.type is not None:
            _chk(self.type, torch.uint8); _chk(self.grad, torch.float64)
        return capi.Features(self.frame.data_ptr(), self.level.data_ptr(), _ptr(self.type), self.px.data_ptr(),
                             self.f.data_ptr(), _ptr(self.grad))


class _Workspace:
    def __init__(self):
        self.buf = None

    def get(self, lib, M: int, device):
        need = lib.svo_hip_match_workspace_bytes(M)
        if self.buf is None or self.buf.numel() < need or self.buf.device != device:
            self.buf = torch.empty(need, dtype=torch.uint8, device=device)
        return self.buf


# ---- feature_alignment ----------------------------------------------------------------
def align_batch(store: PyramidStore, slot, level, patch_with_border, px, n_iter: int, dir=None, use_1d=None, phased: bool = False,
                evaluations=None):
    """feature_alignment::align2D (use_1d None/0) or align1D per trial.  px [M,2] f64 is refined
    in place (level coordinates).  Returns (ok [M] i32, h_inv [M] f64).  phased: run the iterations in three
    launches with the unfinished trials compacted in between (svo_hip_align_batch_phased; same results)."""
    lib = capi.load()
    M = px.shape[0]
    if phased:
        _chk(slot, torch.int32); _chk(level, torch.int32); _chk(patch_with_border, torch.uint8); _chk(px, torch.float64)
        ok = torch.zeros(M, dtype=torch.int32, device=px.device)
        h_inv = torch.zeros(M, dtype=torch.float64, device=px.device)
        ws = torch.empty(max(int(lib.svo_hip_align_workspace_bytes(M)), 256), dtype=torch.uint8, device=px.device)
        capi.check(lib.svo_hip_align_batch_phased(C.byref(store.layout), store.ptr, M, slot.data_ptr(), level.data_ptr(),
                                                  patch_with_border.data_ptr(), _ptr(dir), _ptr(use_1d), n_iter, px.data_ptr(),
                                                  ok.data_ptr(), h_inv.data_ptr(), _ptr(evaluations), ws.data_ptr(), ws.numel(),
                                                  _stream_ptr(store.device)), "svo_hip_align_batch_phased")
        return ok, h_inv
    _chk(slot, torch.int32); _chk(level, torch.int32); _chk(patch_with_border, torch.uint8); _chk(px, torch.float64)
    assert patch_with_border.shape == (M, 100)
    ok = torch.zeros(M, dtype=torch.int32, device=px.device)
    h_inv = torch.zeros(M, dtype=torch.float64, device=px.device)
    capi.check(lib.svo_hip_align_batch(C.byref(store.layout), store.ptr, M, slot.data_ptr(), level.data_ptr(),
                                       patch_with_border.data_ptr(), _ptr(dir), _ptr(use_1d), n_iter, px.data_ptr(),
                                       ok.data_ptr(), h_inv.data_ptr(), _stream_ptr(store.device)), "svo_hip_align_batch")
    return ok, h_inv


# ---- Matcher ----------------------------------------------------------------------------
@dataclass
class MatchResult:
    ok: torch.Tensor            # [M] i32   findMatchDirect's return value
    px_cur: torch.Tensor        # [M,2] f64 refined pixel (level 0)
    ref_obs: torch.Tensor       # [M] i32   observation chosen as ref_ftr_
    search_level: torch.Tensor  # [M] i32
    A_cur_ref: torch.Tensor     # [M,4] f64
    patch_with_border: torch.Tensor  # [M,100] u8


class Matcher:
    """Batched svo::Matcher (matcher.h:64-127): options as in Matcher::Options."""

    def __init__(self, align_max_iter: int = 10, n_pyr_levels: int = 3):
        self.align_max_iter = align_max_iter
        self.n_pyr_levels = n_pyr_levels  # Config::nPyrLevels()
        self.lib = capi.load()
        self._ws = _Workspace()

    def alloc_result(self, M: int, device) -> MatchResult:
        return MatchResult(torch.zeros(M, dtype=torch.int32, device=device), torch.zeros(M, 2, dtype=torch.float64, device=device),
                           torch.zeros(M, dtype=torch.int32, device=device), torch.zeros(M, dtype=torch.int32, device=device),
                           torch.zeros(M, 4, dtype=torch.float64, device=device),
                           torch.zeros(M, 100, dtype=torch.uint8, device=device))

    def find_match_direct(self, store: PyramidStore, cam, frames: FrameTable, cur_frame, pt_pos, obs_ptr,
                          obs: FeatureSet, px_cur, out: MatchResult | None = None) -> MatchResult:
        """out: a result block from alloc_result() to reuse (every field is overwritten by the
        kernels; px_cur is copied into out.px_cur first)."""
        M = pt_pos.shape[0]
        dev = store.device
        _chk(cur_frame, torch.int32); _chk(pt_pos, torch.float64); _chk(obs_ptr, torch.int32); _chk(px_cur, torch.float64)
        res = out if out is not None else self.alloc_result(M, dev)
        res.px_cur.copy_(px_cur)
        ws = self._ws.get(self.lib, M, dev)
        c, fr, ob = capi.camera(cam), frames.struct(), obs.struct()
        capi.check(self.lib.svo_hip_find_match_direct(
            C.byref(store.layout), store.ptr, C.byref(c), C.byref(fr), M, cur_frame.data_ptr(), pt_pos.data_ptr(),
            obs_ptr.data_ptr(), C.byref(ob), self.n_pyr_levels, self.align_max_iter, res.px_cur.data_ptr(),
            res.ok.data_ptr(), res.ref_obs.data_ptr(), res.search_level.data_ptr(), res.A_cur_ref.data_ptr(),
            res.patch_with_border.data_ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(dev)), "svo_hip_find_match_direct")
        return res


def find_epipolar_match_direct(store: PyramidStore, cam, frames: FrameTable, cur_frame, ftr: FeatureSet, d_estimate, d_min,
                               d_max, n_pyr_levels: int = 3, align_1d: bool = False, align_max_iter: int = 10,
                               max_epi_search_steps: int = 1000, subpix_refinement: bool = True,
                               epi_search_edgelet_filtering: bool = True, epi_search_edgelet_max_angle: float = 0.7):
    """Batched Matcher::findEpipolarMatchDirect (matcher.h:113-123) for S queries.  Returns
    (ok [S] i32, depth [S] f64, px_cur [S,2], search_level [S] i32)."""
    lib = capi.load()
    S = ftr.px.shape[0]
    dev = store.device
    opt = capi.DepthFilterOptions(0, 0, 0.0, int(align_1d), align_max_iter, max_epi_search_steps, int(subpix_refinement),
                                  int(epi_search_edgelet_filtering), n_pyr_levels, epi_search_edgelet_max_angle)
    ok = torch.zeros(S, dtype=torch.int32, device=dev)
    depth = torch.zeros(S, dtype=torch.float64, device=dev)
    px = torch.zeros(S, 2, dtype=torch.float64, device=dev)
    lvl = torch.zeros(S, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.svo_hip_match_workspace_bytes(S), dtype=torch.uint8, device=dev)
    c, fr, ft = capi.camera(cam), frames.struct(), ftr.struct()
    capi.check(lib.svo_hip_find_epipolar_match_direct(
        C.byref(store.layout), store.ptr, C.byref(c), C.byref(fr), S, _chk(cur_frame, torch.int32).data_ptr(), C.byref(ft),
        _chk(d_estimate, torch.float64).data_ptr(), _chk(d_min, torch.float64).data_ptr(), _chk(d_max, torch.float64).data_ptr(),
        C.byref(opt), ok.data_ptr(), depth.data_ptr(), px.data_ptr(), lvl.data_ptr(), ws.data_ptr(), ws.numel(),
        _stream_ptr(dev)), "svo_hip_find_epipolar_match_direct")
    return ok, depth, px, lvl


def reproject_points(cam, frames: FrameTable, cur_frame, pt_pos, cell_size: int, grid_n_cols: int, out=None):
    """Reprojector::reprojectPoint for M points: (cell [M] i32 (-1 = not in frame), px [M,2]).
    out: (cell, px) tensors to reuse."""
    lib = capi.load()
    M = pt_pos.shape[0]
    dev = pt_pos.device
    if out is not None:
        cell, px = out
    else:
        cell = torch.zeros(M, dtype=torch.int32, device=dev)
        px = torch.zeros(M, 2, dtype=torch.float64, device=dev)
    c, fr = capi.camera(cam), frames.struct()
    capi.check(lib.svo_hip_reproject_points(C.byref(c), C.byref(fr), M, _chk(cur_frame, torch.int32).data_ptr(),
                                            _chk(pt_pos, torch.float64).data_ptr(), cell_size, grid_n_cols,
                                            cell.data_ptr(), px.data_ptr(), _stream_ptr(dev)), "svo_hip_reproject_points")
    return cell, px


def compose_poses(A, B, out=None, out_index=None):
    """out[idx[i]] = A[i] * B[i] (SE3 product), e.g. T_f_w(cur) = T_cur_from_ref * T_f_w(ref)."""
    lib = capi.load()
    n = A.shape[0]
    if out is None:
        out = torch.empty_like(A)
    capi.check(lib.svo_hip_compose_poses(n, _chk(A, torch.float64).data_ptr(), _chk(B, torch.float64).data_ptr(),
                                         _chk(out, torch.float64).data_ptr(), _ptr(out_index), _stream_ptr(A.device)),
               "svo_hip_compose_poses")
    return out


def cam2world(cam, px, out=None):
    """Unit bearings of pixels px [n,2] (vk::PinholeCamera::cam2world)."""
    lib = capi.load()
    n = px.shape[0]
    if out is None:
        out = torch.empty(n, 3, dtype=torch.float64, device=px.device)
    c = capi.camera(cam)
    capi.check(lib.svo_hip_cam2world(C.byref(c), n, _chk(px, torch.float64).data_ptr(), out.data_ptr(),
                                     _stream_ptr(px.device)), "svo_hip_cam2world")
    return out


def select_matches(cam, cell, ok, px, level, pos, max_fts):
    """Reprojector::reprojectMap's cell loop over the results of M trials given in visiting order
    (svo/src/reprojector.cpp:131-139, 150-200): per cell the first trial that matched, at most max_fts + 1 of them.
    Returns (n [1] int32, sel, f, level_out, pos_out, has_point), the arrays min(M, max_fts + 1) long: the
    observations of the selected trials in Frame::fts_ order, as svo_hip_pose_optimize reads them."""
    lib = capi.load()
    M = cell.shape[0]
    dev = px.device
    cap = max(1, min(M, max_fts + 1))
    n = torch.zeros(1, dtype=torch.int32, device=dev)
    sel = torch.full((cap,), -1, dtype=torch.int32, device=dev)
    f = torch.zeros(cap, 3, dtype=torch.float64, device=dev)
    level_out = torch.zeros(cap, dtype=torch.int32, device=dev)
    pos_out = torch.zeros(cap, 3, dtype=torch.float64, device=dev)
    has_point = torch.zeros(cap, dtype=torch.uint8, device=dev)
    c = capi.camera(cam)
    capi.check(lib.svo_hip_select_matches(C.byref(c), M, _chk(cell, torch.int32).data_ptr(), _chk(ok, torch.int32).data_ptr(),
                                          _chk(px, torch.float64).data_ptr(), _chk(level, torch.int32).data_ptr(),
                                          _chk(pos, torch.float64).data_ptr(), int(max_fts), n.data_ptr(), sel.data_ptr(),
                                          f.data_ptr(), level_out.data_ptr(), pos_out.data_ptr(), has_point.data_ptr(),
                                          None, 0, _stream_ptr(dev)), "svo_hip_select_matches")
    return n, sel, f, level_out, pos_out, has_point


# ---- pose_optimizer ---------------------------------------------------------------------
@dataclass
class PoseOptResult:
    T_f_w: torch.Tensor       # [B,12]
    Cov: torch.Tensor         # [B,36]
    stats: torch.Tensor       # [B,4] estimated_scale, error_init, error_final, num_obs
    ran: torch.Tensor         # [B] i32
    has_point: torch.Tensor   # [B,ns] u8 after pruning


def optimize_gauss_newton(cam, n, f, level, pos, has_point, T_f_w, reproj_thresh: float = 2.0, n_iter: int = 10,
                          ordered: bool = False, out: PoseOptResult | None = None, deferred: bool = False) -> PoseOptResult:
    """pose_optimizer::optimizeGaussNewton for B frames (reproj_thresh = Config::poseOptimThresh(),
    n_iter = Config::poseOptimNumIter(), frame_handler_mono.cpp:163-165).  ordered=True runs the
    kernel that adds the normal equations in the reference's observation order (the checker)."""
    lib = capi.load()
    B, ns = f.shape[0], f.shape[1]
    dev = f.device
    _chk(n, torch.int32); _chk(f, torch.float64); _chk(level, torch.int32); _chk(pos, torch.float64)
    _chk(has_point, torch.uint8); _chk(T_f_w, torch.float64)
    if out is not None:  # reuse a result block: pose and flags are in/out for the kernel
        res = out
        res.T_f_w.copy_(T_f_w)
        res.has_point.copy_(has_point)
    else:
        res = PoseOptResult(T_f_w.clone(), torch.zeros(B, 36, dtype=torch.float64, device=dev),
                            torch.zeros(B, 4, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev),
                            has_point.clone())
    c = capi.camera(cam)
    # deferred: svo_hip_pose_optimize_deferred -- frames the wave kernel hands over come back untouched with ran == 2
    fn = lib.svo_hip_pose_optimize_ordered if ordered else (lib.svo_hip_pose_optimize_deferred if deferred else lib.svo_hip_pose_optimize)
    capi.check(fn(C.byref(c), B, n.data_ptr(), ns, f.data_ptr(), level.data_ptr(), pos.data_ptr(),
                  res.has_point.data_ptr(), reproj_thresh, n_iter, res.T_f_w.data_ptr(),
                  res.Cov.data_ptr(), res.stats.data_ptr(), res.ran.data_ptr(), _stream_ptr(dev)),
               "svo_hip_pose_optimize")
    return res


# ---- Point::optimize ----------------------------------------------------------------------
def point_optimize(frames: FrameTable, obs_ptr, obs_frame, obs_f, pos, n_iter: int = 5):
    """Point::optimize(n_iter) for P points; returns the optimised positions [P,3]."""
    lib = capi.load()
    out = _chk(pos, torch.float64).clone()
    fr = frames.struct()
    capi.check(lib.svo_hip_point_optimize(C.byref(fr), pos.shape[0], _chk(obs_ptr, torch.int32).data_ptr(),
                                          _chk(obs_frame, torch.int32).data_ptr(), _chk(obs_f, torch.float64).data_ptr(),
                                          n_iter, out.data_ptr(), _stream_ptr(pos.device)), "svo_hip_point_optimize")
    return out


# ---- DepthFilter ----------------------------------------------------------------------------
@dataclass
class SeedSet:
    """SoA of svo::Seed (depth_filter.h:35-51); updated in place."""
    a: torch.Tensor
    b: torch.Tensor
    mu: torch.Tensor
    z_range: torch.Tensor
    sigma2: torch.Tensor
    batch_id: torch.Tensor

    def struct(self) -> capi.Seeds:
        for t in (self.a, self.b, self.mu, self.z_range, self.sigma2):
            _chk(t, torch.float32)
        _chk(self.batch_id, torch.int32)
        return capi.Seeds(self.a.data_ptr(), self.b.data_ptr(), self.mu.data_ptr(), self.z_range.data_ptr(),
                          self.sigma2.data_ptr(), self.batch_id.data_ptr())


class DepthFilter:
    """Batched svo::DepthFilter::updateSeeds.  Options: DepthFilter::Options + Matcher::Options."""

    def __init__(self, max_n_kfs: int = 3, seed_convergence_sigma2_thresh: float = 200.0, n_pyr_levels: int = 3,
                 align_1d: bool = False, align_max_iter: int = 10, max_epi_search_steps: int = 1000,
                 subpix_refinement: bool = True, epi_search_edgelet_filtering: bool = True,
                 epi_search_edgelet_max_angle: float = 0.7):
        self.opt = capi.DepthFilterOptions(max_n_kfs, 0, seed_convergence_sigma2_thresh, int(align_1d), align_max_iter,
                                           max_epi_search_steps, int(subpix_refinement),
                                           int(epi_search_edgelet_filtering), n_pyr_levels, epi_search_edgelet_max_angle)
        self.lib = capi.load()
        self._ws = _Workspace()

    def update_seeds(self, store: PyramidStore, cam, frames: FrameTable, cur_frame, ftr: FeatureSet, seeds: SeedSet,
                     batch_counter: int, out=None):
        """Returns (status [S] i32 capi.SEED_*, xyz_world [S,3], px_cur [S,2]); out: those three to reuse."""
        S = seeds.mu.shape[0]
        dev = store.device
        self.opt.batch_counter = batch_counter
        if out is not None:
            status, xyz, px = out
        else:
            status = torch.zeros(S, dtype=torch.int32, device=dev)
            xyz = torch.zeros(S, 3, dtype=torch.float64, device=dev)
            px = torch.zeros(S, 2, dtype=torch.float64, device=dev)
        ws = self._ws.get(self.lib, S, dev)
        self.last_workspace = ws
        c, fr, ft, sd = capi.camera(cam), frames.struct(), ftr.struct(), seeds.struct()
        capi.check(self.lib.svo_hip_update_seeds(C.byref(store.layout), store.ptr, C.byref(c), C.byref(fr), S,
                                                 _chk(cur_frame, torch.int32).data_ptr(), C.byref(ft), C.byref(sd),
                                                 C.byref(self.opt), status.data_ptr(), xyz.data_ptr(), px.data_ptr(),
                                                 ws.data_ptr(), ws.numel(), _stream_ptr(dev)), "svo_hip_update_seeds")
        return status, xyz, px

    def update_seeds_resident(self, store: PyramidStore, cam, frames: FrameTable, cur_frame: int, slot_of, store_ftr: FeatureSet,
                              store_seeds: SeedSet, batch_counter: int):
        """svo_hip_update_seeds_resident (row N2): the S seeds at slots `slot_of` [S] i32 of a resident store (store_ftr /
        store_seeds: columns indexed by slot, FeatureSet.frame holding frame-table indices), all updated with frame
        `cur_frame` of the table.  Returns (status [S], xyz_world [S,3], px_cur [S,2], state [4,S] = a, b, mu, sigma2)."""
        S = slot_of.shape[0]
        dev = store.device
        self.opt.batch_counter = batch_counter
        status = torch.zeros(S, dtype=torch.int32, device=dev)
        xyz = torch.zeros(S, 3, dtype=torch.float64, device=dev)
        px = torch.zeros(S, 2, dtype=torch.float64, device=dev)
        state = torch.zeros(4, S, dtype=torch.float32, device=dev)
        ws = self._ws.get(self.lib, S, dev)
        c, fr, ft, sd = capi.camera(cam), frames.struct(), store_ftr.struct(), store_seeds.struct()
        capi.check(self.lib.svo_hip_update_seeds_resident(
            C.byref(store.layout), store.ptr, C.byref(c), C.byref(fr), int(cur_frame), S, _chk(slot_of, torch.int32).data_ptr(),
            C.byref(ft), C.byref(sd), C.byref(self.opt), status.data_ptr(), xyz.data_ptr(), px.data_ptr(), state.data_ptr(),
            ws.data_ptr(), ws.numel(), _stream_ptr(dev)), "svo_hip_update_seeds_resident")
        return status, xyz, px, state

    @staticmethod
    def seed_store_patch(slots, src_ftr: FeatureSet, src_seeds: SeedSet, store_ftr: FeatureSet, store_seeds: SeedSet):
        """svo_hip_seed_store_patch: record i of src_* goes to slot slots[i] of the store's columns."""
        lib = capi.load()
        p = capi.SeedPatch(int(slots.shape[0]), 0, _chk(slots, torch.int32).data_ptr(), src_ftr.struct(), src_seeds.struct())
        ft, sd = store_ftr.struct(), store_seeds.struct()
        capi.check(lib.svo_hip_seed_store_patch(C.byref(p), C.byref(ft), C.byref(sd), _stream_ptr(slots.device)), "svo_hip_seed_store_patch")

    @staticmethod
    def update_seed(x, tau2, seeds: SeedSet):
        """static DepthFilter::updateSeed(x, tau2, seed) for S independent measurements."""
        lib = capi.load()
        sd = seeds.struct()
        capi.check(lib.svo_hip_update_seed_batch(x.shape[0], _chk(x, torch.float32).data_ptr(),
                                                 _chk(tau2, torch.float32).data_ptr(), C.byref(sd),
                                                 _stream_ptr(x.device)), "svo_hip_update_seed_batch")
