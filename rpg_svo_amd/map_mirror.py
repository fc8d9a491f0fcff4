"""Device-resident mirror of the map for Reprojector::reprojectMap (row N2; include/svo_hip.h: svo_hip_map,
svo_hip_reproject_map).  Host mirror for replay workloads and tests: torch tensors hold the records, the C ABI does
the work.  The C++ drop-in (rpg_svo_amd/host/dropin/map_mirror.h) keeps the same records in sync with the reference's
pointer graph.

    m = MapMirror(capacity_points, capacity_obs, device)
    m.patch(index, pos, type, order, obs_begin, obs_count, obs_index=..., obs=...)    # entries that changed
    r = m.reproject(cam, frames_T, cur_frame, kf_rank, grid, first_cell, max_cells_with_trials)
    r.header / r.visit_point / r.trial_* ...   (device tensors; r.n_visits(), r.n_trials() synchronise)

There is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import capi
from .pyramid import _stream_ptr

TYPE_DELETED, TYPE_CANDIDATE, TYPE_UNKNOWN, TYPE_GOOD = 0, 1, 2, 3   # svo::Point::PointType (point.h:38-43)


@dataclass
class Grid:
    """Reprojector::Grid (reprojector.h:79-86) + the visiting rank of every cell (the inverse of grid_.cell_order)."""
    cell_size: int
    n_cols: int
    n_rows: int
    cell_rank: torch.Tensor   # [n_cells] i32, device

    @staticmethod
    def for_camera(width: int, height: int, cell_size: int, cell_order, device) -> "Grid":
        n_cols = -(-width // cell_size)
        n_rows = -(-height // cell_size)
        order = np.asarray(cell_order, dtype=np.int64)
        assert sorted(order.tolist()) == list(range(n_cols * n_rows))
        rank = np.empty(n_cols * n_rows, dtype=np.int32)
        rank[order] = np.arange(n_cols * n_rows, dtype=np.int32)
        return Grid(cell_size, n_cols, n_rows, torch.as_tensor(rank, device=device))

    def struct(self) -> capi.Grid:
        return capi.Grid(self.cell_size, self.n_cols, self.n_rows, self.n_cols * self.n_rows, self.cell_rank.data_ptr())


@dataclass
class Reprojection:
    header: torch.Tensor
    point_cell: torch.Tensor
    point_px: torch.Tensor
    kf_count: torch.Tensor
    visit_point: torch.Tensor
    visit_cell: torch.Tensor
    visit_trial: torch.Tensor
    trial_cur: torch.Tensor
    trial_pos: torch.Tensor
    trial_obs_begin: torch.Tensor
    trial_obs_end: torch.Tensor
    trial_cell: torch.Tensor
    trial_px: torch.Tensor

    def struct(self) -> capi.Reprojection:
        return capi.Reprojection(*[getattr(self, n[2:]).data_ptr() for n, _ in capi.Reprojection._fields_])

    def counts(self):
        """(status, points in frame, visits, trials, end_cell) -- synchronises"""
        h = self.header.cpu().numpy()
        return int(h[0]), int(h[1]), int(h[2]), int(h[3]), int(h[4])


class MapMirror:
    def __init__(self, capacity_points: int, capacity_obs: int, device):
        assert capacity_points <= 8192
        self.device = torch.device(device)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=self.device)
        P, O = capacity_points, capacity_obs
        self.n_points = 0
        self.n_obs = 0
        self.pos = z((P, 3), torch.float64)
        self.type = z(P, torch.int32)
        self.order = z(P, torch.int32)
        self.obs_begin = z(P, torch.int32)
        self.obs_count = z(P, torch.int32)
        self.obs_frame = z(O, torch.int32)
        self.obs_order = z(O, torch.int32)
        self.obs_level = z(O, torch.int32)
        self.obs_type = z(O, torch.uint8)
        self.obs_px = z((O, 2), torch.float64)
        self.obs_f = z((O, 3), torch.float64)
        self.obs_grad = z((O, 2), torch.float64)
        self.lib = capi.load()
        self._pending = None

    def struct(self) -> capi.Map:
        return capi.Map(self.n_points, self.n_obs, *[getattr(self, n[2:]).data_ptr() for n, _ in capi.Map._fields_[2:]])

    def obs_features(self) -> capi.Features:
        """the observation records as the svo_hip_features the match kernels read"""
        return capi.Features(self.obs_frame.data_ptr(), self.obs_level.data_ptr(), self.obs_type.data_ptr(), self.obs_px.data_ptr(),
                             self.obs_f.data_ptr(), self.obs_grad.data_ptr())

    def patch(self, index, pos, type, order, obs_begin, obs_count, obs_index=None, obs_frame=None, obs_order=None, obs_level=None,
              obs_type=None, obs_px=None, obs_f=None, obs_grad=None):
        """Queue the entries the next reproject() writes before it reads the map (host arrays)."""
        dev = self.device
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        index = np.asarray(index, dtype=np.int32)
        p = {"index": t(index, torch.int32), "pos": t(np.asarray(pos, dtype=np.float64).reshape(-1, 3), torch.float64),
             "type": t(type, torch.int32), "order": t(order, torch.int32), "obs_begin": t(obs_begin, torch.int32),
             "obs_count": t(obs_count, torch.int32)}
        if index.size:
            self.n_points = max(self.n_points, int(index.max()) + 1)
        n_obs = 0 if obs_index is None else len(obs_index)
        if n_obs:
            oi = np.asarray(obs_index, dtype=np.int32)
            self.n_obs = max(self.n_obs, int(oi.max()) + 1)
            p.update(obs_index=t(oi, torch.int32), obs_frame=t(obs_frame, torch.int32), obs_order=t(obs_order, torch.int32),
                     obs_level=t(obs_level, torch.int32), obs_type=t(obs_type, torch.uint8),
                     obs_px=t(np.asarray(obs_px, dtype=np.float64).reshape(-1, 2), torch.float64),
                     obs_f=t(np.asarray(obs_f, dtype=np.float64).reshape(-1, 3), torch.float64),
                     obs_grad=t(np.asarray(obs_grad, dtype=np.float64).reshape(-1, 2), torch.float64))
        assert self.n_points <= self.pos.shape[0] and self.n_obs <= self.obs_frame.shape[0], "MapMirror capacity exceeded"
        assert self._pending is None, "one patch per reproject()"
        self._pending = (p, int(index.size), n_obs)

    def reproject(self, cam, frames_T: torch.Tensor, cur_frame: int, kf_rank: torch.Tensor, grid: Grid, first_cell: int = 0,
                  max_cells_with_trials: int = 1 << 30, max_visits: int = 4096, max_trials: int = 4096) -> Reprojection:
        """Reprojector::reprojectMap up to the first findMatchDirect for one frame (svo_hip_reproject_map)."""
        dev = self.device
        n_frames = frames_T.shape[0]
        assert frames_T.dtype == torch.float64 and frames_T.is_cuda and kf_rank.dtype == torch.int32 and kf_rank.shape[0] == n_frames
        zi = lambda n: torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        zd = lambda n, k: torch.zeros((max(n, 1), k), dtype=torch.float64, device=dev)
        P = max(self.n_points, 1)
        r = Reprojection(zi(capi.REPROJ_HEADER), zi(P), zd(P, 2), zi(n_frames), zi(max_visits), zi(max_visits), zi(max_visits),
                         zi(max_trials), zd(max_trials, 3), zi(max_trials), zi(max_trials), zi(max_trials), zd(max_trials, 2))
        patch = None
        keep = None
        if self._pending is not None:
            p, n_pts, n_obs = self._pending
            self._pending = None
            keep = p
            ptr = lambda k: p[k].data_ptr() if k in p else None
            patch = capi.MapPatch(n_pts, n_obs, ptr("index"), ptr("pos"), ptr("type"), ptr("order"), ptr("obs_begin"), ptr("obs_count"),
                                  ptr("obs_index"), ptr("obs_order"),
                                  capi.Features(ptr("obs_frame"), ptr("obs_level"), ptr("obs_type"), ptr("obs_px"), ptr("obs_f"), ptr("obs_grad")))
        frames = capi.Frames(n_frames, 0, None, frames_T.data_ptr())
        mp, g, rs, c = self.struct(), grid.struct(), r.struct(), capi.camera(cam)
        capi.check(self.lib.svo_hip_reproject_map(C.byref(c), C.byref(frames), cur_frame, kf_rank.data_ptr(), C.byref(mp),
                                                  C.byref(patch) if patch is not None else None, C.byref(g), first_cell,
                                                  min(max_cells_with_trials, 1 << 30), max_visits, max_trials, C.byref(rs),
                                                  _stream_ptr(dev)), "svo_hip_reproject_map")
        r._keep = keep  # the patch tensors must outlive the launch
        return r
