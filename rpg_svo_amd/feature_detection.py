"""Batched host mirror of svo::feature_detection::FastDetector (svo/include/svo/
feature_detection.h:60-100) over svo_hip_fast_detect (K7).  Device-resident tensors only."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import capi
from .pyramid import PyramidStore, _stream_ptr


class FastDetector:
    """FastDetector(img_width, img_height, cell_size, n_pyr_levels); detect() for a batch of
    store slots.  Grid occupancy (setExistingFeatures / setGridOccpuancy) is a [n, cells] u8 tensor."""

    def __init__(self, img_width: int, img_height: int, cell_size: int, n_pyr_levels: int, fast_threshold: int = 20):
        self.cell_size = cell_size
        self.n_pyr_levels = n_pyr_levels
        self.grid_n_cols = math.ceil(img_width / cell_size)
        self.grid_n_rows = math.ceil(img_height / cell_size)
        self.fast_threshold = fast_threshold
        self.lib = capi.load()
        self._ws = None

    @property
    def n_cells(self) -> int:
        return self.grid_n_cols * self.grid_n_rows

    def occupancy_from_features(self, px: torch.Tensor) -> torch.Tensor:
        """AbstractDetector::setExistingFeatures for px [n, m, 2] (level-0 pixels; NaN rows ignored)."""
        n, m, _ = px.shape
        occ = torch.zeros(n, self.n_cells, dtype=torch.uint8, device=px.device)
        ok = ~torch.isnan(px[..., 0])
        k = (px[..., 1] / self.cell_size).to(torch.int64) * self.grid_n_cols + (px[..., 0] / self.cell_size).to(torch.int64)
        k = torch.where(ok, k, torch.zeros_like(k))
        occ.scatter_(1, k, ok.to(torch.uint8))
        return occ

    def detect(self, store: PyramidStore, slots: torch.Tensor, detection_threshold: float = 20.0, occupancy=None):
        """-> (xy [n,cells,2] i32 level-0 px (-1 = empty cell), level [n,cells] i32, score [n,cells] f32).
        A Feature exists where score > detection_threshold."""
        assert slots.dtype == torch.int32 and slots.is_cuda
        n = slots.shape[0]
        dev = store.device
        need = self.lib.svo_hip_fast_workspace_bytes(C.byref(store.layout), n, self.n_cells)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        xy = torch.empty(n, self.n_cells, 2, dtype=torch.int32, device=dev)
        level = torch.empty(n, self.n_cells, dtype=torch.int32, device=dev)
        score = torch.empty(n, self.n_cells, dtype=torch.float32, device=dev)
        if occupancy is not None:
            assert occupancy.dtype == torch.uint8 and occupancy.shape == (n, self.n_cells) and occupancy.is_contiguous()
        capi.check(self.lib.svo_hip_fast_detect(
            C.byref(store.layout), store.ptr, n, slots.data_ptr(), self.n_pyr_levels, self.fast_threshold, self.cell_size,
            self.grid_n_cols, self.grid_n_rows, None if occupancy is None else occupancy.data_ptr(), detection_threshold,
            xy.data_ptr(), level.data_ptr(), score.data_ptr(), self._ws.data_ptr(), self._ws.numel(), _stream_ptr(dev)),
            "svo_hip_fast_detect")
        return xy, level, score
