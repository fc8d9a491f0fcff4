"""Device-resident image pyramids ("pyramid store") -- host mirror of
svo::Frame::img_pyr_ / frame_utils::createImgPyramid (svo/src/frame.cpp:156-165).

One slot per frame; levels sit at fixed byte offsets, each cut into 16 x 8 pixel
tiles of one 128-byte line (see svo_hip_pyr_layout in include/svo_hip.h): the
store is filled and read through the library only.  torch owns the HBM
allocation; the K0 kernel in libsvo_hip.so fills the levels.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import capi


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class PyramidStore:
    def __init__(self, width: int, height: int, n_levels: int, n_slots: int, device="cuda:0",
                 halfsample: int = capi.HALFSAMPLE_AUTO):
        self.lib = capi.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise capi.SvoHipError("PyramidStore needs a HIP device; there is no CPU path")
        self.layout = capi.pyr_layout(width, height, n_levels)
        self.n_slots = n_slots
        self.halfsample = halfsample
        self.buf = torch.zeros(capi.pyr_store_bytes(self.layout, n_slots), dtype=torch.uint8, device=self.device)
        # packed level-0 image on its way from the host into the tiled store (upload())
        self._stage = torch.empty(width * height, dtype=torch.uint8, device=self.device)

    @property
    def ptr(self) -> int:
        return self.buf.data_ptr()

    @property
    def n_levels(self) -> int:
        return self.layout.n_levels

    def load_images(self, images: torch.Tensor, first_slot: int = 0, build: bool = True, fused: bool = True,
                    tile: int = 0) -> None:
        """images: uint8 [n,h,w] on the device (contiguous).  fused (default): level 0 and all
        further levels in one pass (svo_hip_pyramid_build_from_images).  tile: level-0 tile of the
        fused kernel (svo_hip_pyramid_build_tiled; 0 = chosen by image size)."""
        assert images.dtype == torch.uint8 and images.is_cuda and images.is_contiguous()
        n, h, w = images.shape
        assert h == self.layout.h[0] and w == self.layout.w[0] and first_slot + n <= self.n_slots
        if build and fused:
            capi.check(self.lib.svo_hip_pyramid_build_tiled(C.byref(self.layout), self.ptr, first_slot, n,
                                                            images.data_ptr(), h * w, w, self.halfsample, tile,
                                                            _stream_ptr(self.device)),
                       "svo_hip_pyramid_build_tiled")
            return
        capi.check(self.lib.svo_hip_pyramid_load_level0(C.byref(self.layout), self.ptr, first_slot, n,
                                                        images.data_ptr(), h * w, w, _stream_ptr(self.device)),
                   "svo_hip_pyramid_load_level0")
        if build:
            self.build(first_slot, n, tile)

    def upload(self, slot: int, image: np.ndarray, build: bool = True) -> None:
        """image: uint8 [h,w] host array (a new camera frame)."""
        image = np.ascontiguousarray(image, dtype=np.uint8)
        assert image.shape == (self.layout.h[0], self.layout.w[0])
        if build:  # H2D + one kernel: tiled level 0 and every further level
            capi.check(self.lib.svo_hip_pyramid_upload_build(C.byref(self.layout), self.ptr, slot, image.ctypes.data,
                                                             image.shape[1], self.halfsample, self._stage.data_ptr(),
                                                             _stream_ptr(self.device)),
                       "svo_hip_pyramid_upload_build")
        else:
            capi.check(self.lib.svo_hip_pyramid_upload_level0(C.byref(self.layout), self.ptr, slot,
                                                              image.ctypes.data, image.shape[1], self._stage.data_ptr(),
                                                              _stream_ptr(self.device)),
                       "svo_hip_pyramid_upload_level0")
        torch.cuda.current_stream(self.device).synchronize()  # host buffer and staging may be reused

    def build(self, first_slot: int = 0, n_slots: int | None = None, tile: int = 0) -> None:
        n = self.n_slots - first_slot if n_slots is None else n_slots
        capi.check(self.lib.svo_hip_pyramid_build_tiled(C.byref(self.layout), self.ptr, first_slot, n, None, 0, 0,
                                                        self.halfsample, tile, _stream_ptr(self.device)),
                   "svo_hip_pyramid_build_tiled")

    def build_per_level(self, first_slot: int = 0, n_slots: int | None = None) -> None:
        """The pre-fusion builder (one launch per level); A/B timing and tests only."""
        n = self.n_slots - first_slot if n_slots is None else n_slots
        capi.check(self.lib.svo_hip_pyramid_build_per_level(C.byref(self.layout), self.ptr, first_slot, n,
                                                            self.halfsample, _stream_ptr(self.device)),
                   "svo_hip_pyramid_build_per_level")

    def level(self, slot: int, level: int) -> np.ndarray:
        out = np.zeros((self.layout.h[level], self.layout.w[level]), dtype=np.uint8)
        capi.check(self.lib.svo_hip_pyramid_download_level(C.byref(self.layout), self.ptr, slot, level,
                                                           out.ctypes.data, _stream_ptr(self.device)),
                   "svo_hip_pyramid_download_level")
        return out

    def bytes_per_pyramid(self) -> int:
        return sum(self.layout.w[i] * self.layout.h[i] for i in range(self.layout.n_levels))
