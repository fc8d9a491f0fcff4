"""K0 (row N1) without a GPU: rpg_svo_amd/csrc/pyramid.hip compiled for the host through tests/host/hip_emu.h -- the C-ABI
entry points themselves, their kernels run by host threads -- against the oracle's halfSample / createImgPyramid, bit for bit,
like tests/test_pyramid_gpu.py on the device: the fused pass from packed images, the fused pass from the store, the
per-level builder, the upload paths, every level-0 tile of the fused kernel, the three half-sample flavours."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from rpg_svo_amd import capi
from helpers import FUZZ, fuzz_rng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu():
    from emu_build import build_emulated
    return build_emulated()


class HostStore:
    """rpg_svo_amd.pyramid.PyramidStore on host memory (same entry points, same arguments)"""

    def __init__(self, lib, w, h, n_levels, n_slots, halfsample=capi.HALFSAMPLE_AUTO):
        self.lib, self.layout, self.n_slots, self.halfsample = lib, capi.pyr_layout(w, h, n_levels), n_slots, halfsample
        self.buf = np.zeros(capi.pyr_store_bytes(self.layout, n_slots), np.uint8)

    def load_images(self, images, first_slot=0, build=True, fused=True, tile=0):
        n, h, w = images.shape
        images = np.ascontiguousarray(images)
        if build and fused:
            rc = self.lib.svo_hip_pyramid_build_tiled(C.byref(self.layout), C.c_void_p(self.buf.ctypes.data), first_slot, n,
                                                      C.c_void_p(images.ctypes.data), C.c_longlong(h * w), w, self.halfsample, tile, None)
            assert rc == 0, rc
            return
        rc = self.lib.svo_hip_pyramid_load_level0(C.byref(self.layout), C.c_void_p(self.buf.ctypes.data), first_slot, n,
                                                  C.c_void_p(images.ctypes.data), C.c_longlong(h * w), w, None)
        assert rc == 0, rc
        if build:
            rc = self.lib.svo_hip_pyramid_build_tiled(C.byref(self.layout), C.c_void_p(self.buf.ctypes.data), first_slot, n, None,
                                                      C.c_longlong(0), 0, self.halfsample, tile, None)
            assert rc == 0, rc

    def build_per_level(self, first_slot, n):
        rc = self.lib.svo_hip_pyramid_build_per_level(C.byref(self.layout), C.c_void_p(self.buf.ctypes.data), first_slot, n, self.halfsample, None)
        assert rc == 0, rc

    def upload(self, slot, image):
        image = np.ascontiguousarray(image, np.uint8)
        rc = self.lib.svo_hip_pyramid_upload_build(C.byref(self.layout), C.c_void_p(self.buf.ctypes.data), slot, C.c_void_p(image.ctypes.data),
                                                   image.shape[1], self.halfsample, None, None)
        assert rc == 0, rc

    def level(self, slot, level):
        out = np.zeros((self.layout.h[level], self.layout.w[level]), np.uint8)
        rc = self.lib.svo_hip_pyramid_download_level(C.byref(self.layout), C.c_void_p(self.buf.ctypes.data), slot, level,
                                                     C.c_void_p(out.ctypes.data), None)
        assert rc == 0, rc
        return out


def _check(store, oracle, imgs, levels, mode, first):
    for i in range(imgs.shape[0]):
        ref = oracle.create_img_pyramid(imgs[i], levels, mode)
        for l in range(levels):
            got = store.level(first + i, l)
            assert got.shape == ref[l].shape and np.array_equal(got, ref[l]), f"slot {i} level {l} differs"


@pytest.mark.parametrize("w,h,levels", [(640, 480, 4), (752, 480, 5), (322, 242, 3)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_emulated_pyramid_bit_exact(emu, oracle, w, h, levels, mode):
    rng = fuzz_rng(w * 7 + h + mode)
    imgs = rng.integers(0, 256, size=(2, h, w), dtype=np.uint8)
    for tile in (128, 257):
        store = HostStore(emu, w, h, levels, 3, halfsample=mode)
        store.load_images(imgs, first_slot=1, tile=tile)                 # fused, level 0 filled from the packed images
        _check(store, oracle, imgs, levels, mode, 1)
        assert int(store.buf[: store.layout.slot_bytes].sum()) == 0      # slot 0 untouched
    store2 = HostStore(emu, w, h, levels, 2, halfsample=mode)
    store2.load_images(imgs, first_slot=0, fused=False, tile=256)        # level 0 copied first, fused build from the store
    _check(store2, oracle, imgs, levels, mode, 0)
    store3 = HostStore(emu, w, h, levels, 2, halfsample=mode)
    store3.load_images(imgs, first_slot=0, build=False)
    store3.build_per_level(0, 2)                                         # the per-level builder
    _check(store3, oracle, imgs, levels, mode, 0)


def test_emulated_unaligned_rows_and_upload(emu, oracle):
    rng = fuzz_rng(9)
    imgs = rng.integers(0, 256, size=(2, 45, 67), dtype=np.uint8)        # rows not 16-byte aligned: the byte path of the loader
    store = HostStore(emu, 67, 45, 3, 2)
    store.load_images(imgs)
    _check(store, oracle, imgs, 3, oracle.HALFSAMPLE_AUTO, 0)
    img = rng.integers(0, 256, size=(480, 640), dtype=np.uint8)
    up = HostStore(emu, 640, 480, 4, 1)
    up.upload(0, img)                                                    # the host upload path (stream-ordered temporary)
    _check(up, oracle, img[None], 4, oracle.HALFSAMPLE_AUTO, 0)
