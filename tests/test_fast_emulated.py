"""Row N4 without a GPU: svo_hip_fast_detect of the host-emulated library (tests/emu_build.py: fast_detect.hip compiled
for the CPU through tests/host/hip_emu.h) -- FAST-9 score, 3x3 non-maximum suppression, Shi-Tomasi score and the per-cell
maximum through 64-bit atomicMax keys -- against the oracle's FastDetector::detect: corners, levels and scores identical in
every bit, with and without grid occupancy, on rendered frames and on a corner-dense noise image."""
import ctypes as C

import numpy as np
import pytest

from oracle import pytrack
from rpg_svo_amd import capi, synth
from helpers import FUZZ, fuzz_rng


@pytest.fixture(scope="module")
def emu():
    from emu_build import build_emulated
    return build_emulated(())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)  # (the pointer object keeps the array alive)


def detect(emu, imgs, n_pyr, levels, cell, occ, thresh=20.0):
    n, h, w = imgs.shape
    layout = capi.pyr_layout(w, h, n_pyr)
    store = np.zeros(capi.pyr_store_bytes(layout, n), np.uint8)
    imgs = np.ascontiguousarray(imgs)
    assert emu.svo_hip_pyramid_build_tiled(C.byref(layout), _p(store), 0, n, _p(imgs), C.c_longlong(h * w), w, capi.HALFSAMPLE_AUTO, 0, None) == 0
    cols, rows = -(-w // cell), -(-h // cell)
    nc = cols * rows
    emu.svo_hip_fast_workspace_bytes.restype = C.c_size_t
    ws = np.full(emu.svo_hip_fast_workspace_bytes(C.byref(layout), n, nc), 0xFF, np.uint8)   # (poisoned: NaN / -1 to whoever reads scratch it did not write)
    slots = np.arange(n, dtype=np.int32)
    xy, lvl, sc = np.zeros((n, nc, 2), np.int32), np.zeros((n, nc), np.int32), np.zeros((n, nc), np.float32)
    rc = emu.svo_hip_fast_detect(C.byref(layout), _p(store), n, _p(slots), levels, 20, cell, cols, rows, None if occ is None else _p(occ),
                                 C.c_double(thresh), _p(xy), _p(lvl), _p(sc), _p(ws), C.c_size_t(ws.size), None)
    assert rc == 0, rc
    return xy, lvl, sc, cols, rows


@pytest.mark.parametrize("w,h,f,levels,cell", [(376, 240, 160.0, 3, 30), (320, 240, 200.0, 4, 25)])
def test_emulated_fast_detect_bit_exact(emu, oracle, w, h, f, levels, cell):
    cam = synth.Camera(w, h, f, f, w / 2.0, h / 2.0)
    tex = synth.make_texture(seed=12345)
    T = synth.make_trajectory(3, seed=7 + FUZZ, max_step=0.03, max_rot_deg=0.5)
    imgs = synth.render(tex, T, cam).numpy()
    rng = fuzz_rng(1)
    imgs[2] = rng.integers(0, 256, size=imgs[2].shape, dtype=np.uint8)  # corner-dense stress image
    n_pyr = max(levels, 4)
    cols, rows = -(-w // cell), -(-h // cell)
    occ = (rng.uniform(size=(3, cols * rows)) < 0.25).astype(np.uint8)
    occ[0] = 0
    xy, lvl, sc, cols, rows = detect(emu, imgs, n_pyr, levels, cell, occ)
    for i in range(3):
        pyr = oracle.create_img_pyramid(imgs[i], n_pyr)
        exy, elvl, esc, n = pytrack.fast_detect_grid(pyr, levels, cell, cols, rows, occ[i], 20, 20.0)
        assert n > 20
        assert np.array_equal(sc[i].view(np.uint32), esc.view(np.uint32)), f"image {i}: scores differ"
        assert np.array_equal(xy[i], exy) and np.array_equal(lvl[i], elvl)


def test_emulated_fast_detect_empty_and_full_occupancy(emu):
    flat = np.full((2, 240, 320), 127, dtype=np.uint8)
    flat[1] = fuzz_rng(2).integers(0, 256, size=(240, 320), dtype=np.uint8)
    xy, lvl, sc, cols, rows = detect(emu, flat, 3, 3, 30, None)
    assert (lvl[0] == -1).all() and (xy[0] == -1).all() and (sc[0] == 20.0).all()   # textureless: no corner
    assert (lvl[1] >= 0).sum() > 50
    full = np.ones((2, cols * rows), np.uint8)
    xy, lvl, sc, _, _ = detect(emu, flat, 3, 3, 30, full)
    assert (lvl == -1).all()                                                        # every cell occupied
