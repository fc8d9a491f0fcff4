"""K7 parity: FastDetector::detect on the device vs the C restatement (bit-exact corners and
scores) and vs the reference's own feature_detection.cpp run on the shimmed FAST library."""
import numpy as np
import pytest
import torch

from rpg_svo_amd import synth
from helpers import FUZZ, fuzz_rng


def _images(n, cam, seed):
    tex = synth.make_texture(seed=12345)
    T = synth.make_trajectory(n, seed=seed + FUZZ, max_step=0.03, max_rot_deg=0.5)
    return synth.render(tex, T, cam).numpy()


def test_oracle_fast_matches_reference_detector(oracle):
    """Pins orc_fast_detect_grid against the reference's FastDetector (oracle/_ref)."""
    from oracle import pytrack
    if not pytrack.ref_available() and not pytrack.build_ref():
        pytest.skip("oracle/_ref not available")
    cam = synth.Camera(752, 480, 315.5, 315.5, 376.0, 240.0)
    imgs = _images(3, cam, 4)
    cell, n_levels = 30, 3
    cols, rows = -(-cam.width // cell), -(-cam.height // cell)
    rng = fuzz_rng(0)
    for i, img in enumerate(imgs):
        pyr = oracle.create_img_pyramid(img, 5)
        occ = (rng.uniform(size=cols * rows) < 0.3).astype(np.uint8) if i else None
        xy, lvl, sc, n = pytrack.fast_detect_grid(pyr, n_levels, cell, cols, rows, occ, 20, 20.0)
        px_ref, lvl_ref = pytrack.ref_fast_detect(pyr, cam, n_levels, cell, occ, 20.0)
        sel = sc > 20.0
        assert n == sel.sum() == len(px_ref) and n > 150
        assert np.array_equal(xy[sel].astype(np.float64), px_ref) and np.array_equal(lvl[sel], lvl_ref)
        if occ is not None:
            assert not sel[occ.astype(bool)].any()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,f,levels,cell", [(752, 480, 315.5, 3, 30), (640, 480, 400.0, 4, 25), (1280, 960, 800.0, 5, 30)])
def test_fast_detect_bit_exact(oracle, gpu_device, w, h, f, levels, cell):
    from oracle import pytrack
    from rpg_svo_amd.feature_detection import FastDetector
    from rpg_svo_amd.pyramid import PyramidStore
    cam = synth.Camera(w, h, f, f, w / 2.0, h / 2.0)
    imgs = _images(4, cam, 7)
    rng = fuzz_rng(1)
    imgs[3] = rng.integers(0, 256, size=imgs[3].shape, dtype=np.uint8)  # corner-dense stress image
    n_pyr = max(levels, 5) if (w, h) != (640, 480) else levels
    store = PyramidStore(w, h, n_pyr, 4, device=gpu_device)
    store.load_images(torch.from_numpy(imgs).to(gpu_device))
    det = FastDetector(w, h, cell, levels)
    occ = (rng.uniform(size=(4, det.n_cells)) < 0.25).astype(np.uint8)
    occ[0] = 0
    slots = torch.arange(4, dtype=torch.int32, device=gpu_device)
    xy, lvl, sc = det.detect(store, slots, 20.0, torch.from_numpy(occ).to(gpu_device))
    xy, lvl, sc = xy.cpu().numpy(), lvl.cpu().numpy(), sc.cpu().numpy()
    for i in range(4):
        pyr = oracle.create_img_pyramid(imgs[i], n_pyr)
        exy, elvl, esc, n = pytrack.fast_detect_grid(pyr, levels, cell, det.grid_n_cols, det.grid_n_rows, occ[i], 20, 20.0)
        assert n > 50
        assert np.array_equal(sc[i].view(np.uint32), esc.view(np.uint32)), f"image {i}: scores differ"
        assert np.array_equal(xy[i], exy) and np.array_equal(lvl[i], elvl)
        if pytrack.ref_available():
            # the reference's own FastDetector::detect (oracle/_ref) on the same pyramid: the features it
            # creates are the device's cells with score > threshold, in cell order
            px_ref, lvl_ref = pytrack.ref_fast_detect(pyr, cam, levels, cell, occ[i], 20.0)
            sel = sc[i] > 20.0
            assert np.array_equal(xy[i][sel].astype(np.float64), px_ref) and np.array_equal(lvl[i][sel], lvl_ref)


@pytest.mark.gpu
def test_fast_detect_empty_and_full_occupancy(gpu_device):
    from rpg_svo_amd.feature_detection import FastDetector
    from rpg_svo_amd.pyramid import PyramidStore
    flat = np.full((2, 480, 640), 127, dtype=np.uint8)
    rng = fuzz_rng(2)
    flat[1] = rng.integers(0, 256, size=(480, 640), dtype=np.uint8)
    store = PyramidStore(640, 480, 3, 2, device=gpu_device)
    store.load_images(torch.from_numpy(flat).to(gpu_device))
    det = FastDetector(640, 480, 30, 3)
    slots = torch.arange(2, dtype=torch.int32, device=gpu_device)
    full = torch.ones(2, det.n_cells, dtype=torch.uint8, device=gpu_device)
    xy, lvl, sc = det.detect(store, slots, 20.0)
    assert (lvl[0] == -1).all() and (xy[0] == -1).all() and (sc[0] == 20.0).all()   # textureless: no corner
    assert (lvl[1] >= 0).sum() > 100
    xy, lvl, sc = det.detect(store, slots, 20.0, full)
    assert (lvl == -1).all()                                                        # every cell occupied
