"""Odd shapes through the emulated kernels (tests/emu_build.py): image sizes that are no multiple of the store's 16 x 8 tiles,
of a workgroup's footprint or of anything else, levels down to a few pixels, grid cells larger than the image, frames with
9 ... 300 patches (every workgroup size of K1) -- each against the oracle, bit for bit where the GPU tests ask for that.  The
border arithmetic is where an out-of-bounds access would hide: scripts/emu_sanitize.sh runs this file under
AddressSanitizer, too."""
import numpy as np
import pytest

from helpers import make_batch, run_oracle
from oracle import pytrack
from rpg_svo_amd import capi, se3, synth
from test_fast_emulated import detect
from test_pyramid_emulated import HostStore, _check
from test_sparse_align_emulated import run_emulated


@pytest.fixture(scope="module")
def emu():
    from emu_build import build_emulated
    return build_emulated(())


@pytest.mark.parametrize("w,h", [(17, 16), (16, 16), (31, 33), (64, 8), (100, 37), (129, 65), (255, 254), (333, 17), (48, 480), (640, 2),
                                 (18, 130), (97, 97)])
def test_pyramid_of_any_shape(emu, oracle, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    for levels in (1, 2, 3, 4, 5):
        if (w >> (levels - 1)) < 1 or (h >> (levels - 1)) < 1:
            continue
        imgs = rng.integers(0, 256, size=(3, h, w), dtype=np.uint8)
        for path in ("fused", "from_store", "per_level", "upload"):
            st = HostStore(emu, w, h, levels, 4)
            st.buf[:] = 0xEE
            if path == "fused":
                st.load_images(imgs, first_slot=1)
            elif path == "from_store":
                st.load_images(imgs, first_slot=1, fused=False)
            elif path == "per_level":
                st.load_images(imgs, first_slot=1, build=False)
                st.build_per_level(1, 3)
            else:
                for i in range(3):
                    st.upload(1 + i, imgs[i])
            _check(st, oracle, imgs, levels, capi.HALFSAMPLE_AUTO, 1)


@pytest.mark.parametrize("w,h,levels,cell", [(64, 48, 1, 30), (97, 61, 2, 25), (130, 18, 2, 10), (33, 200, 3, 32), (255, 254, 4, 30),
                                             (48, 40, 3, 7), (160, 120, 3, 200), (31, 33, 1, 5)])
def test_fast_on_any_shape(emu, oracle, w, h, levels, cell):
    rng = np.random.default_rng(w + h)
    imgs = rng.integers(0, 256, size=(2, h, w), dtype=np.uint8)
    imgs[1] = (np.add.outer(np.arange(h) // 5, np.arange(w) // 7) % 2 * 200 + 20).astype(np.uint8)   # corners everywhere
    cols, rows = -(-w // cell), -(-h // cell)
    occ = (rng.uniform(size=(2, cols * rows)) < 0.2).astype(np.uint8)
    xy, lvl, sc, cols, rows = detect(emu, imgs, levels, levels, cell, occ)
    for i in range(2):
        pyr = oracle.create_img_pyramid(imgs[i], levels)
        exy, elvl, esc, n = pytrack.fast_detect_grid(pyr, levels, cell, cols, rows, occ[i], 20, 20.0)
        assert np.array_equal(sc[i].view(np.uint32), esc.view(np.uint32))
        assert np.array_equal(xy[i], exy) and np.array_equal(lvl[i], elvl)


@pytest.mark.parametrize("w,h,n,levels,lo,hi", [(160, 120, 9, 3, 0, 2), (200, 150, 65, 3, 1, 2), (322, 242, 129, 4, 0, 3),
                                                (328, 248, 257, 4, 2, 3), (336, 256, 300, 3, 0, 2)])
def test_sparse_align_on_any_shape(emu, oracle, w, h, n, levels, lo, hi):
    """(64-, 128-, 256- and 512-lane workgroups; a frame with a third of the patches in the middle of the batch)"""
    cam = synth.Camera(w, h, w * 0.6, w * 0.6, w / 2.0, h / 2.0)
    seq = synth.make_sequence(4, n, cam=cam, seed=n, margin=12, cell=max(8, int((w * h / n) ** 0.5 * 0.7)))
    b = make_batch(seq, [(0, 1), (1, 2), (2, 3)], levels)
    b.n[1] = max(6, n // 3)
    T_o, res_o, _ = run_oracle(oracle, b, hi, lo)
    T_h, ntr, iters, H, status = run_emulated(emu, b, hi, lo)
    assert se3.log_norm(T_h, T_o).max() <= 1e-4
    assert np.array_equal(ntr, np.array([r["n_tracked"] for r in res_o]))


@pytest.mark.parametrize("kind", ["pinhole", "atan"])
def test_matcher_and_depth_filter_on_an_odd_image_size(emu, oracle, kind):
    """find_match_direct and update_seeds (tests/test_track_emulated.py) on 438 x 410 images: 27.4 x 51.25 tiles of the store"""
    import test_track_emulated as tte
    cam = synth.Camera(438, 410, 260.0, 260.0, 219.0, 205.0) if kind == "pinhole" else \
        synth.Camera.atan(438, 410, 0.509326, 0.796651, 0.45905, 0.510056, 0.9320)
    scene = synth.make_track_scene(n_kf=4, n_feat=100, cam=cam)
    tte.test_emulated_find_match_direct(emu, oracle, scene)
    tte.test_emulated_update_seeds(emu, oracle, scene, 0, 1)
