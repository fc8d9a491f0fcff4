"""K0 parity: GPU pyramid vs oracle vk::halfSample restatement, bit-exact -- through the fused
single-pass builder (from packed images and from a level 0 already in the store) and the
one-launch-per-level builder it replaced."""
import numpy as np
import pytest
import torch
from helpers import FUZZ, fuzz_rng

pytestmark = pytest.mark.gpu

SHAPES = [(640, 480, 4), (752, 480, 5), (1280, 960, 5), (70, 50, 3), (67, 35, 2), (131, 197, 4), (640, 480, 6),
          (752, 480, 1), (16, 16, 2)]


def _check(store, oracle, imgs, levels, mode, first):
    from oracle import pytrack
    trk_ref = pytrack.Track("ref") if pytrack.ref_available() else None
    for i in range(len(imgs)):
        ref = oracle.create_img_pyramid(imgs[i], levels, mode)
        if trk_ref is not None:  # frame_utils::createImgPyramid of the reference itself (oracle/_ref)
            rr = trk_ref.create_img_pyramid(imgs[i], levels, mode)
            assert all(np.array_equal(a, b) for a, b in zip(ref, rr))
        for l in range(levels):
            got = store.level(first + i, l)
            assert got.shape == ref[l].shape
            assert np.array_equal(got, ref[l]), f"slot {i} level {l} differs"


@pytest.mark.parametrize("w,h,levels", SHAPES)
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_pyramid_bit_exact(oracle, gpu_device, hip_lib, w, h, levels, mode):
    """One case per image shape and half-sample flavour; every level-0 tile of the fused kernel inside it."""
    from rpg_svo_amd.pyramid import PyramidStore
    rng = fuzz_rng(w * 7 + h + mode)
    imgs = rng.integers(0, 256, size=(3, h, w), dtype=np.uint8)
    dimgs = torch.from_numpy(imgs).to(gpu_device)
    for tile in (128, 256, 257, 512):
        # (a) fused, level 0 filled from the packed images in the same pass
        store = PyramidStore(w, h, levels, 4, device=gpu_device, halfsample=mode)
        store.load_images(dimgs, first_slot=1, tile=tile)
        _check(store, oracle, imgs, levels, mode, 1)
        # (b) level 0 copied first, fused build from the store
        store2 = PyramidStore(w, h, levels, 4, device=gpu_device, halfsample=mode)
        store2.load_images(dimgs, first_slot=0, fused=False, tile=tile)
        _check(store2, oracle, imgs, levels, mode, 0)
    # (c) the per-level builder
    store3 = PyramidStore(w, h, levels, 4, device=gpu_device, halfsample=mode)
    store3.load_images(dimgs, first_slot=0, build=False)
    store3.build_per_level(0, 3)
    _check(store3, oracle, imgs, levels, mode, 0)
    # padding bytes of the store stay out of the way: slot 3 / slot 0 of (a) untouched
    assert int(store.buf[: store.layout.slot_bytes].sum().item()) == 0


def test_unaligned_source_rows(oracle, gpu_device):
    """Packed source whose rows are not 16-byte aligned (width 67): the fused loader takes the
    byte path and still fills level 0 and the levels above it exactly."""
    from rpg_svo_amd.pyramid import PyramidStore
    rng = fuzz_rng(9)
    imgs = rng.integers(0, 256, size=(5, 45, 67), dtype=np.uint8)
    store = PyramidStore(67, 45, 3, 5, device=gpu_device)
    store.load_images(torch.from_numpy(imgs).to(gpu_device))
    _check(store, oracle, imgs, 3, oracle.HALFSAMPLE_AUTO, 0)


def test_upload_path_matches_load_path(oracle, gpu_device):
    from rpg_svo_amd.pyramid import PyramidStore
    rng = fuzz_rng(5)
    img = rng.integers(0, 256, size=(480, 640), dtype=np.uint8)
    store = PyramidStore(640, 480, 4, 2, device=gpu_device)
    store.upload(0, img)
    store.load_images(torch.from_numpy(img[None]).to(gpu_device), first_slot=1)
    for l in range(4):
        assert np.array_equal(store.level(0, l), store.level(1, l))
    ref = oracle.create_img_pyramid(img, 4, oracle.HALFSAMPLE_AUTO)
    assert np.array_equal(store.level(0, 3), ref[3])


def test_host_uploads_through_a_stream_ordered_temporary(oracle, gpu_device, hip_lib):
    """svo_hip_pyramid_upload_level0 / upload_level / upload_build with d_staging = NULL (hipMallocAsync scratch), and
    one level uploaded on its own (the cv::Mat a direct caller hands to feature_alignment::align2D): the tiled store
    gives back the rows that went in."""
    import ctypes as C
    from rpg_svo_amd import capi
    from rpg_svo_amd.pyramid import PyramidStore, _stream_ptr
    rng = fuzz_rng(21)
    w, h = 188, 120
    store = PyramidStore(w, h, 3, 3, device=gpu_device)
    img = rng.integers(0, 256, size=(h, w + 5), dtype=np.uint8)  # row stride > width
    lvl1 = rng.integers(0, 256, size=(h // 2, w // 2), dtype=np.uint8)
    lib, L, st = hip_lib, store.layout, _stream_ptr(store.device)
    capi.check(lib.svo_hip_pyramid_upload_level0(C.byref(L), store.ptr, 0, img.ctypes.data, img.shape[1], None, st))
    capi.check(lib.svo_hip_pyramid_upload_level(C.byref(L), store.ptr, 0, 1, lvl1.ctypes.data, lvl1.shape[1], None, st))
    capi.check(lib.svo_hip_pyramid_upload_build(C.byref(L), store.ptr, 1, img.ctypes.data, img.shape[1], capi.HALFSAMPLE_AUTO, None, st))
    torch.cuda.synchronize()
    assert np.array_equal(store.level(0, 0), img[:, :w])
    assert np.array_equal(store.level(0, 1), lvl1)
    ref = oracle.create_img_pyramid(np.ascontiguousarray(img[:, :w]), 3, oracle.HALFSAMPLE_AUTO)
    for l in range(3):
        assert np.array_equal(store.level(1, l), ref[l])
