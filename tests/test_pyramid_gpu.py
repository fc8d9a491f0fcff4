"""K0 parity: GPU pyramid vs oracle vk::halfSample restatement, bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,levels", [(640, 480, 4), (752, 480, 5), (1280, 960, 5), (70, 50, 3), (67, 35, 2)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_pyramid_bit_exact(oracle, gpu_device, w, h, levels, mode):
    from rpg_svo_amd.pyramid import PyramidStore
    rng = np.random.default_rng(w * 7 + h + mode)
    imgs = rng.integers(0, 256, size=(3, h, w), dtype=np.uint8)
    store = PyramidStore(w, h, levels, 4, device=gpu_device, halfsample=mode)
    store.load_images(torch.from_numpy(imgs).to(gpu_device), first_slot=1)
    for i in range(3):
        ref = oracle.create_img_pyramid(imgs[i], levels, mode)
        for l in range(levels):
            got = store.level(1 + i, l)
            assert got.shape == ref[l].shape
            assert np.array_equal(got, ref[l]), f"slot {i} level {l} differs"


def test_upload_path_matches_load_path(oracle, gpu_device):
    from rpg_svo_amd.pyramid import PyramidStore
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(480, 640), dtype=np.uint8)
    store = PyramidStore(640, 480, 4, 2, device=gpu_device)
    store.upload(0, img)
    store.load_images(torch.from_numpy(img[None]).to(gpu_device), first_slot=1)
    for l in range(4):
        assert np.array_equal(store.level(0, l), store.level(1, l))
    ref = oracle.create_img_pyramid(img, 4, oracle.HALFSAMPLE_AUTO)
    assert np.array_equal(store.level(0, 3), ref[3])
