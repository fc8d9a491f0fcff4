"""Rows a8-a10 without a GPU: svo_hip_find_match_direct of the host-emulated library (tests/emu_build.py: matcher.hip and
feature_align.hip compiled for the CPU through tests/host/hip_emu.h) -- match_prepare, warp_kernel with its LDS regions
and same-wave hand-overs, align_kernel -- against the oracle's Matcher::findMatchDirect on the scene of the GPU test, with
the same requirements: verdicts, chosen observations, search levels and the 10 x 10 patches identical, refined pixels
identical in every bit, A to 1e-12.  Also with the queued opt-in builds, whose results must not differ from the default's."""
import ctypes as C

import numpy as np
import pytest

from helpers import camera_models
from oracle import pytrack
from rpg_svo_amd import capi, synth

VARIANTS = [(), ("PREP_LOAD_FIRST", "WARP_PACKED", "ALIGN_LOAD_FIRST", "ALIGN_G_F16")]


@pytest.fixture(scope="module", params=VARIANTS, ids=["default", "queued-variants"])
def emu(request):
    from emu_build import build_emulated
    return build_emulated(request.param)


@pytest.fixture(scope="module", params=["pinhole", "atan"])
def scene(request):
    return synth.make_track_scene(n_kf=4, n_feat=100, cam=camera_models()[request.param])


def _p(a):
    return C.c_void_p(a.ctypes.data)


def _store(emu, scene, n_levels=5):
    imgs = np.ascontiguousarray(scene.images.cpu().numpy())
    n, h, w = imgs.shape
    layout = capi.pyr_layout(w, h, n_levels)
    buf = np.zeros(capi.pyr_store_bytes(layout, n), np.uint8)
    rc = emu.svo_hip_pyramid_build_tiled(C.byref(layout), _p(buf), 0, n, _p(imgs), C.c_longlong(h * w), w, capi.HALFSAMPLE_AUTO, 0, None)
    assert rc == 0
    return layout, buf


def test_emulated_find_match_direct(emu, oracle, scene):
    orc = pytrack.Track("orc")
    pyrs = [orc.create_img_pyramid(im, 5) for im in scene.images.cpu().numpy()]
    T = scene.T_f_w.copy()
    T[scene.cur] = scene.T_cur_prior
    T = np.ascontiguousarray(T)
    layout, store = _store(emu, scene)
    n_frames = T.shape[0]
    slots = np.arange(n_frames, dtype=np.int32)
    frames = capi.Frames(n_frames, 0, slots.ctypes.data, T.ctypes.data)
    P = len(scene.obs)
    ptr = np.zeros(P + 1, np.int32)
    flat = []
    for i, o in enumerate(scene.obs):
        ptr[i + 1] = ptr[i] + len(o)
        flat.extend(o)
    c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
    o_frame, o_level = c([o[0] for o in flat], np.int32), c([o[3] for o in flat], np.int32)
    o_px, o_f = c([o[1] for o in flat], np.float64), c([o[2] for o in flat], np.float64)
    o_type, o_grad = c([o[4] for o in flat], np.uint8), c([o[5] for o in flat], np.float64)
    obs = capi.Features(o_frame.ctypes.data, o_level.ctypes.data, o_type.ctypes.data, o_px.ctypes.data, o_f.ctypes.data, o_grad.ctypes.data)
    cur = np.full(P, scene.cur, np.int32)
    pos = c(scene.pt_pos, np.float64)
    px = c(scene.px_init, np.float64).copy()
    ok, ref_obs, sl = np.zeros(P, np.int32), np.zeros(P, np.int32), np.zeros(P, np.int32)
    A, patches = np.zeros((P, 4)), np.zeros((P, 100), np.uint8)
    emu.svo_hip_match_workspace_bytes.restype = C.c_size_t
    ws = np.zeros(emu.svo_hip_match_workspace_bytes(P) + 256, np.uint8)
    cam = capi.camera(scene.cam)
    rc = emu.svo_hip_find_match_direct(C.byref(layout), _p(store), C.byref(cam), C.byref(frames), P, _p(cur), _p(pos), _p(ptr), C.byref(obs),
                                       5, 10, _p(px), _p(ok), _p(ref_obs), _p(sl), _p(A), _p(patches), _p(ws), C.c_size_t(ws.size), None)
    assert rc == 0, rc
    oframes = pytrack.make_frames(pyrs, T)
    opt = pytrack.matcher_options(n_pyr_levels=5)
    n_ok = n_edge = 0
    for i in range(P):
        o = [pytrack.make_feature(*x) for x in scene.obs[i]]
        o_ok, o_pxr, r = orc.find_match_direct(oframes, scene.cam, scene.cur, scene.pt_pos[i], o, scene.px_init[i], opt)
        assert bool(ok[i]) == o_ok, i
        assert ref_obs[i] - ptr[i] == r["ref_obs"], i
        if r["A_cur_ref"].any():
            assert sl[i] == r["search_level"]
            assert np.abs(A[i].reshape(2, 2) - r["A_cur_ref"]).max() < 1e-12
            assert np.array_equal(patches[i], r["patch_with_border"]), i
        assert np.array_equal(px[i], o_pxr), (i, px[i], o_pxr)
        n_ok += o_ok
        n_edge += o_ok and scene.obs[i][r["ref_obs"]][4] == 1
    assert n_ok > 200 and n_edge > 20
