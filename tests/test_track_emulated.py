"""Rows a8-a10 without a GPU: svo_hip_find_match_direct of the host-emulated library (tests/emu_build.py: matcher.hip and
feature_align.hip compiled for the CPU through tests/host/hip_emu.h) -- match_prepare, warp_kernel with its LDS regions
and same-wave hand-overs, align_kernel -- against the oracle's Matcher::findMatchDirect on the scene of the GPU test, with
the same requirements: verdicts, chosen observations, search levels and the 10 x 10 patches identical, refined pixels
identical in every bit, A to 1e-12.  Also with the queued opt-in builds, whose results must not differ from the default's."""
import ctypes as C

import numpy as np
import pytest

from helpers import FUZZ, camera_models, fuzz_rng
from oracle import pytrack
from rpg_svo_amd import capi, synth


@pytest.fixture(scope="module", params=[0], ids=["default"])
def emu(request):
    from emu_build import build_emulated
    from emu_build import BUILDS
    return build_emulated(BUILDS[request.param])


@pytest.fixture(scope="module", params=["pinhole", "atan"])
def scene(request):
    return synth.make_track_scene(n_kf=4, n_feat=100, cam=camera_models()[request.param], seed=777 + FUZZ)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)  # (the pointer object keeps the array alive)


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
    ws = np.full(emu.svo_hip_match_workspace_bytes(P) + 256, 0xFF, np.uint8)   # (poisoned: NaN / -1 to whoever reads scratch it did not write)
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


# ---- row a12: svo_hip_update_seeds (seed_prepare -> warp -> epipolar scan -> alignment -> seed_finish) -----------------


@pytest.fixture(scope="module", params=[0, 1], ids=["default", "lane_kernel_with_finish"])
def emu_seeds(request):
    from emu_build import build_emulated
    from emu_build import BUILDS
    return build_emulated(BUILDS[request.param])


def _make_seeds(scene, orc, rng):  # (tests/test_tracking_gpu.py)
    seeds, feats = [], []
    for i in range(len(scene.obs)):
        o = [x for x in scene.obs[i] if x[0] != scene.cur][0]
        c_ref = -scene.T_f_w[o[0], :9].reshape(3, 3).T @ scene.T_f_w[o[0], 9:]
        d_true = np.linalg.norm(scene.pt_pos[i] - c_ref)
        s = orc.seed_init(d_true * (1 + 0.1 * rng.normal()), d_true * 0.6)
        s.ftr = pytrack.make_feature(*o)
        s.batch_id = int(rng.integers(0, 6))
        if i % 7 == 0:
            s.sigma2 = np.float32(s.sigma2 * 7e-4)
        if i % 31 == 0:
            s.mu = np.float32(-0.3)
        if i % 5 == 0:
            s.sigma2 = np.float32(s.sigma2 * 1e-3)
        seeds.append(s)
        feats.append(o)
    return seeds, feats


@pytest.mark.parametrize("align_1d,subpix", [(0, 1), (1, 1), (0, 0)])
def test_emulated_update_seeds(emu_seeds, oracle, scene, align_1d, subpix):
    """DepthFilter::updateSeeds as the five kernels of svo_hip_update_seeds run it, on the CPU, with the requirements of the
    GPU test -- and, the geometry being host-compiled without contraction like the oracle's, tighter ones: statuses identical
    (off the convergence threshold), mu / sigma2 / a / b to 1e-5, refined pixels to 1e-9."""
    emu = emu_seeds
    orc = pytrack.Track("orc")
    pyrs = [orc.create_img_pyramid(im, 5) for im in scene.images.cpu().numpy()]
    layout, store = _store(emu, scene)
    T = np.ascontiguousarray(scene.T_f_w)
    n_frames = T.shape[0]
    slots = np.arange(n_frames, dtype=np.int32)
    frames = capi.Frames(n_frames, 0, slots.ctypes.data, T.ctypes.data)
    rng = fuzz_rng(8)
    seeds, feats = _make_seeds(scene, orc, rng)
    S = len(seeds)
    opt = pytrack.matcher_options(n_pyr_levels=5, align_1d=align_1d, subpix_refinement=subpix)
    oframes = pytrack.make_frames(pyrs, scene.T_f_w)
    nu, so, io = orc.update_seeds(oframes, scene.cam, scene.cur, seeds, batch_counter=5, opt=opt)
    c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
    f_frame, f_level = c([o[0] for o in feats], np.int32), c([o[3] for o in feats], np.int32)
    f_px, f_f = c([o[1] for o in feats], np.float64), c([o[2] for o in feats], np.float64)
    f_type, f_grad = c([o[4] for o in feats], np.uint8), c([o[5] for o in feats], np.float64)
    ftr = capi.Features(f_frame.ctypes.data, f_level.ctypes.data, f_type.ctypes.data, f_px.ctypes.data, f_f.ctypes.data, f_grad.ctypes.data)
    a, b = c([s.a for s in seeds], np.float32), c([s.b for s in seeds], np.float32)
    mu, zr = c([s.mu for s in seeds], np.float32), c([s.z_range for s in seeds], np.float32)
    s2, bid = c([s.sigma2 for s in seeds], np.float32), c([s.batch_id for s in seeds], np.int32)
    sd = capi.Seeds(a.ctypes.data, b.ctypes.data, mu.ctypes.data, zr.ctypes.data, s2.ctypes.data, bid.ctypes.data)
    dopt = capi.DepthFilterOptions(3, 5, 200.0, int(align_1d), 10, 1000, int(subpix), 1, 5, 0.7)
    cur = np.full(S, scene.cur, np.int32)
    status, xyz, px = np.zeros(S, np.int32), np.zeros((S, 3)), np.zeros((S, 2))
    emu.svo_hip_match_workspace_bytes.restype = C.c_size_t
    ws = np.full(emu.svo_hip_match_workspace_bytes(S) + 256, 0xFF, np.uint8)   # (poisoned: NaN / -1 to whoever reads scratch it did not write)
    cam = capi.camera(scene.cam)
    state0 = [x.copy() for x in (a, b, mu, zr, s2, bid)]
    rc = emu.svo_hip_update_seeds(C.byref(layout), _p(store), C.byref(cam), C.byref(frames), S, _p(cur), C.byref(ftr), C.byref(sd),
                                  C.byref(dopt), _p(status), _p(xyz), _p(px), _p(ws), C.c_size_t(ws.size), None)
    assert rc == 0, rc
    # The same seeds keyframe by keyframe, as a seed list holds them: a wave of seed_prepare_kernel then finds a handful of
    # RUNS of equal (reference, current) pairs whose first seed files the pair's poses for seed_finish (in the list order
    # above the keyframe changes from seed to seed: nearly every seed a run of its own).  Same bits either way.
    order = np.argsort(f_frame, kind="stable")
    assert len(np.unique(f_frame)) >= 2 and np.count_nonzero(np.diff(f_frame) != 0) > 64 > np.count_nonzero(np.diff(f_frame[order]) != 0)
    for ws_bytes in (ws.size,):
        pf = [c(x[order], x.dtype) for x in (f_frame, f_level, f_type, f_px, f_f, f_grad)]
        ps = [c(x[order], x.dtype) for x in state0]
        pftr = capi.Features(pf[0].ctypes.data, pf[1].ctypes.data, pf[2].ctypes.data, pf[3].ctypes.data, pf[4].ctypes.data, pf[5].ctypes.data)
        psd = capi.Seeds(*[x.ctypes.data for x in ps])
        st2, xyz2, px2 = np.zeros(S, np.int32), np.zeros((S, 3)), np.zeros((S, 2))
        ws2 = np.full(ws.size, 0xFF, np.uint8)
        rc = emu.svo_hip_update_seeds(C.byref(layout), _p(store), C.byref(cam), C.byref(frames), S, _p(cur), C.byref(pftr), C.byref(psd),
                                      C.byref(dopt), _p(st2), _p(xyz2), _p(px2), _p(ws2), C.c_size_t(ws_bytes), None)
        assert rc == 0, rc
        assert np.array_equal(st2, status[order]) and np.array_equal(px2, px[order])
        conv = status[order] == pytrack.SEED_CONVERGED
        assert np.array_equal(xyz2[conv], xyz[order][conv])
        for got, want in zip(ps, (a, b, mu, zr, s2, bid)):
            assert np.array_equal(got, want[order])
    hist = {}
    for i in range(S):
        st = io[i].status
        hist[st] = hist.get(st, 0) + 1
        if st in (pytrack.SEED_UPDATED, pytrack.SEED_CONVERGED):
            margin = abs(np.sqrt(max(so[i].sigma2, 0.0)) * 200.0 / so[i].z_range - 1.0)
            if margin > 1e-3:
                assert status[i] == st, (i, status[i], st)
            else:
                assert status[i] in (pytrack.SEED_UPDATED, pytrack.SEED_CONVERGED)
        else:
            assert status[i] == st, (i, status[i], st)
        if st in (pytrack.SEED_UPDATED, pytrack.SEED_CONVERGED, pytrack.SEED_NO_MATCH):
            assert np.isclose(mu[i], so[i].mu, rtol=2e-6, atol=0), (i, mu[i], so[i].mu)
            assert abs(float(s2[i]) - so[i].sigma2) <= 1e-4 * abs(so[i].sigma2) + 1e-6 * so[i].mu ** 2, (i, s2[i], so[i].sigma2)
            assert np.allclose([a[i], b[i]], [so[i].a, so[i].b], rtol=1e-4, atol=1e-5), (i, a[i], so[i].a, b[i], so[i].b)
        if st in (pytrack.SEED_UPDATED, pytrack.SEED_CONVERGED):
            assert np.abs(px[i] - np.array(io[i].px_cur[:])).max() < 1e-9
        if st == pytrack.SEED_CONVERGED:
            assert np.abs(xyz[i] - np.array(io[i].xyz_world[:])).max() < 1e-5
    assert hist.get(pytrack.SEED_UPDATED, 0) > 50 and hist.get(pytrack.SEED_CONVERGED, 0) > 2
    assert hist.get(pytrack.SEED_ERASED_OLD, 0) > 5 and hist.get(pytrack.SEED_BEHIND, 0) > 2


def test_emulated_update_seeds_on_the_resident_store(emu_seeds, scene):
    """Row N2, seeds, on the CPU: the seeds scattered over a resident store (svo_hip_seed_store_patch) and updated in list
    order through svo_hip_update_seeds_resident give the bits svo_hip_update_seeds gives on the flattened list -- statuses,
    points, px_cur, the state in the store and in the dense read-back -- and leave every other slot alone."""
    emu = emu_seeds
    orc = pytrack.Track("orc")
    layout, store = _store(emu, scene)
    T = np.ascontiguousarray(scene.T_f_w)
    n_frames = T.shape[0]
    slots_f = np.arange(n_frames, dtype=np.int32)
    frames = capi.Frames(n_frames, 0, slots_f.ctypes.data, T.ctypes.data)
    rng = fuzz_rng(8)
    seeds, feats = _make_seeds(scene, orc, rng)
    S = len(seeds)
    c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)

    def columns(idx):
        f = [c([feats[i][0] for i in idx], np.int32), c([feats[i][3] for i in idx], np.int32), c([feats[i][4] for i in idx], np.uint8),
             c([feats[i][1] for i in idx], np.float64), c([feats[i][2] for i in idx], np.float64), c([feats[i][5] for i in idx], np.float64)]
        sd = [c([seeds[i].a for i in idx], np.float32), c([seeds[i].b for i in idx], np.float32), c([seeds[i].mu for i in idx], np.float32),
              c([seeds[i].z_range for i in idx], np.float32), c([seeds[i].sigma2 for i in idx], np.float32), c([seeds[i].batch_id for i in idx], np.int32)]
        return f, sd
    structs = lambda f, sd: (capi.Features(*[x.ctypes.data for x in f]), capi.Seeds(*[x.ctypes.data for x in sd]))
    dopt = capi.DepthFilterOptions(3, 5, 200.0, 0, 10, 1000, 1, 1, 5, 0.7)
    cam = capi.camera(scene.cam)
    emu.svo_hip_match_workspace_bytes.restype = C.c_size_t
    ws = np.full(emu.svo_hip_match_workspace_bytes(S) + 256, 0xFF, np.uint8)
    # flattened list
    f0, s0 = columns(range(S))
    ftr0, sd0 = structs(f0, s0)
    cur = np.full(S, scene.cur, np.int32)
    st0, xyz0, px0 = np.zeros(S, np.int32), np.zeros((S, 3)), np.zeros((S, 2))
    assert emu.svo_hip_update_seeds(C.byref(layout), _p(store), C.byref(cam), C.byref(frames), S, _p(cur), C.byref(ftr0), C.byref(sd0),
                                    C.byref(dopt), _p(st0), _p(xyz0), _p(px0), _p(ws), C.c_size_t(ws.size), None) == 0
    # resident store of 3 S slots, patched in two instalments
    cap = 3 * S
    slot_of = rng.permutation(cap)[:S].astype(np.int32)
    fS = [np.full(cap, 77, np.int32), np.full(cap, 77, np.int32), np.full(cap, 77, np.uint8), np.full((cap, 2), 77.0), np.full((cap, 3), 77.0), np.full((cap, 2), 77.0)]
    sS = [np.full(cap, 77, np.float32) for _ in range(5)] + [np.full(cap, 77, np.int32)]
    ftrS, sdS = structs(fS, sS)
    for part in (list(range(S // 2)), list(range(S // 2, S))):
        fp, sp = columns(part)
        sl = np.ascontiguousarray(slot_of[part])
        ftp, sdp = structs(fp, sp)
        patch = capi.SeedPatch(len(part), 0, sl.ctypes.data, ftp, sdp)
        assert emu.svo_hip_seed_store_patch(C.byref(patch), C.byref(ftrS), C.byref(sdS), None) == 0
    ws[:] = 0xFF
    st1, xyz1, px1, state = np.zeros(S, np.int32), np.zeros((S, 3)), np.zeros((S, 2)), np.zeros((4, S), np.float32)
    assert emu.svo_hip_update_seeds_resident(C.byref(layout), _p(store), C.byref(cam), C.byref(frames), int(scene.cur), S, _p(slot_of), C.byref(ftrS),
                                             C.byref(sdS), C.byref(dopt), _p(st1), _p(xyz1), _p(px1), _p(state), _p(ws), C.c_size_t(ws.size), None) == 0
    assert np.array_equal(st0, st1) and np.array_equal(px0, px1)
    conv = st0 == pytrack.SEED_CONVERGED
    assert np.array_equal(xyz0[conv], xyz1[conv]) and conv.sum() > 2
    touched = np.isin(st0, (pytrack.SEED_NO_MATCH, pytrack.SEED_UPDATED, pytrack.SEED_CONVERGED, pytrack.SEED_NAN))
    for k, name in enumerate(("a", "b", "mu", None, "sigma2")):
        if name is None:
            continue
        flat, resident = s0[k], sS[k][slot_of]
        assert np.array_equal(flat.view(np.int32), resident.view(np.int32)), name
        row = {"a": 0, "b": 1, "mu": 2, "sigma2": 3}[name]
        assert np.array_equal(state[row][touched].view(np.int32), flat[touched].view(np.int32)), name
    free = np.ones(cap, bool)
    free[slot_of] = False
    assert (sS[2][free] == 77).all() and (fS[1][free] == 77).all() and (sS[5][free] == 77).all() and touched.sum() > 100
    # svo_hip_update_seeds_resident_pose: the current frame's pose by value, its row of the table poisoned (a host that uploaded
    # the tables before the pose was known) -- on a second store patched with the same records: the same bits
    fS2 = [np.full(cap, 77, np.int32), np.full(cap, 77, np.int32), np.full(cap, 77, np.uint8), np.full((cap, 2), 77.0), np.full((cap, 3), 77.0), np.full((cap, 2), 77.0)]
    sS2 = [np.full(cap, 77, np.float32) for _ in range(5)] + [np.full(cap, 77, np.int32)]
    ftrS2, sdS2 = structs(fS2, sS2)
    fp, sp = columns(list(range(S)))
    ftp, sdp = structs(fp, sp)
    sl = np.ascontiguousarray(slot_of)
    patch = capi.SeedPatch(S, 0, sl.ctypes.data, ftp, sdp)
    assert emu.svo_hip_seed_store_patch(C.byref(patch), C.byref(ftrS2), C.byref(sdS2), None) == 0
    T_poisoned = T.copy()
    T_cur = np.ascontiguousarray(T[scene.cur].copy())
    T_poisoned[scene.cur] = np.nan
    frames_p = capi.Frames(n_frames, 0, slots_f.ctypes.data, T_poisoned.ctypes.data)
    ws[:] = 0xFF
    st2, xyz2, px2, state2 = np.zeros(S, np.int32), np.zeros((S, 3)), np.zeros((S, 2)), np.zeros((4, S), np.float32)
    assert emu.svo_hip_update_seeds_resident_pose(C.byref(layout), _p(store), C.byref(cam), C.byref(frames_p), int(scene.cur), _p(T_cur), S,
                                                  _p(slot_of), C.byref(ftrS2), C.byref(sdS2), C.byref(dopt), _p(st2), _p(xyz2), _p(px2),
                                                  _p(state2), _p(ws), C.c_size_t(ws.size), None) == 0
    assert np.array_equal(st2, st1) and np.array_equal(px2, px1) and np.array_equal(xyz2[conv], xyz1[conv])
    assert np.array_equal(state2[:, touched].view(np.int32), state[:, touched].view(np.int32))
    for k in (0, 1, 2, 4):
        assert np.array_equal(sS2[k].view(np.int32), sS[k].view(np.int32))
    assert emu.svo_hip_update_seeds_resident_pose(C.byref(layout), _p(store), C.byref(cam), C.byref(frames_p), int(scene.cur), None, S,
                                                  _p(slot_of), C.byref(ftrS2), C.byref(sdS2), C.byref(dopt), _p(st2), _p(xyz2), _p(px2),
                                                  _p(state2), _p(ws), C.c_size_t(ws.size), None) == -1  # (EINVAL: no pose)
