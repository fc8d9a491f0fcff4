"""Pins the C restatement (oracle/libsvo_oracle.so) against the reference's OWN translation
units compiled in place (oracle/_ref/libsvo_ref.so, see oracle/Makefile + oracle/shim/).

Both libraries are built with the same flags and share the shimmed third-party arithmetic
(Eigen/Sophus/vikit/boost restatements), so every reference-owned computation must agree
BIT FOR BIT: that is what these tests assert (np.array_equal on doubles).

The _ref library is compiled from /root/reference in the build container and travels with
the repo snapshot (it is git-ignored, not gpurun-ignored); where neither the library nor the
reference checkout exists the module is skipped.
"""
import numpy as np
import pytest

from oracle import pyoracle, pytrack
from rpg_svo_amd import se3, synth

pytrack.build_ref()
pytestmark = pytest.mark.skipif(not pytrack.ref_available(), reason="oracle/_ref/libsvo_ref.so not built (no reference checkout)")


@pytest.fixture(scope="module")
def libs():
    pyoracle.build()
    return pytrack.Track("orc"), pytrack.Track("ref")


from helpers import CAMERA_KINDS, camera_models


@pytest.fixture(scope="module", params=CAMERA_KINDS)
def cam(request):
    """every reference-owned computation is pinned under the three vikit camera models"""
    return camera_models()[request.param]


@pytest.fixture(scope="module")
def seq(cam):
    return synth.make_sequence(6, 80, cam=cam, margin=40)


@pytest.fixture(scope="module")
def scene(cam):
    return synth.make_track_scene(n_kf=3, n_feat=60, cam=cam)


@pytest.fixture(scope="module")
def scene_frames(scene, libs):
    orc, _ = libs
    imgs = scene.images.cpu().numpy()
    pyrs = [orc.create_img_pyramid(im, 5) for im in imgs]
    return pyrs


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("shape", [(480, 640), (120, 188), (61, 75)])
def test_pyramid(libs, mode, shape):
    orc, ref = libs
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    a = orc.create_img_pyramid(img, 4, mode)
    b = ref.create_img_pyramid(img, 4, mode)
    for x, y in zip(a, b):
        assert same(x, y)


@pytest.mark.parametrize("levels", [(3, 0), (4, 2), (2, 2)])
def test_sparse_img_align(libs, seq, levels):
    orc, ref = libs
    imgs = seq.images.cpu().numpy()
    pyrs = [orc.create_img_pyramid(im, 5) for im in imgs]
    px, f, pos = seq.px.numpy(), seq.f.numpy(), seq.pos.numpy()
    rng = np.random.default_rng(0)
    for r, c in [(0, 1), (1, 3), (2, 5)]:
        hp = (rng.uniform(size=px.shape[1]) > 0.1).astype(np.uint8)
        p = px[r].copy()
        p[:5] = [[1.0, 1.0], [638.5, 470.0], [30.0, 477.9], [24.0, 24.0], [23.9, 300.0]]  # border cases
        To, ro = orc.sparse_img_align_run(pyrs[r], pyrs[c], seq.cam, seq.T_f_w[r], seq.T_f_w[r], p, f[r], hp, pos[r], *levels)
        Tr, rr = ref.sparse_img_align_run(pyrs[r], pyrs[c], seq.cam, seq.T_f_w[r], seq.T_f_w[r], p, f[r], hp, pos[r], *levels)
        assert same(To, Tr)
        for k in ("n_tracked", "stop", "iters", "chi2", "H", "visible"):
            assert same(ro[k], rr[k]), k
        # the driver can only re-derive T_cur_from_ref from the two frame poses afterwards
        assert np.abs(ro["T_cur_from_ref"] - rr["T_cur_from_ref"]).max() < 1e-12
        assert ro["n_tracked"] > 30


def test_sparse_img_align_degenerate(libs, seq):
    orc, ref = libs
    imgs = seq.images.cpu().numpy()
    pyrs = [orc.create_img_pyramid(im, 4) for im in imgs[:2]]
    px, f, pos = seq.px.numpy()[0], seq.f.numpy()[0], seq.pos.numpy()[0]
    cases = [
        (0, np.ones(0, np.uint8)),                       # no features: return 0, pose untouched
        (10, np.zeros(10, np.uint8)),                    # features without points: H = 0 -> NaN solve -> stop
        (1, np.ones(1, np.uint8)),                       # rank-deficient system
        (2, np.ones(2, np.uint8)),
    ]
    for n, hp in cases:
        To, ro = orc.sparse_img_align_run(pyrs[0], pyrs[1], seq.cam, seq.T_f_w[0], seq.T_f_w[0], px[:n], f[:n], hp, pos[:n], 3, 0)
        Tr, rr = ref.sparse_img_align_run(pyrs[0], pyrs[1], seq.cam, seq.T_f_w[0], seq.T_f_w[0], px[:n], f[:n], hp, pos[:n], 3, 0)
        assert same(To, Tr), n
        for k in ("n_tracked", "stop", "iters"):
            assert same(ro[k], rr[k]), (n, k)


def _patches(img, u, v):
    iu, iv = int(u), int(v)
    pwb = img[iv - 5:iv + 5, iu - 5:iu + 5].copy()
    return pwb.ravel(), pwb[1:9, 1:9].copy().ravel()


def test_align2d_align1d(libs, seq):
    orc, ref = libs
    img = seq.images.cpu().numpy()[0]
    img2 = seq.images.cpu().numpy()[1]
    rng = np.random.default_rng(3)
    n_conv = 0
    for i in range(200):
        u, v = rng.uniform(20, 620), rng.uniform(20, 460)
        pwb, patch = _patches(img, u, v)
        target = img if i % 2 == 0 else img2
        start = np.array([int(u) + rng.uniform(-2.5, 2.5), int(v) + rng.uniform(-2.5, 2.5)])
        if i % 17 == 0:
            start = np.array([3.0 + rng.uniform(0, 2), v])      # leaves the image
        if i % 23 == 0:
            pwb = np.full(100, 77, np.uint8); patch = np.full(64, 77, np.uint8)  # singular H -> NaN
        n_iter = [10, 3, 1][i % 3]
        ok_o, p_o = orc.align2d(target, pwb, patch, n_iter, start)
        ok_r, p_r = ref.align2d(target, pwb, patch, n_iter, start)
        assert ok_o == ok_r and same(p_o, p_r), i
        n_conv += ok_o
        d = rng.normal(size=2); d /= np.linalg.norm(d)
        ok_o, p_o, h_o = orc.align1d(target, d, pwb, patch, n_iter, start)
        ok_r, p_r, h_r = ref.align1d(target, d, pwb, patch, n_iter, start)
        assert ok_o == ok_r and same(p_o, p_r) and same(h_o, h_r), i
    assert n_conv > 50


def test_warp(libs, scene, scene_frames):
    orc, ref = libs
    rng = np.random.default_rng(5)
    T_cr = se3.mul(scene.T_f_w[scene.cur], se3.inv(scene.T_f_w[0]))
    for i in range(100):
        px = np.array([rng.uniform(30, 610), rng.uniform(30, 450)])
        f = synth._bearing(scene.cam, px[None])[0]
        depth = rng.uniform(0.5, 6.0)
        lvl = int(rng.integers(0, 3))
        A_o = orc.get_warp_matrix_affine(scene.cam, px, f, depth, T_cr, lvl)
        A_r = ref.get_warp_matrix_affine(scene.cam, px, f, depth, T_cr, lvl)
        assert same(A_o, A_r)
        A = A_o * rng.uniform(0.3, 4.0) if i % 3 else A_o
        assert orc.get_best_search_level(A, 4) == ref.get_best_search_level(A, 4)
        sl = orc.get_best_search_level(A, 4)
        img = scene_frames[0][lvl]
        if i % 10 == 0:
            px = np.array([2.0, 3.0])       # partially outside -> zeros
        ok_o, p_o = orc.warp_affine(A, img, px, lvl, sl, 5)
        ok_r, p_r = ref.warp_affine(A, img, px, lvl, sl, 5)
        assert ok_o == ok_r and same(p_o, p_r)
    # NaN warp ("camera has no translation", matcher.cpp:83-87): patch left untouched.
    # (an exactly singular A gives +-inf and makes the reference itself read out of bounds)
    A = np.array([[np.nan, 0.0], [0.0, 1.0]])
    ok_o, p_o = orc.warp_affine(A, scene_frames[0][0], np.array([100.0, 100.0]), 0, 0, 5)
    ok_r, p_r = ref.warp_affine(A, scene_frames[0][0], np.array([100.0, 100.0]), 0, 0, 5)
    assert ok_o == ok_r and same(p_o, p_r)


def _features(obs):
    return [pytrack.make_feature(*o) for o in obs]


def test_find_match_direct(libs, scene, scene_frames):
    orc, ref = libs
    T = scene.T_f_w.copy()
    T[scene.cur] = scene.T_cur_prior
    frames = pytrack.make_frames(scene_frames, T)
    opt = pytrack.matcher_options(n_pyr_levels=5)
    n_ok, n_edge = 0, 0
    for i in range(0, len(scene.obs), 2):
        obs = _features(scene.obs[i])
        ok_o, px_o, r_o = orc.find_match_direct(frames, scene.cam, scene.cur, scene.pt_pos[i], obs, scene.px_init[i], opt)
        ok_r, px_r, r_r = ref.find_match_direct(frames, scene.cam, scene.cur, scene.pt_pos[i], obs, scene.px_init[i], opt)
        assert ok_o == ok_r and same(px_o, px_r), i
        assert r_o["ref_obs"] == r_r["ref_obs"]
        if r_o["A_cur_ref"].any():
            for k in ("search_level", "A_cur_ref", "patch", "patch_with_border"):
                assert same(r_o[k], r_r[k]), (i, k)
        n_ok += ok_o
        n_edge += ok_o and scene.obs[i][r_o["ref_obs"]][4] == 1
        if ok_o:
            # sanity of the scene, not parity: corners land on the true projection; edgelets are
            # only constrained along their gradient
            is_edge = scene.obs[i][r_o["ref_obs"]][4] == 1
            assert np.linalg.norm(px_o - scene.px_true[i]) < (3.0 if is_edge else 1.0)
    assert n_ok > 50 and n_edge > 3


def test_find_epipolar_match_direct(libs, scene, scene_frames):
    orc, ref = libs
    frames = pytrack.make_frames(scene_frames, scene.T_f_w)
    rng = np.random.default_rng(11)
    n_ok = 0
    n_short = 0
    for i in range(0, len(scene.obs), 3):
        o = [x for x in scene.obs[i] if x[0] != scene.cur][0]
        ftr = pytrack.make_feature(*o)
        c_ref = -scene.T_f_w[o[0], :9].reshape(3, 3).T @ scene.T_f_w[o[0], 9:]
        d_true = np.linalg.norm(scene.pt_pos[i] - c_ref)
        spread = [0.4, 0.1, 0.0005][(i // 3) % 3]
        d_est = d_true * (1 + rng.normal() * spread * 0.3)
        d_min, d_max = d_est * (1 - spread), d_est * (1 + spread)
        for align_1d, subpix in ((0, 1), (1, 1), (0, 0)):   # subpix 0: triangulate from uv_best (matcher.cpp:316-318)
            opt = pytrack.matcher_options(n_pyr_levels=5, align_1d=align_1d, subpix_refinement=subpix)
            ok_o, r_o = orc.find_epipolar_match_direct(frames, scene.cam, o[0], scene.cur, ftr, d_est, d_min, d_max, opt)
            ok_r, r_r = ref.find_epipolar_match_direct(frames, scene.cam, o[0], scene.cur, ftr, d_est, d_min, d_max, opt)
            assert ok_o == ok_r, i
            for k in ("search_level", "reject", "A_cur_ref", "epi_length"):
                assert same(r_o[k], r_r[k]), (i, k)
            if not r_o["reject"]:
                assert same(r_o["patch"], r_r["patch"])
            if ok_o:
                assert same(r_o["depth"], r_r["depth"]) and same(r_o["px_cur"], r_r["px_cur"])
                n_ok += 1
                n_short += r_o["epi_length"] < 2.0
    assert n_ok > 60 and n_short > 7


def test_find_epipolar_match_direct_search_step_cap(libs, scene, scene_frames):
    """Matcher::Options::max_epi_search_steps below the scan length (matcher.cpp:248-256, "skip epipolar search"): the C port
    leaves the search where the reference does -- same verdict, and what the reference's Matcher holds at its early return
    (search level, warp matrix, segment length) is what the port reports."""
    orc, ref = libs
    frames = pytrack.make_frames(scene_frames, scene.T_f_w)
    rng = np.random.default_rng(23)
    n_cut = n_ok = 0
    for i in range(0, len(scene.obs), 3):
        o = [x for x in scene.obs[i] if x[0] != scene.cur][0]
        ftr = pytrack.make_feature(*o)
        c_ref = -scene.T_f_w[o[0], :9].reshape(3, 3).T @ scene.T_f_w[o[0], 9:]
        d_true = np.linalg.norm(scene.pt_pos[i] - c_ref)
        spread = [0.5, 0.25, 0.05][(i // 3) % 3]
        d_est = d_true * (1 + rng.normal() * spread * 0.2)
        d_min, d_max = d_est * (1 - spread), d_est * (1 + spread)
        free = pytrack.matcher_options(n_pyr_levels=5)
        ok_free, _ = ref.find_epipolar_match_direct(frames, scene.cam, o[0], scene.cur, ftr, d_est, d_min, d_max, free)
        for cap in (6, 20):
            opt = pytrack.matcher_options(n_pyr_levels=5, max_epi_search_steps=cap)
            ok_o, r_o = orc.find_epipolar_match_direct(frames, scene.cam, o[0], scene.cur, ftr, d_est, d_min, d_max, opt)
            ok_r, r_r = ref.find_epipolar_match_direct(frames, scene.cam, o[0], scene.cur, ftr, d_est, d_min, d_max, opt)
            assert ok_o == ok_r, (i, cap)
            for k in ("search_level", "reject", "A_cur_ref", "epi_length"):
                assert same(r_o[k], r_r[k]), (i, cap, k)
            if ok_o:
                assert same(r_o["depth"], r_r["depth"]) and same(r_o["px_cur"], r_r["px_cur"])
                n_ok += 1
            n_cut += int(ok_free and not ok_r and r_r["epi_length"] / 0.7 > cap)
    assert n_cut > 20 and n_ok > 20, (n_cut, n_ok)


def test_pose_optimize(libs, scene):
    orc, ref = libs
    rng = np.random.default_rng(2)
    P = len(scene.pt_pos)
    f = synth._bearing(scene.cam, scene.px_true + rng.normal(size=(P, 2)) * 0.3)
    level = rng.integers(0, 3, size=P).astype(np.int32)
    pos = scene.pt_pos.copy()
    pos[::15] += rng.normal(size=pos[::15].shape) * 0.2           # outliers
    for trial in range(4):
        hp = (rng.uniform(size=P) > 0.2).astype(np.uint8)
        n = [P, 40, 7, 3][trial]
        T0 = se3.mul(se3.exp(rng.normal(size=6) * 5e-3), scene.T_f_w[scene.cur])
        a = orc.pose_optimize(scene.cam, T0, f[:n], level[:n], hp[:n], pos[:n], n_iter=[10, 10, 3, 10][trial])
        b = ref.pose_optimize(scene.cam, T0, f[:n], level[:n], hp[:n], pos[:n], n_iter=[10, 10, 3, 10][trial])
        for k in ("T_f_w", "Cov", "estimated_scale", "error_init", "error_final", "num_obs", "ran", "has_point"):
            assert same(a[k], b[k]), (trial, k)
        if trial == 0:
            assert se3.log_norm(a["T_f_w"][None], scene.T_f_w[scene.cur][None])[0] < 6e-3   # scene sanity (noisy observations), not parity
            assert a["num_obs"] < hp.sum()
    # no observation with a point: nothing happens
    a = orc.pose_optimize(scene.cam, T0, f[:5], level[:5], np.zeros(5, np.uint8), pos[:5])
    b = ref.pose_optimize(scene.cam, T0, f[:5], level[:5], np.zeros(5, np.uint8), pos[:5])
    assert a["ran"] == b["ran"] == 0


def test_point_optimize(libs, scene):
    orc, ref = libs
    rng = np.random.default_rng(4)
    for i in range(0, len(scene.obs), 5):
        o = scene.obs[i]
        T = np.array([scene.T_f_w[x[0]] for x in o])
        f = np.array([x[2] for x in o]) + rng.normal(size=(len(o), 3)) * 1e-3
        p0 = scene.pt_pos[i] + rng.normal(size=3) * 0.05
        a = orc.point_optimize(T, f, p0, 5)
        b = ref.point_optimize(T, f, p0, 5)
        assert same(a, b), i


def test_update_seed_and_tau(libs):
    orc, ref = libs
    rng = np.random.default_rng(6)
    s_o = orc.seed_init(2.1, 0.7)
    s_r = ref.seed_init(2.1, 0.7)
    assert bytes(s_o)[-20:] == bytes(s_r)[-20:]
    for i in range(500):
        x = np.float32(1.0 / rng.uniform(0.5, 5))
        tau2 = np.float32(10.0 ** rng.uniform(-8, 0))
        if i % 50 == 0:
            tau2 = np.float32(0.0)
        if i % 77 == 0:
            x = np.float32(50.0)   # far outlier
        n_o = orc.update_seed(x, tau2, s_o)
        n_r = ref.update_seed(x, tau2, s_r)
        assert bytes(n_o)[-20:] == bytes(n_r)[-20:], i
        if not np.isnan(n_o.mu) and not np.isnan(n_o.a) and not np.isnan(n_o.sigma2):
            s_o, s_r = n_o, n_r
        T = se3.exp(rng.normal(size=6) * 0.3)
        f = rng.normal(size=3); f /= np.linalg.norm(f)
        z = rng.uniform(0.3, 8)
        assert same(orc.compute_tau(T, f, z, 0.0025), ref.compute_tau(T, f, z, 0.0025))


def test_update_seeds(libs, scene, scene_frames):
    orc, ref = libs
    frames = pytrack.make_frames(scene_frames, scene.T_f_w)
    rng = np.random.default_rng(8)
    seeds = []
    for i in range(0, len(scene.obs), 2):
        o = [x for x in scene.obs[i] if x[0] != scene.cur][0]
        c_ref = -scene.T_f_w[o[0], :9].reshape(3, 3).T @ scene.T_f_w[o[0], 9:]
        d_true = np.linalg.norm(scene.pt_pos[i] - c_ref)
        s = orc.seed_init(d_true * (1 + 0.1 * rng.normal()), d_true * 0.6)
        s.ftr = pytrack.make_feature(*o)
        s.batch_id = int(rng.integers(0, 6))
        if i % 7 == 0:
            s.sigma2 = np.float32(s.sigma2 * 1e-4)   # nearly converged
        if i % 31 == 0:
            s.mu = np.float32(-0.3)                   # behind the camera
        seeds.append(s)
    opt = pytrack.matcher_options(n_pyr_levels=5)
    nu_o, s_o, i_o = orc.update_seeds(frames, scene.cam, scene.cur, seeds, batch_counter=5, opt=opt)
    nu_r, s_r, i_r = ref.update_seeds(frames, scene.cam, scene.cur, seeds, batch_counter=5, opt=opt)
    assert nu_o == nu_r
    hist = {}
    for k, (a, b, ia, ib) in enumerate(zip(s_o, s_r, i_o, i_r)):
        st = ia.status
        hist[st] = hist.get(st, 0) + 1
        st_cmp = 0 if st in (pytrack.SEED_BEHIND, pytrack.SEED_NOT_IN_FRAME) else st
        assert st_cmp == ib.status, (k, st, ib.status)
        if st not in (pytrack.SEED_ERASED_OLD, pytrack.SEED_CONVERGED, pytrack.SEED_NAN):
            assert bytes(a)[-20:] == bytes(b)[-20:], k
        if st == pytrack.SEED_CONVERGED:
            assert same(np.array(ia.xyz_world[:]), np.array(ib.xyz_world[:]))
            # the state AFTER the converging update: the port reports it, the reference's driver recovers mu and sigma2 from
            # the converged callback (the variance it is handed; the new point's distance from the seed's frame = 1 / mu)
            assert np.float32(a.mu).tobytes() == np.float32(b.mu).tobytes(), (k, a.mu, b.mu)
            assert np.float32(a.sigma2).tobytes() == np.float32(b.sigma2).tobytes(), (k, a.sigma2, b.sigma2)
    assert hist.get(pytrack.SEED_UPDATED, 0) > 20 and hist.get(pytrack.SEED_CONVERGED, 0) > 2
    assert hist.get(pytrack.SEED_ERASED_OLD, 0) > 2


def test_reproject_point(libs, scene):
    orc, ref = libs
    for i in range(0, len(scene.pt_pos), 3):
        k_o, p_o = orc.reproject_point(scene.cam, scene.T_cur_prior, scene.pt_pos[i], 30, 22)
        k_r, p_r = ref.reproject_point(scene.cam, scene.T_cur_prior, scene.pt_pos[i], 30, 22)
        assert k_o == k_r and same(p_o, p_r)


def test_cam2world(libs, cam):
    """Frame::c2f -- what the Feature constructor stores in Feature::f (feature.h:44-52) -- for pixels all over the image"""
    orc, ref = libs
    rng = np.random.default_rng(8)
    px = np.stack([rng.uniform(0, cam.width, 500), rng.uniform(0, cam.height, 500)], axis=1)
    assert same(orc.cam2world(cam, px), ref.cam2world(cam, px))


def _walk_cells(cell, ok, max_fts):
    """Reprojector::reprojectMap's cell loop as the reference writes it (reprojector.cpp:131-139, 150-200): a list per
    cell, erase on failure, return at the first success."""
    cells, order = {}, []
    for m, c in enumerate(cell):
        if c not in cells:
            cells[c] = []
            order.append(c)
        cells[c].append(m)
    chosen, n_matches = [], 0
    for c in order:
        lst = cells[c]
        matched = False
        while lst:
            m = lst.pop(0)
            if not ok[m]:
                continue
            chosen.append(m)
            matched = True
            break
        if matched:
            n_matches += 1
        if n_matches > max_fts:
            break
    return chosen


@pytest.mark.parametrize("max_fts", [0, 5, 120, 10000])
def test_select_matches(libs, cam, max_fts):
    orc, _ = libs
    rng = np.random.default_rng(max_fts + 1)
    for M in (0, 1, 7, 300, 1500):
        runs = rng.integers(1, 9, size=M + 1)
        cell = np.repeat(rng.permutation(M + 1), runs)[:M].astype(np.int32)   # cells in a shuffled order, runs of 1-8 trials
        ok = (rng.uniform(size=M) < 0.4).astype(np.int32)
        px = np.stack([rng.uniform(0, cam.width, M), rng.uniform(0, cam.height, M)], axis=1).reshape(M, 2)
        level = rng.integers(0, 4, size=M).astype(np.int32)
        pos = rng.normal(size=(M, 3))
        sel, f, lvl, p = pytrack.select_matches(cam, cell, ok, px, level, pos, max_fts)
        want = _walk_cells(cell.tolist(), ok.tolist(), max_fts)
        assert sel.tolist() == want and len(want) <= max_fts + 1
        assert same(f, orc.cam2world(cam, px[want]).reshape(-1, 3)) and same(lvl, level[want]) and same(p, pos[want])
