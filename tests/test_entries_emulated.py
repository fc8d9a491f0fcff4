"""The remaining C-ABI entry points of the tracking kernels without a GPU (host-emulated library, tests/emu_build.py), each with
the requirements of its GPU test in tests/test_tracking_gpu.py: svo_hip_align_batch / _counted / _phased (the phased form --
three launches, survivors compacted through atomically filled queues, parked loop state -- starts at 2048 trials in the
emulated build), svo_hip_find_epipolar_match_direct, svo_hip_update_seed_batch, svo_hip_compute_tau_batch,
svo_hip_select_matches, svo_hip_compose_poses, svo_hip_cam2world."""
import ctypes as C

import numpy as np
import pytest

from helpers import camera_models
from oracle import pytrack
from rpg_svo_amd import capi, se3, synth


@pytest.fixture(scope="module", params=[0], ids=["default"])
def emu(request):
    from emu_build import BUILDS, build_emulated
    return build_emulated(BUILDS[request.param])


@pytest.fixture(scope="module")
def emu_default():
    """(the default build only: tests of kernels no queued flag touches)"""
    from emu_build import build_emulated
    return build_emulated(())


@pytest.fixture(scope="module")
def scene():
    return synth.make_track_scene(n_kf=4, n_feat=100, cam=camera_models()["pinhole"])


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)  # (the pointer object keeps the array alive)


def _store(emu, images, n_levels=5):
    imgs = np.ascontiguousarray(images)
    n, h, w = imgs.shape
    layout = capi.pyr_layout(w, h, n_levels)
    buf = np.zeros(capi.pyr_store_bytes(layout, n), np.uint8)
    assert emu.svo_hip_pyramid_build_tiled(C.byref(layout), _p(buf), 0, n, _p(imgs), C.c_longlong(h * w), w, capi.HALFSAMPLE_AUTO, 0, None) == 0
    return layout, buf


def test_emulated_align_batch_and_its_phased_form(emu, oracle, scene):
    orc = pytrack.Track("orc")
    imgs = scene.images.cpu().numpy()
    pyrs = [orc.create_img_pyramid(im, 5) for im in imgs]
    layout, store = _store(emu, imgs)
    rng = np.random.default_rng(3)
    M = 4096
    slot = rng.integers(0, imgs.shape[0], size=M).astype(np.int32)
    level = rng.integers(0, 3, size=M).astype(np.int32)
    pwb, px0 = np.zeros((M, 100), np.uint8), np.zeros((M, 2))
    dirs = rng.normal(size=(M, 2)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    use_1d = (rng.uniform(size=M) < 0.3).astype(np.uint8)
    for t in range(M):
        img = pyrs[slot[t]][level[t]]
        h, w = img.shape
        u, v = rng.integers(8, w - 8), rng.integers(8, h - 8)
        src = pyrs[(slot[t] + (t % 2)) % imgs.shape[0]][level[t]]
        pwb[t] = src[v - 5:v + 5, u - 5:u + 5].ravel()
        px0[t] = [u + rng.uniform(-4.0, 4.0), v + rng.uniform(-4.0, 4.0)]
        if t % 37 == 0:
            px0[t] = [3.0 + rng.uniform(0, 2), v]                       # leaves the image
        if t % 41 == 0:
            pwb[t] = 77                                                 # singular H -> NaN
    n_iter = 10

    def run(entry, extra):
        px, ok, h_inv = px0.copy(), np.zeros(M, np.int32), np.zeros(M)
        rc = getattr(emu, entry)(C.byref(layout), _p(store), M, _p(slot), _p(level), _p(pwb), _p(dirs), _p(use_1d), n_iter, _p(px), _p(ok),
                                 _p(h_inv), *extra, None)
        assert rc == 0, (entry, rc)
        return px, ok, h_inv

    px_a, ok_a, h_a = run("svo_hip_align_batch", ())
    ev_c = np.zeros(M, np.int32)
    px_c, ok_c, h_c = run("svo_hip_align_batch_counted", (_p(ev_c),))
    emu.svo_hip_align_workspace_bytes.restype = C.c_size_t
    need = emu.svo_hip_align_workspace_bytes(M)
    assert need > 0                                                     # (the emulated build's threshold: the phased path)
    raw, ev_p = np.full(need + 512, 0xFF, np.uint8), np.zeros(M, np.int32)
    ws = raw[(-raw.ctypes.data) % 256:][:need + 256]                     # (the entry wants 256-byte alignment, like hipMalloc's)
    px_p, ok_p, h_p = run("svo_hip_align_batch_phased", (_p(ev_p), _p(ws), C.c_size_t(ws.size)))
    for px, ok, h in ((px_c, ok_c, h_c), (px_p, ok_p, h_p)):
        assert np.array_equal(ok, ok_a) and np.array_equal(px.view(np.uint64), px_a.view(np.uint64))
        assert np.array_equal(h.view(np.uint64), h_a.view(np.uint64))
    assert np.array_equal(ev_c, ev_p)
    assert (ev_c > 6).sum() > 20 and (ev_c <= 3).sum() > 100, np.bincount(ev_c)   # all three launches had work
    n_conv = 0
    for t in range(0, M, 4):                                             # the single launch against the reference's functions
        img = pyrs[slot[t]][level[t]]
        patch = pwb[t].reshape(10, 10)[1:9, 1:9].ravel()
        if use_1d[t]:
            o, p, hi = orc.align1d(img, dirs[t], pwb[t], patch, n_iter, px0[t])
            assert hi == h_a[t] or (np.isnan(hi) and np.isnan(h_a[t])) or (np.isinf(hi) and np.isinf(h_a[t])), t
        else:
            o, p = orc.align2d(img, pwb[t], patch, n_iter, px0[t])
        assert bool(ok_a[t]) == o, t
        assert np.array_equal(p, px_a[t], equal_nan=True), (t, p, px_a[t])
        n_conv += o
    assert n_conv > M // 16


@pytest.mark.parametrize("n_iter", [0, 1, 3, 10])
def test_emulated_wave_alignment_iteration_caps_and_borders(emu, oracle, scene, n_iter):
    """The wave-per-trial kernel (csrc/align_wave.h: batches of up to 8192 trials) against the reference's align2D / align1D
    at iteration caps 0 / 1 / 3 / 10 (Matcher::Options::align_max_iter), with starts on the rim of the valid region (the
    first bounds test fails / the position leaves after a step), flat templates (singular H: NaN updates) and templates from
    a different place (chi2 rises: align1D's early exit): verdict, refined pixel, h_inv and evaluation count per trial."""
    orc = pytrack.Track("orc")
    imgs = scene.images.cpu().numpy()
    pyrs = [orc.create_img_pyramid(im, 5) for im in imgs]
    layout, store = _store(emu, imgs)
    rng = np.random.default_rng(100 + n_iter)
    M = 192
    slot = rng.integers(0, imgs.shape[0], size=M).astype(np.int32)
    level = rng.integers(0, 4, size=M).astype(np.int32)
    pwb, px0 = np.zeros((M, 100), np.uint8), np.zeros((M, 2))
    dirs = rng.normal(size=(M, 2)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    use_1d = (np.arange(M) % 2).astype(np.uint8)
    for t in range(M):
        img = pyrs[slot[t]][level[t]]
        h, w = img.shape
        u, v = rng.integers(8, w - 8), rng.integers(8, h - 8)
        kind = t % 6
        src = pyrs[(slot[t] + 1) % imgs.shape[0]][level[t]]
        pwb[t] = src[v - 5:v + 5, u - 5:u + 5].ravel()
        px0[t] = [u + rng.uniform(-3.0, 3.0), v + rng.uniform(-3.0, 3.0)]
        if kind == 1:
            px0[t] = [4.0 + rng.uniform(0, 0.9), v]                     # on the rim: inside now, one step from leaving
        elif kind == 2:
            px0[t] = [w - 4.0 + rng.uniform(0, 0.5), v]                 # the first bounds test fails: no evaluation
        elif kind == 3:
            pwb[t] = 128                                                # flat template: singular H
        elif kind == 4:
            uu, vv = rng.integers(8, w - 8), rng.integers(8, h - 8)     # a template from somewhere else
            pwb[t] = img[vv - 5:vv + 5, uu - 5:uu + 5].ravel()
    px, ok, h_inv, ev = px0.copy(), np.zeros(M, np.int32), np.zeros(M), np.full(M, -1, np.int32)
    rc = emu.svo_hip_align_batch_counted(C.byref(layout), _p(store), M, _p(slot), _p(level), _p(pwb), _p(dirs), _p(use_1d), n_iter, _p(px),
                                         _p(ok), _p(h_inv), _p(ev), None)
    assert rc == 0
    assert ev.min() >= 0 and ev.max() <= n_iter and (n_iter == 0 or (ev == 0).sum() >= M // 8)
    for t in range(M):
        img = pyrs[slot[t]][level[t]]
        patch = pwb[t].reshape(10, 10)[1:9, 1:9].ravel()
        if use_1d[t]:
            o, p, hi = orc.align1d(img, dirs[t], pwb[t], patch, n_iter, px0[t])
            assert hi == h_inv[t] or (np.isnan(hi) and np.isnan(h_inv[t])) or (np.isinf(hi) and np.isinf(h_inv[t])), (t, hi, h_inv[t])
        else:
            o, p = orc.align2d(img, pwb[t], patch, n_iter, px0[t])
        assert bool(ok[t]) == o, (t, n_iter)
        assert np.array_equal(p, px[t], equal_nan=True), (t, n_iter, p, px[t])


def test_emulated_find_epipolar_match_direct(emu, oracle, scene):
    orc = pytrack.Track("orc")
    imgs = scene.images.cpu().numpy()
    pyrs = [orc.create_img_pyramid(im, 5) for im in imgs]
    layout, store = _store(emu, imgs)
    T = np.ascontiguousarray(scene.T_f_w)
    slots = np.arange(T.shape[0], dtype=np.int32)
    frames = capi.Frames(T.shape[0], 0, slots.ctypes.data, T.ctypes.data)
    oframes = pytrack.make_frames(pyrs, scene.T_f_w)
    rng = np.random.default_rng(11)
    feats, de, dmin, dmax = [], [], [], []
    for i in range(0, len(scene.obs), 2):
        o = [x for x in scene.obs[i] if x[0] != scene.cur][0]
        c_ref = -scene.T_f_w[o[0], :9].reshape(3, 3).T @ scene.T_f_w[o[0], 9:]
        d_true = np.linalg.norm(scene.pt_pos[i] - c_ref)
        spread = [0.4, 0.1, 0.0005][(i // 2) % 3]
        d_est = d_true * (1 + rng.normal() * spread * 0.3)
        feats.append(o); de.append(d_est); dmin.append(d_est * (1 - spread)); dmax.append(d_est * (1 + spread))
    S = len(feats)
    c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
    f_frame, f_level = c([o[0] for o in feats], np.int32), c([o[3] for o in feats], np.int32)
    f_px, f_f = c([o[1] for o in feats], np.float64), c([o[2] for o in feats], np.float64)
    f_type, f_grad = c([o[4] for o in feats], np.uint8), c([o[5] for o in feats], np.float64)
    ftr = capi.Features(f_frame.ctypes.data, f_level.ctypes.data, f_type.ctypes.data, f_px.ctypes.data, f_f.ctypes.data, f_grad.ctypes.data)
    cur = np.full(S, scene.cur, np.int32)
    de, dmin, dmax = c(de, np.float64), c(dmin, np.float64), c(dmax, np.float64)
    cam = capi.camera(scene.cam)
    emu.svo_hip_match_workspace_bytes.restype = C.c_size_t
    ws = np.full(emu.svo_hip_match_workspace_bytes(S) + 256, 0xFF, np.uint8)   # (poisoned: NaN / -1 to whoever reads scratch it did not write)
    for align_1d in (0, 1):
        o_ = capi.DepthFilterOptions(0, 0, 0.0, int(align_1d), 10, 1000, 1, 1, 5, 0.7)
        ok, depth, px, lvl = np.zeros(S, np.int32), np.zeros(S), np.zeros((S, 2)), np.zeros(S, np.int32)
        rc = emu.svo_hip_find_epipolar_match_direct(C.byref(layout), _p(store), C.byref(cam), C.byref(frames), S, _p(cur), C.byref(ftr), _p(de),
                                                    _p(dmin), _p(dmax), C.byref(o_), _p(ok), _p(depth), _p(px), _p(lvl), _p(ws),
                                                    C.c_size_t(ws.size), None)
        assert rc == 0, rc
        opt = pytrack.matcher_options(n_pyr_levels=5, align_1d=align_1d)
        n_ok = 0
        for k in range(S):
            ok_o, r = orc.find_epipolar_match_direct(oframes, scene.cam, feats[k][0], scene.cur, pytrack.make_feature(*feats[k]),
                                                     de[k], dmin[k], dmax[k], opt)
            assert bool(ok[k]) == ok_o, (k, ok[k], ok_o)
            if ok_o:
                n_ok += 1
                assert lvl[k] == r["search_level"]
                assert np.abs(px[k] - r["px_cur"]).max() < 1e-9 and abs(depth[k] - r["depth"]) < 1e-9 * abs(r["depth"])
        assert n_ok > S // 3


def test_emulated_update_seed_and_compute_tau_batches(emu, oracle):
    orc = pytrack.Track("orc")
    rng = np.random.default_rng(6)
    S = 3000
    seeds = []
    for i in range(S):
        s = orc.seed_init(rng.uniform(0.5, 5), rng.uniform(0.2, 0.5))
        s.a, s.b = np.float32(rng.uniform(5, 30)), np.float32(rng.uniform(5, 30))
        seeds.append(s)
    x = (1.0 / rng.uniform(0.5, 5, size=S)).astype(np.float32)
    tau2 = (10.0 ** rng.uniform(-8, 0, size=S)).astype(np.float32)
    tau2[::50] = 0.0
    x[::77] = 50.0
    c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
    a, b, mu = c([s.a for s in seeds], np.float32), c([s.b for s in seeds], np.float32), c([s.mu for s in seeds], np.float32)
    zr, s2, bid = c([s.z_range for s in seeds], np.float32), c([s.sigma2 for s in seeds], np.float32), np.zeros(S, np.int32)
    ss = capi.Seeds(a.ctypes.data, b.ctypes.data, mu.ctypes.data, zr.ctypes.data, s2.ctypes.data, bid.ctypes.data)
    assert emu.svo_hip_update_seed_batch(S, _p(x), _p(tau2), C.byref(ss), None) == 0
    got = np.stack([a, b, mu, s2], axis=1).astype(np.float64)
    want = np.array([[n.a, n.b, n.mu, n.sigma2] for n in (orc.update_seed(x[i], tau2[i], seeds[i]) for i in range(S))])
    fin = np.isfinite(want).all(axis=1)
    assert np.array_equal(np.isfinite(got).all(axis=1), fin)
    g, w = got[fin], want[fin]
    # host-compiled without contraction, libm's exp on both sides: the reference's bits in all but a few seeds
    assert np.mean(np.all(g == w, axis=1)) >= 0.97
    assert np.allclose(g[:, 2], w[:, 2], rtol=2e-6, atol=0)
    assert (np.abs(g[:, 3] - w[:, 3]) <= 1e-4 * np.abs(w[:, 3]) + 1e-6 * w[:, 2] ** 2).all()
    assert np.allclose(g[:, :2], w[:, :2], rtol=1e-4, atol=0)
    # computeTau for S independent measurements
    n = 500
    t_rc = rng.normal(size=(n, 3)) * 0.3
    f = rng.normal(size=(n, 3)) * 0.3 + np.array([0, 0, 1.0])
    f /= np.linalg.norm(f, axis=1, keepdims=True)
    z = rng.uniform(0.5, 10.0, size=n)
    ang = np.arctan(1.0 / (2.0 * 315.5)) * 2.0
    tau = np.zeros(n)
    t_rc, f = np.ascontiguousarray(t_rc), np.ascontiguousarray(f)
    assert emu.svo_hip_compute_tau_batch(n, _p(t_rc), _p(f), _p(z), C.c_double(ang), _p(tau), None) == 0
    for i in range(n):
        T = np.concatenate([np.eye(3).ravel(), t_rc[i]])
        want_tau = orc.compute_tau(T, f[i], z[i], ang)
        assert np.isnan(tau[i]) == np.isnan(want_tau)
        if not np.isnan(want_tau):
            assert abs(tau[i] - want_tau) <= 1e-9 * abs(want_tau), (i, tau[i], want_tau)   # (bit-equal in the default build)


@pytest.mark.parametrize("kind", ["pinhole", "atan"])
def test_emulated_select_matches_compose_poses_cam2world(emu_default, oracle, kind):
    emu = emu_default
    cam = camera_models()[kind]
    cs = capi.camera(cam)
    rng = np.random.default_rng(77)
    for M, max_fts in ((0, 120), (1, 120), (7, 0), (130, 120), (300, 40), (1500, 120), (1500, 10000)):
        runs = rng.integers(1, 9, size=M + 1)
        cell = np.repeat(rng.permutation(M + 1), runs)[:M].astype(np.int32)
        ok = (rng.uniform(size=M) < 0.45).astype(np.int32)
        px = np.ascontiguousarray(np.stack([rng.uniform(0, cam.width, M), rng.uniform(0, cam.height, M)], axis=1).reshape(M, 2))
        level = rng.integers(0, 4, size=M).astype(np.int32)
        pos = rng.normal(size=(M, 3))
        sel_o, f_o, lvl_o, pos_o = pytrack.select_matches(cam, cell, ok, px, level, pos, max_fts)
        cap = max(M, 1)
        n, sel, f = np.zeros(1, np.int32), np.zeros(cap, np.int32), np.zeros((cap, 3))
        lvl, p, has = np.zeros(cap, np.int32), np.zeros((cap, 3)), np.zeros(cap, np.uint8)
        rc = emu.svo_hip_select_matches(C.byref(cs), M, _p(cell), _p(ok), _p(px), _p(level), _p(pos), max_fts, _p(n), _p(sel), _p(f), _p(lvl),
                                        _p(p), _p(has), None, 0, None)
        assert rc == 0, rc
        k = int(n[0])
        assert k == len(sel_o)
        assert np.array_equal(sel[:k], sel_o) and np.array_equal(lvl[:k], lvl_o)
        assert np.array_equal(p[:k], pos_o) and bool((has[:k] == 1).all())
        if k:
            assert np.abs(f[:k] - f_o).max() <= 1e-14
    # glue: SE(3) products (optionally scattered) and bearings
    n = 300
    A = np.stack([se3.exp(rng.normal(size=6) * 0.3) for _ in range(n)])
    B = np.stack([se3.exp(rng.normal(size=6) * 0.3) for _ in range(n)])
    out = np.zeros((n, 12))
    assert emu.svo_hip_compose_poses(n, _p(A), _p(B), _p(out), None, None) == 0
    assert np.abs(out - se3.mul(A, B)).max() < 1e-14
    idx = rng.permutation(n).astype(np.int32)
    out2 = np.zeros((n, 12))
    assert emu.svo_hip_compose_poses(n, _p(A), _p(B), _p(out2), _p(idx), None) == 0
    assert np.array_equal(out2[idx], out)
    px = np.ascontiguousarray(np.stack([rng.uniform(0, cam.width, n), rng.uniform(0, cam.height, n)], axis=1))
    fb = np.zeros((n, 3))
    assert emu.svo_hip_cam2world(C.byref(cs), n, _p(px), _p(fb), None) == 0
    assert np.abs(fb - synth._bearing(cam, px)).max() < 1e-14


def _host_frame_pose(T_cur_ref, q_ref, t_ref):
    """poseToRt(SE3(R, t) * T_ref) as the host forms it (oracle/orc_math.h: orc_quat_from_R, orc_se3_compose, orc_quat_to_R),
    statement for statement in Python floats (IEEE doubles, no contraction): the bits svo_hip_frame_pose_compose must give."""
    import math
    R, t = [float(x) for x in T_cur_ref[:9]], [float(x) for x in T_cur_ref[9:]]
    tr = R[0] + R[4] + R[8]
    q = [0.0] * 4
    if tr > 0.0:
        s = math.sqrt(tr + 1.0)
        q[0] = 0.5 * s
        s = 0.5 / s
        q[1] = (R[7] - R[5]) * s; q[2] = (R[2] - R[6]) * s; q[3] = (R[3] - R[1]) * s
    else:
        i = 0
        if R[4] > R[0]: i = 1
        if R[8] > R[i * 3 + i]: i = 2
        j = (i + 1) % 3; k = (j + 1) % 3
        s = math.sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0)
        q[1 + i] = 0.5 * s
        s = 0.5 / s
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * s
        q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * s
        q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * s
    b, v = [float(x) for x in q_ref], [float(x) for x in t_ref]
    ux = q[2] * v[2] - q[3] * v[1]; uy = q[3] * v[0] - q[1] * v[2]; uz = q[1] * v[1] - q[2] * v[0]
    ux += ux; uy += uy; uz += uz
    cx = q[2] * uz - q[3] * uy; cy = q[3] * ux - q[1] * uz; cz = q[1] * uy - q[2] * ux
    rt = [v[0] + q[0] * ux + cx, v[1] + q[0] * uy + cy, v[2] + q[0] * uz + cz]
    to = [t[0] + rt[0], t[1] + rt[1], t[2] + rt[2]]
    w = q[0] * b[0] - q[1] * b[1] - q[2] * b[2] - q[3] * b[3]
    x = q[0] * b[1] + q[1] * b[0] + q[2] * b[3] - q[3] * b[2]
    y = q[0] * b[2] + q[2] * b[0] + q[3] * b[1] - q[1] * b[3]
    z = q[0] * b[3] + q[3] * b[0] + q[1] * b[2] - q[2] * b[1]
    n = math.sqrt(w * w + x * x + y * y + z * z)
    w /= n; x /= n; y /= n; z /= n
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([1.0 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0 - (txx + tzz), tyz - twx, txz - twy, tyz + twx,
                     1.0 - (txx + tyy)] + to)


def frame_pose_compose_cases(rng, n=200):
    """(T_cur_ref [12], q_ref [4], t_ref [3]) with rotations in every branch of the matrix -> quaternion conversion"""
    cases = []
    for i in range(n):
        big = i % 4 == 0  # rotations beyond 120 degrees: trace <= 0
        T = se3.exp(rng.normal(size=6) * (2.5 if big else 0.05))
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        cases.append((np.ascontiguousarray(T), q, rng.normal(size=3) * 3.0))
    return cases


def host_keyframe_ranks(cam, T12, q_cur, key_pos, key_valid, table, n_kf, max_n_kfs):
    """Map::getCloseKeyframes + the reprojector's stable closest-first sort and cut, in Python floats: per frame-table entry the
    rank or -1.  T12: the composed (R | t); q_cur: its unit quaternion (w, x, y, z) as the SE3 object holds it."""
    import math
    w, x, y, z = [float(v) for v in q_cur]
    t = [float(v) for v in T12[9:]]
    dist = [-1.0] * len(table)
    for i in range(n_kf):
        for k in range(5):
            if not key_valid[i, k]:
                continue
            v = [float(c) for c in key_pos[i, k]]
            ux = y * v[2] - z * v[1]; uy = z * v[0] - x * v[2]; uz = x * v[1] - y * v[0]
            ux += ux; uy += uy; uz += uz
            cx = y * uz - z * uy; cy = z * ux - x * uz; cz = x * uy - y * ux
            f = [(v[0] + w * ux + cx) + t[0], (v[1] + w * uy + cy) + t[1], (v[2] + w * uz + cz) + t[2]]
            if f[2] < 0.0:
                continue
            px = (cam.fx * (f[0] / f[2]) + cam.cx, cam.fy * (f[1] / f[2]) + cam.cy)  # (the undistorted pinhole)
            if px[0] >= 0.0 and px[1] >= 0.0 and px[0] < cam.width and px[1] < cam.height:
                d = [t[c] - float(table[i][9 + c]) for c in range(3)]
                dist[i] = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
                break
    ranks = []
    for i in range(len(table)):
        r = -1
        if dist[i] >= 0.0:
            r = sum(1 for j in range(n_kf) if dist[j] >= 0.0 and (dist[j] < dist[i] or (dist[j] == dist[i] and j < i)))
            if r >= max_n_kfs:
                r = -1
        ranks.append(r)
    return np.array(ranks, np.int32)


def keyframe_rank_case(rng, n_kf=12, n_tab=14):
    """a frame table of n_tab poses around the origin looking at points near z = 2, key points per keyframe (some missing, some
    behind / outside), two keyframes at EXACTLY the same distance (the stable order decides)"""
    table = np.stack([se3.exp(np.concatenate([rng.normal(size=3) * 0.3, rng.normal(size=3) * 0.05])) for _ in range(n_tab)])
    table[5, 9:] = table[3, 9:]  # equal translations -> equal distances
    key_pos = np.concatenate([rng.uniform(-2.5, 2.5, size=(n_kf, 5, 2)), rng.uniform(-1.0, 4.0, size=(n_kf, 5, 1))], axis=2)
    key_valid = (rng.uniform(size=(n_kf, 5)) < 0.8).astype(np.uint8)
    key_valid[7] = 0  # a keyframe without key points is never close
    return np.ascontiguousarray(table), np.ascontiguousarray(key_pos), key_valid


def _quat_of(T12):
    """unit quaternion the host's SE3(R, t) constructor derives (orc_quat_from_R): via _host_frame_pose's first lines"""
    import math
    R = [float(x) for x in T12[:9]]
    tr = R[0] + R[4] + R[8]
    q = [0.0] * 4
    if tr > 0.0:
        s = math.sqrt(tr + 1.0); q[0] = 0.5 * s; s = 0.5 / s
        q[1] = (R[7] - R[5]) * s; q[2] = (R[2] - R[6]) * s; q[3] = (R[3] - R[1]) * s
    else:
        i = 0
        if R[4] > R[0]: i = 1
        if R[8] > R[i * 3 + i]: i = 2
        j = (i + 1) % 3; k = (j + 1) % 3
        s = math.sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0)
        q[1 + i] = 0.5 * s; s = 0.5 / s
        q[0] = (R[k * 3 + j] - R[j * 3 + k]) * s; q[1 + j] = (R[j * 3 + i] + R[i * 3 + j]) * s; q[1 + k] = (R[k * 3 + i] + R[i * 3 + k]) * s
    return q


def _composed_quat(T_cur_ref, q_ref):
    """the unit quaternion of SE3(R, t) * T_ref before it is turned into a matrix (what Frame::isVisible rotates with)"""
    import math
    q = _quat_of(T_cur_ref); b = [float(x) for x in q_ref]
    w = q[0] * b[0] - q[1] * b[1] - q[2] * b[2] - q[3] * b[3]
    x = q[0] * b[1] + q[1] * b[0] + q[2] * b[3] - q[3] * b[2]
    y = q[0] * b[2] + q[2] * b[0] + q[3] * b[1] - q[1] * b[3]
    z = q[0] * b[3] + q[3] * b[0] + q[1] * b[2] - q[2] * b[1]
    n = math.sqrt(w * w + x * x + y * y + z * z)
    return [w / n, x / n, y / n, z / n]


def test_emulated_frame_pose_compose_is_the_hosts_product(emu_default):
    """svo_hip_frame_pose_compose (round 6: the frame's pose formed on the stream behind K1, rpg_svo_amd/host/dropin/
    frame_chain.h) gives, bit for bit, the rotation matrix and translation the host gets from SE3(R, t) * T_ref -- the check
    Reprojector::reprojectMap's drop-in makes before it takes the batch the chain enqueued -- and ranks the overlapping
    keyframes like Map::getCloseKeyframes + the reprojector's sort."""
    emu = emu_default
    rng = np.random.default_rng(5)
    for T, q, t in frame_pose_compose_cases(rng):
        table = np.zeros((3, 12)); copy = np.zeros(12); out = np.zeros(12); sig = np.zeros(1, np.int32)
        assert emu.svo_hip_frame_pose_compose(_p(T), _p(q), _p(t), _p(table), 1, _p(copy), _p(out), None, 0, 0, None, None, 0, None, None,
                                              _p(sig), 7, None) == 0
        want = _host_frame_pose(T, q, t)
        assert np.array_equal(table[1], want) and np.array_equal(copy, want) and np.array_equal(out, want) and sig[0] == 7
        assert not table[0].any() and not table[2].any()
    assert emu.svo_hip_frame_pose_compose(None, _p(q), _p(t), _p(table), 1, None, None, None, 0, 0, None, None, 0, None, None, None, 0,
                                          None) == -1  # SVO_HIP_EINVAL
    # the ranking, on the undistorted pinhole (the distorted models' projection has its own bit-exact tests)
    cam = camera_models()["pinhole"]
    cs = capi.camera(cam)
    for trial in range(40):
        table, key_pos, key_valid = keyframe_rank_case(rng)
        n_tab, n_kf = len(table), len(key_pos)
        T = np.ascontiguousarray(se3.exp(rng.normal(size=6) * 0.05)); q = rng.normal(size=4); q /= np.linalg.norm(q); t = rng.normal(size=3) * 0.2
        for max_n in (3, 10):
            tab = table.copy(); rank = np.full(n_tab, 99, np.int32); rank2 = np.full(n_tab, 99, np.int32)
            assert emu.svo_hip_frame_pose_compose(_p(T), _p(q), _p(t), _p(tab), n_tab - 2, None, None, C.byref(cs), n_tab, n_kf, _p(key_pos),
                                                  _p(key_valid), max_n, _p(rank), _p(rank2), None, 0, None) == 0
            want = host_keyframe_ranks(cam, _host_frame_pose(T, q, t), _composed_quat(T, q), key_pos, key_valid, tab, n_kf, max_n)
            assert np.array_equal(rank, want) and np.array_equal(rank2, want), (trial, rank, want)
            assert (rank[n_kf:] == -1).all() and rank.max() < max_n and rank[7] == -1
            if want[3] >= 0 and want[5] >= 0:
                assert want[5] == want[3] + 1  # equal distances: map order


def test_emulated_indirect_match_batch_is_the_direct_one(emu, scene):
    """svo_hip_find_match_direct_indirect / svo_hip_select_matches_indirect (batch size read by the kernels, observation ranges
    instead of CSR offsets: what follows svo_hip_reproject_map on the mirror's path) against the plain entry points on the
    same trials: identical outputs, nothing written beyond the device-side batch size (tests/test_map_mirror_gpu.py)."""
    imgs = scene.images.cpu().numpy()
    layout, store = _store(emu, imgs)
    T = scene.T_f_w.copy()
    T[scene.cur] = scene.T_cur_prior
    T = np.ascontiguousarray(T)
    slots = np.arange(T.shape[0], dtype=np.int32)
    frames = capi.Frames(T.shape[0], 0, slots.ctypes.data, T.ctypes.data)
    M = len(scene.obs)
    ptr = np.zeros(M + 1, np.int32)
    flat = []
    for i, o in enumerate(scene.obs):
        ptr[i + 1] = ptr[i] + len(o)
        flat.extend(o)
    c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
    o_frame, o_level = c([o[0] for o in flat], np.int32), c([o[3] for o in flat], np.int32)
    o_px, o_f = c([o[1] for o in flat], np.float64), c([o[2] for o in flat], np.float64)
    o_type, o_grad = c([o[4] for o in flat], np.uint8), c([o[5] for o in flat], np.float64)
    obs = capi.Features(o_frame.ctypes.data, o_level.ctypes.data, o_type.ctypes.data, o_px.ctypes.data, o_f.ctypes.data, o_grad.ctypes.data)
    cam = capi.camera(scene.cam)
    emu.svo_hip_match_workspace_bytes.restype = C.c_size_t
    cap = M + 37
    pad = lambda x: np.ascontiguousarray(np.concatenate([x, np.zeros((cap - M,) + x.shape[1:], x.dtype)]))
    cur, pos = np.full(M, scene.cur, np.int32), c(scene.pt_pos, np.float64)

    def outputs(n, fill):
        return dict(px=pad(c(scene.px_init, np.float64))[:n].copy(), ok=np.full(n, fill, np.int32), ref_obs=np.zeros(n, np.int32),
                    sl=np.zeros(n, np.int32), A=np.zeros((n, 4)), patches=np.zeros((n, 100), np.uint8))

    ref = outputs(M, 0)
    ws = np.full(emu.svo_hip_match_workspace_bytes(cap) + 256, 0xFF, np.uint8)   # (poisoned: NaN / -1 to whoever reads scratch it did not write)
    rc = emu.svo_hip_find_match_direct(C.byref(layout), _p(store), C.byref(cam), C.byref(frames), M, _p(cur), _p(pos), _p(ptr), C.byref(obs),
                                       5, 10, _p(ref["px"]), _p(ref["ok"]), _p(ref["ref_obs"]), _p(ref["sl"]), _p(ref["A"]), _p(ref["patches"]),
                                       _p(ws), C.c_size_t(ws.size), None)
    assert rc == 0, rc
    res = outputs(cap, -7)
    d_M = np.array([M], np.int32)
    ob_begin, ob_end, cur_p, pos_p = pad(ptr[:-1].copy()), pad(ptr[1:].copy()), pad(cur), pad(pos)
    rc = emu.svo_hip_find_match_direct_indirect(C.byref(layout), _p(store), C.byref(cam), C.byref(frames), cap, _p(d_M), _p(cur_p), _p(pos_p),
                                                _p(ob_begin), _p(ob_end), C.byref(obs), 5, 10, _p(res["px"]), _p(res["ok"]), _p(res["ref_obs"]),
                                                _p(res["sl"]), _p(res["A"]), _p(res["patches"]), _p(ws), C.c_size_t(ws.size), None)
    assert rc == 0, rc
    for k in ref:
        assert np.array_equal(res[k][:M], ref[k]), k
    assert (res["ok"][M:] == -7).all() and ref["ok"].sum() > M // 2
    cell = (np.arange(M) // 3).astype(np.int32)
    cell_p = pad(cell)   # (kept in a name: a temporary would be gone before the call reads it)
    outs = []
    for indirect in (False, True):
        n, sel = np.zeros(1, np.int32), np.full(121, -1, np.int32)
        f, pos_o, lvl_o, has = np.zeros((121, 3)), np.zeros((121, 3)), np.zeros(121, np.int32), np.zeros(121, np.uint8)
        if indirect:
            rc = emu.svo_hip_select_matches_indirect(C.byref(cam), cap, _p(d_M), _p(cell_p), _p(res["ok"]), _p(res["px"]), _p(res["sl"]),
                                                     _p(pos_p), 120, _p(n), _p(sel), _p(f), _p(lvl_o), _p(pos_o), _p(has), None, 0, None)
        else:
            rc = emu.svo_hip_select_matches(C.byref(cam), M, _p(cell), _p(ref["ok"]), _p(ref["px"]), _p(ref["sl"]), _p(pos), 120, _p(n),
                                            _p(sel), _p(f), _p(lvl_o), _p(pos_o), _p(has), None, 0, None)
        assert rc == 0, rc
        outs.append((n, sel, f, lvl_o, pos_o, has))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert outs[0][0][0] > 20
