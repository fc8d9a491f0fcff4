"""The device math of the kernels, compiled for the CPU and checked against the oracle WITHOUT a GPU.

rpg_svo_amd/csrc/device_math.h, track_math.h and matcher_device.h hold the formulas every kernel after K1 runs (SE(3)
exp / compose / inverse as Sophus does them on quaternions, the three camera models of vikit, the affine warp matrix and
its search level, the LDLT solves).  tests/host/device_math_on_host.cpp compiles those same headers with g++
(SVO_HOST_MATH_TEST) behind a C interface; here every function is compared with the oracle's restatement of the
reference on seeded inputs.  Both sides are built without floating-point contraction, and where both follow the
reference's order of operations the results are equal bit for bit -- asserted where that holds, to 1e-12 otherwise.
The bits the GPU produces are the business of the `-m gpu` parity tests."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle  # noqa: E402

SRC = os.path.join(ROOT, "tests", "host", "device_math_on_host.cpp")
LIB = os.path.join(ROOT, "build", "emu", "libdevice_math_on_host.so")
D = C.POINTER(C.c_double)
F = C.POINTER(C.c_float)


def _p(a, t=D):
    return a.ctypes.data_as(t)


def _build_host_lib(lib_path, defines=()):
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    deps = [SRC] + [os.path.join(ROOT, "rpg_svo_amd", "csrc", h)
                    for h in ("device_math.h", "track_math.h", "matcher_device.h", "seed_math.h", "align_lanes.h", "pyr_addr.h", "warp_sample.h")]
    if not os.path.exists(lib_path) or any(os.path.getmtime(d) > os.path.getmtime(lib_path) for d in deps):
        # ROCm's clang++ as a plain C++ compiler for the host (the lane bodies use clang's ext_vector_type pairs)
        cxx = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang++")
        if not os.path.exists(cxx):
            pytest.skip("no ROCm clang++ to compile the kernels' headers for the host")
        subprocess.run([cxx, "-std=c++17", "-O2", "-ffp-contract=off", "-fno-math-errno", "-fPIC", "-shared", "-Wall",
                        "-Wno-unknown-pragmas", "-Wno-pass-failed", *[f"-D{d}" for d in defines], "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "rpg_svo_amd", "csrc"), SRC, "-o", lib_path], check=True)
    lib = C.CDLL(lib_path)
    lib.hm_best_search_level.restype = C.c_int
    lib.hm_floor_to_int.restype = C.c_int
    lib.hm_floor_to_int.argtypes = [C.c_float]
    lib.hm_sincos_small.argtypes = [C.c_double, D, D]
    return lib


@pytest.fixture(scope="module")
def hm():
    return _build_host_lib(LIB)


def _random_pose(rng, angle=0.5, trans=1.0):
    xi = np.concatenate([rng.uniform(-trans, trans, 3), rng.normal(size=3)])
    xi[3:] *= rng.uniform(0, angle) / np.linalg.norm(xi[3:])
    return pyoracle.se3_exp(xi)


def test_se3_exp_follows_sophus(hm):
    """se3_exp (quaternion form, own sin / cos series) against the oracle's Sophus SE3::exp: tiny angles, the angles a
    Gauss-Newton step produces, and large ones (the series halves its argument and doubles back).  Sophus forms the
    translation's coefficients as (1 - cos t) / t^2 and (t - sin t) / t^3, which cancel for 1e-10 < t < 1e-5 (below 1e-10
    it switches to V = R): there the two sides -- libm on one, the series on the other -- round the cancellation differently,
    by up to ~2e-16 / t relative to |upsilon| (a GN step has |upsilon| ~ t, i.e. 1e-16 absolute)."""
    rng = np.random.default_rng(1)
    for scale in (0.0, 1e-12, 1e-8, 1e-6, 1e-4, 1e-2, 0.3, 1.0, 3.0):
        for _ in range(100):
            xi = np.concatenate([rng.uniform(-1, 1, 3), rng.normal(size=3) * scale])
            T = np.empty(12)
            hm.hm_se3_exp(_p(xi), _p(T))
            T_o = pyoracle.se3_exp(xi)
            theta = np.linalg.norm(xi[3:])
            assert np.abs(T[:9] - T_o[:9]).max() < 5e-15, (scale, xi)
            tol_t = 1e-14 * max(1.0, np.abs(T_o).max()) + (4e-16 / theta * np.linalg.norm(xi[:3]) if theta >= 1e-10 else 0.0)
            assert np.abs(T[9:] - T_o[9:]).max() < tol_t, (scale, xi, np.abs(T[9:] - T_o[9:]).max(), tol_t)
            # a Gauss-Newton step: translation and rotation parts of the same magnitude
            xi_s = xi.copy()
            xi_s[:3] *= max(theta, 1e-12)
            hm.hm_se3_exp(_p(xi_s), _p(T))
            T_s = pyoracle.se3_exp(xi_s)
            assert np.abs(T - T_s).max() < 5e-15 * max(1.0, np.abs(T_s).max()), (scale, xi_s)


def test_sincos_small_is_accurate_to_the_last_bits(hm):
    """|x| <= 0.5 (half the angle of any Gauss-Newton step): the series alone, a few 1e-16.  Larger arguments are halved
    until they fit and doubled back with the double-angle identities: every doubling doubles the error."""
    rng = np.random.default_rng(2)

    def worst(xs):
        w = 0.0
        for x in xs:
            s, c = C.c_double(), C.c_double()
            hm.hm_sincos_small(float(x), C.byref(s), C.byref(c))
            w = max(w, abs(s.value - np.sin(x)), abs(c.value - np.cos(x)))
        return w

    assert worst(np.concatenate([rng.uniform(-0.5, 0.5, 4000), [0.0, 0.5, -0.5, 1e-300, 1e-9]])) < 4e-16
    assert worst(rng.uniform(-4, 4, 1000)) < 1e-14
    assert worst(rng.uniform(-40, 40, 1000)) < 1e-13


def test_se3_f32_series_agree_with_the_f64_exponential(hm):
    """K1's f32 solver path: the short series (|omega| < 0.01) and the long one against the f64 exponential."""
    rng = np.random.default_rng(3)
    worst = 0.0
    for scale in (1e-6, 1e-3, 0.009, 0.011, 0.1):
        for _ in range(100):
            xi = np.concatenate([rng.uniform(-0.1, 0.1, 3), rng.normal(size=3) * scale]).astype(np.float32)
            q, t = np.empty(4, np.float32), np.empty(3, np.float32)
            hm.hm_se3_exp_f32(_p(xi, F), _p(q, F), _p(t, F))
            T = pyoracle.se3_exp(xi.astype(np.float64))
            w, x, y, z = q.astype(np.float64)
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                          [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                          [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
            worst = max(worst, np.abs(R.ravel() - T[:9]).max(), np.abs(t - T[9:]).max())
    assert worst < 5e-7, worst


def test_compose_inverse_and_frame_position(hm):
    rng = np.random.default_rng(4)
    worst = 0.0
    for _ in range(200):
        A, B = _random_pose(rng, 2.0), _random_pose(rng, 2.0)
        out = np.empty(12)
        hm.hm_se3_mul(_p(A), _p(B), _p(out))
        worst = max(worst, np.abs(out - pyoracle.se3_mul(A, B)).max())
        hm.hm_se3_inv(_p(A), _p(out))
        inv = pyoracle.se3_inv(A)
        worst = max(worst, np.abs(out - inv).max())
        pos = np.empty(3)
        hm.hm_frame_pos(_p(A), _p(pos))  # Frame::pos() = T_f_w.inverse().translation()
        worst = max(worst, np.abs(pos - inv[9:]).max())
        R2 = np.empty(9)
        hm.hm_quat_round_trip(_p(np.ascontiguousarray(A[:9])), _p(R2))
        worst = max(worst, np.abs(R2 - A[:9]).max())
    assert worst < 1e-14, worst


CAMS = [  # (model, k, size, d): the three vk::AbstractCamera implementations (include/svo_hip.h)
    (0, (315.5, 315.5, 376.0, 240.0), (752, 480), (0, 0, 0, 0, 0)),
    (1, (420.0, 418.0, 370.0, 236.0), (752, 480), (-0.28, 0.07, 1e-4, -2e-4, 0.0)),
    (2, (0.51 * 752, 0.79 * 480, 0.495 * 752 - 0.5, 0.51 * 480 - 0.5), (752, 480), None),  # ATAN: s = 0.93
]


def _cam_struct(model, k, size, d):
    from types import SimpleNamespace
    if model == 2:
        s = 0.93
        d = (s, 1.0 / s, 2.0 * np.tan(s / 2.0), 1.0 / (2.0 * np.tan(s / 2.0)), 0.0)
    ns = SimpleNamespace(fx=k[0], fy=k[1], cx=k[2], cy=k[3], width=size[0], height=size[1], model=model, d=d)
    return ns, np.array(k, np.float64), np.array(d, np.float64)


@pytest.mark.parametrize("model,k,size,d", CAMS)
def test_camera_models_match_the_oracle(hm, model, k, size, d):
    """world2cam / cam2world of the pinhole, the radial-tangential pinhole (cv::undistortPoints' float round trips) and
    the ATAN camera: equal to the oracle bit for bit."""
    ns, kk, dd = _cam_struct(model, k, size, d)
    pc = pyoracle.make_cam(ns)
    lib = pyoracle.lib()
    rng = np.random.default_rng(5 + model)
    px = np.stack([rng.uniform(0, size[0], 500), rng.uniform(0, size[1], 500)], 1)
    f_o = np.empty((500, 3))
    lib.orc_cam2world(C.byref(pc), C.c_int(500), _p(np.ascontiguousarray(px)), _p(f_o))
    for i in range(500):
        f = np.empty(3)
        hm.hm_cam2world(_p(kk), size[0], size[1], model, _p(dd), _p(np.ascontiguousarray(px[i])), _p(f))
        assert np.array_equal(f, f_o[i]), (i, f, f_o[i])
        # and back through world2cam, against the oracle's reprojectPoint (identity pose)
        xyz = f * rng.uniform(0.5, 5.0)
        p = np.empty(2)
        hm.hm_world2cam(_p(kk), size[0], size[1], model, _p(dd), _p(xyz), _p(p))
        p_o = np.empty(2)
        T = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float64)
        lib.orc_reproject_point(C.byref(pc), _p(T), _p(xyz), C.c_int(30), C.c_int(26), _p(p_o))
        assert np.array_equal(p, p_o), (i, p, p_o)
        assert np.abs(p - px[i]).max() < (1e-9 if model != 1 else 1.0)  # (radtan: cv::undistortPoints stops after five fixed-point iterations, in float: 0.25 px off at the image border for k1 = -0.28)


@pytest.mark.parametrize("model,k,size,d", CAMS)
def test_affine_warp_matrix_and_search_level(hm, model, k, size, d):
    """warp::getWarpMatrixAffine / getBestSearchLevel (matcher.cpp:33-70) as the kernels compute them, against the oracle:
    equal bit for bit, for every camera model and reference level."""
    ns, kk, dd = _cam_struct(model, k, size, d)
    pc = pyoracle.make_cam(ns)
    lib = C.CDLL(pyoracle.lib()._name)
    lib.orc_get_best_search_level.restype = C.c_int
    rng = np.random.default_rng(9 + model)
    for i in range(300):
        px = np.array([rng.uniform(40, size[0] - 40), rng.uniform(40, size[1] - 40)])
        f = np.empty(3)
        hm.hm_cam2world(_p(kk), size[0], size[1], model, _p(dd), _p(px), _p(f))
        depth = rng.uniform(0.5, 8.0)
        T = _random_pose(rng, 0.6, 0.8)
        level = int(rng.integers(0, 4))
        A, A_o = np.empty(4), np.empty(4)
        hm.hm_warp_matrix_affine(_p(kk), size[0], size[1], model, _p(dd), _p(px), _p(f), C.c_double(depth), _p(T), level, _p(A))
        lib.orc_get_warp_matrix_affine(C.byref(pc), C.byref(pc), _p(px), _p(f), C.c_double(depth), _p(T), C.c_int(level), _p(A_o))
        assert np.array_equal(A, A_o), (i, A, A_o)
        for max_level in (0, 2, 4):
            assert hm.hm_best_search_level(_p(A), max_level) == lib.orc_get_best_search_level(_p(A_o), C.c_int(max_level))


def test_ldlt_solves(hm):
    """The pose optimizer's pivoted LDLT (Eigen's algorithm) and K1's packed unpivoted one on 6 x 6 normal equations of
    a wide range of conditioning."""
    rng = np.random.default_rng(12)
    for trial in range(200):
        J = rng.normal(size=(40, 6)) * 10.0 ** rng.uniform(-3, 3, 6)
        H = J.T @ J
        b = rng.normal(size=6) * np.sqrt(np.diag(H))
        x_ref = np.linalg.solve(H, b)
        x = np.empty(6)
        hm.hm_ldlt6_solve_pivoted(_p(np.ascontiguousarray(H)), _p(b), _p(x))
        scale = np.abs(x_ref) + 1e-300
        cond = np.linalg.cond(H)
        assert np.abs((x - x_ref) / scale).max() < 1e-15 * cond * 50 + 1e-9, (trial, cond)
        x_o = pyoracle.ldlt_solve(H, b)
        assert np.abs((x - x_o) / scale).max() < 1e-15 * cond * 50 + 1e-9
        H21 = np.array([H[i, j] for i in range(6) for j in range(i, 6)])
        hm.hm_ldlt6_solve_packed(_p(H21), _p(b), _p(x))
        assert np.abs((x - x_ref) / scale).max() < 1e-15 * cond * 200 + 1e-9, (trial, cond)


def test_inv3f_and_floor(hm):
    rng = np.random.default_rng(13)
    for _ in range(200):
        J = rng.normal(size=(64, 3)).astype(np.float32)
        J[:, 2] = 1.0
        Hm = (J.T @ J).astype(np.float32)
        r = np.empty(9, np.float32)
        hm.hm_inv3f(_p(np.ascontiguousarray(Hm.ravel()), F), _p(r, F))
        assert np.abs(r.reshape(3, 3).astype(np.float64) @ Hm.astype(np.float64) - np.eye(3)).max() < 1e-3
    for x in (0.0, 0.5, 0.999999, 1.0, 17.25, 639.9999, -0.0, 3.0000002):
        assert hm.hm_floor_to_int(x) == int(np.floor(np.float32(x)))


# ---- the depth filter's closed-form pieces (csrc/seed_math.h) -------------------------------------------------------
def test_update_seed_against_the_oracle(hm):
    """DepthFilter::updateSeed (f32 with the reference's double sub-expressions, boost's normal pdf through the f64 exp):
    the kernels' function compiled for the CPU against the oracle's, chained over 30 measurements per seed -- equal bit
    for bit on this host (the oracle's expf and the f64 exp rounded to float agree on all but ~0.3 % of arguments; the
    chain is asserted to 1e-6 relative and the share of identical seeds to 97 %)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle import pytrack
    lib = C.CDLL(pyoracle.lib()._name)  # (an own handle: prototypes set here must not leak into the other tests' calls)
    hm.hm_update_seed.argtypes = [C.c_float, C.c_float, C.c_float, F]
    lib.orc_update_seed.argtypes = [C.c_float, C.c_float, C.POINTER(pytrack.Seed)]
    lib.orc_seed_init.argtypes = [C.POINTER(pytrack.Seed), C.c_float, C.c_float]
    rng = np.random.default_rng(21)
    identical = 0
    n = 400
    for k in range(n):
        seed = pytrack.Seed()
        depth_mean, depth_min = rng.uniform(1.0, 4.0), rng.uniform(0.3, 0.9)
        lib.orc_seed_init(C.byref(seed), depth_mean, depth_min)
        st = np.array([seed.a, seed.b, seed.mu, seed.sigma2], np.float32)
        zr = seed.z_range
        true_inv = 1.0 / rng.uniform(depth_min * 1.2, depth_mean * 2)
        same = True
        for it in range(30):
            outlier = rng.uniform() < 0.2
            x = np.float32(rng.uniform(0.05, 1.0 / depth_min) if outlier else true_inv + rng.normal() * 0.01)
            tau2 = np.float32(rng.uniform(1e-5, 1e-3))
            lib.orc_update_seed(x, tau2, C.byref(seed))
            hm.hm_update_seed(x, tau2, zr, _p(st, F))
            ref = np.array([seed.a, seed.b, seed.mu, seed.sigma2], np.float32)
            same = same and np.array_equal(st, ref)
            assert np.allclose(st, ref, rtol=1e-6, atol=0), (k, it, st, ref)
        identical += same
    assert identical >= 0.97 * n, identical


def test_compute_tau_and_triangulation(hm):
    """DepthFilter::computeTau against the oracle, bit for bit; depthFromTriangulation recovers the depth of a point seen
    from two poses (the oracle keeps that function private: checked on geometry)."""
    lib = C.CDLL(pyoracle.lib()._name)  # (an own handle, see above)
    lib.orc_compute_tau.restype = C.c_double
    hm.hm_compute_tau.restype = C.c_double
    hm.hm_compute_tau.argtypes = [D, D, C.c_double, C.c_double]
    lib.orc_compute_tau.argtypes = [D, D, C.c_double, C.c_double]
    rng = np.random.default_rng(22)
    px_error_angle = np.arctan(1.0 / (2.0 * 315.5)) * 2.0
    for k in range(500):
        T_ref_cur = _random_pose(rng, 0.4, 0.5)
        f = rng.normal(size=3) * 0.3 + np.array([0, 0, 1.0])
        f /= np.linalg.norm(f)
        z = rng.uniform(0.5, 10.0)
        a = hm.hm_compute_tau(_p(T_ref_cur), _p(f), z, px_error_angle)
        b = lib.orc_compute_tau(_p(T_ref_cur), _p(f), z, px_error_angle)
        assert a == b or (np.isnan(a) and np.isnan(b)), (k, a, b)
        # triangulation: the point at depth z along f in the reference frame, seen from the search frame
        T_search_ref = pyoracle.se3_inv(T_ref_cur)
        p_ref = f * z
        R, t = T_search_ref[:9].reshape(3, 3), T_search_ref[9:]
        p_cur = R @ p_ref + t
        if p_cur[2] < 0.1 or np.linalg.norm(t) < 0.05:
            continue
        f_cur = p_cur / np.linalg.norm(p_cur)
        depth = C.c_double()
        ok = hm.hm_depth_from_triangulation(_p(T_search_ref), _p(f), _p(f_cur), C.byref(depth))
        if ok:  # (matcher.cpp:116: the two bearings nearly parallel -> no depth)
            assert abs(depth.value - z) < 1e-8 * max(1.0, z) / max(1e-3, np.linalg.norm(np.cross(R @ f, f_cur)) ** 2), (k, depth.value, z)


# ---- K3's lane bodies (csrc/align_lanes.h) on the tiled store --------------------------------------------------------
def _texture(rng, h, w):
    """band-limited noise: smooth enough for Lucas-Kanade, textured enough to converge"""
    img = rng.uniform(0, 255, (h // 4 + 3, w // 4 + 3))
    ys, xs = np.arange(h) / 4.0, np.arange(w) / 4.0
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = img[y0][:, x0] * (1 - fy) * (1 - fx) + img[y0][:, x0 + 1] * (1 - fy) * fx
    b = img[y0 + 1][:, x0] * fy * (1 - fx) + img[y0 + 1][:, x0 + 1] * fy * fx
    return np.clip(a + b + rng.normal(0, 2.0, (h, w)), 0, 255).astype(np.uint8)


def _tiled(hm, img):
    h, w = img.shape
    pitch = (w + 15) & ~15
    hm.hm_level_bytes.restype = C.c_longlong
    hm.hm_px_off.restype = C.c_uint
    buf = np.zeros(hm.hm_level_bytes(pitch, h) + 256, np.uint8)  # (+ slack: the window rows are read as 12-byte runs)
    ys, xs = np.mgrid[0:h, 0:w]
    off = (ys >> 3) * 8 * pitch + (ys & 7) * 16 + (xs >> 4) * 128 + (xs & 15)
    assert off[5, 37] == hm.hm_px_off(37, 5, pitch) and off[h - 1, w - 1] == hm.hm_px_off(w - 1, h - 1, pitch)
    buf[off.ravel()] = img.ravel()
    return buf, pitch


@pytest.mark.parametrize("phase", [0, 3])
def test_align2d_lane_is_the_reference_bit_for_bit(hm, phase):
    """align2D of one trial as the kernel's lane runs it -- on the TILED store, window rows as 12-byte runs cut with
    v_alignbyte, packed-f32 pixel loop -- against the oracle on the row-major image: the same verdict and the same refined
    pixel, in every bit, whether the ten iterations run in one go or in phases of three with the state parked in between
    (as the phased launches do).  Includes trials that leave the image and trials that do not converge."""
    from oracle import pytrack
    tr = pytrack.Track("orc")
    rng = np.random.default_rng(31)
    img = _texture(rng, 120, 160)
    buf, pitch = _tiled(hm, img)
    n_conv = n_fail = 0
    for k in range(400):
        x0, y0 = int(rng.integers(8, 152)), int(rng.integers(8, 112))
        pwb = np.ascontiguousarray(img[y0 - 5:y0 + 5, x0 - 5:x0 + 5])
        patch = np.ascontiguousarray(pwb[1:9, 1:9])
        start = np.array([x0 + rng.uniform(-2.5, 2.5), y0 + rng.uniform(-2.5, 2.5)])
        if k % 10 == 0:
            start = np.array([rng.choice([3.5, 156.2]), y0])  # next to the border: the loop leaves at once
        ok_o, px_o = tr.align2d(img, pwb, patch, 10, start)
        px = start.copy()
        ok = hm.hm_align2d(_p(buf, C.POINTER(C.c_uint8)), 160, 120, pitch, _p(pwb, C.POINTER(C.c_uint8)), 10, phase, _p(px))
        assert bool(ok) == ok_o, (k, start, px, px_o)
        if ok_o:  # (the reference writes the pixel back only on convergence)
            assert np.array_equal(px, px_o), (k, start, px, px_o)
            n_conv += 1
        else:
            n_fail += 1
    assert n_conv > 200 and n_fail > 30, (n_conv, n_fail)


@pytest.mark.parametrize("phase", [0, 4])
def test_align1d_lane_is_the_reference_bit_for_bit(hm, phase):
    """The same for align1D (edgelets / the epipolar 1-D refinement), h_inv included."""
    from oracle import pytrack
    tr = pytrack.Track("orc")
    rng = np.random.default_rng(32)
    img = _texture(rng, 120, 160)
    buf, pitch = _tiled(hm, img)
    n_conv = 0
    for k in range(300):
        x0, y0 = int(rng.integers(8, 152)), int(rng.integers(8, 112))
        pwb = np.ascontiguousarray(img[y0 - 5:y0 + 5, x0 - 5:x0 + 5])
        patch = np.ascontiguousarray(pwb[1:9, 1:9])
        ang = rng.uniform(0, 2 * np.pi)
        d = np.array([np.cos(ang), np.sin(ang)], np.float32)
        s = rng.uniform(-2.0, 2.0)
        start = np.array([x0 + s * d[0], y0 + s * d[1]], np.float64)
        ok_o, px_o, h_o = tr.align1d(img, d, pwb, patch, 10, start)
        px = start.copy()
        h_inv = C.c_double(0)
        ok = hm.hm_align1d(_p(buf, C.POINTER(C.c_uint8)), 160, 120, pitch, _p(pwb, C.POINTER(C.c_uint8)), _p(d, F), 10, phase, _p(px),
                           C.byref(h_inv))
        assert bool(ok) == ok_o, (k, start, px, px_o)
        if ok_o:
            assert np.array_equal(px, px_o), (k, start, px, px_o)
            assert h_inv.value == h_o, (k, h_inv.value, h_o)
            n_conv += 1
    assert n_conv > 100, n_conv


# ---- warp_kernel's sample arithmetic (csrc/warp_sample.h) ------------------------------------------------------------
def test_warp_samples_are_the_reference_bit_for_bit(hm):
    """warp::warpAffine's 10 x 10 patch as warp_kernel's lanes compute it -- one output column per lane, floor and fraction
    as the single-instruction forms, a region in rows of 48 bytes -- against the oracle: the same bytes, for the checked
    form (samples outside the image are 0) and the unchecked form (box of the corner samples inside the image), over
    rotations, scales 0.4-2.4, search levels 0-2 and reference levels 0-2."""
    from oracle import pytrack
    tr = pytrack.Track("orc")
    rng = np.random.default_rng(41)
    img = _texture(rng, 64, 48)  # 48 bytes wide: the level is a region in the kernel's layout as it is
    U8 = C.POINTER(C.c_uint8)
    counts = {0: 0, 1: 0}
    zeros_seen = 0
    for k in range(1500):
        ang, scale = rng.uniform(0, 2 * np.pi), rng.uniform(0.4, 2.4)
        shear = rng.uniform(-0.2, 0.2)
        A = scale * np.array([[np.cos(ang), -np.sin(ang) + shear], [np.sin(ang), np.cos(ang)]])
        level_ref, search_level = int(rng.integers(0, 3)), int(rng.integers(0, 3))
        centre = np.array([rng.uniform(-2, 50), rng.uniform(-2, 66)])  # (level coordinates; some near or over the border)
        px_ref = centre * (1 << level_ref)
        ok_o, patch_o = tr.warp_affine(A, img, px_ref, level_ref, search_level, 5)
        for mode in (0, 1):
            out = np.zeros(100, np.uint8)
            r = hm.hm_warp_patch(_p(img, U8), 64, _p(np.ascontiguousarray(A.ravel())), _p(px_ref), level_ref, search_level, mode, _p(out, U8))
            if r < 0:
                continue
            assert bool(r) == ok_o
            assert np.array_equal(out, patch_o), (k, mode, A, px_ref, level_ref, search_level, out.reshape(10, 10), patch_o.reshape(10, 10))
            counts[mode] += 1
        zeros_seen += int((patch_o == 0).sum() > 20)
    assert counts[0] == 1500 and counts[1] > 150, counts
    assert zeros_seen > 100  # (patches hanging over the border exercised the bounds test)


def test_compute_tau_algebraic_form(hm):
    """What seed_finish_kernel runs: tau = z_plus - z from the two cosines, their square-root sines and the angle-sum formulas
    instead of acos / sin.  Against the oracle's computeTau: the same number up to the rounding that a small parallax angle
    (sin(gamma_plus): the difference of three angles there, of two products here) and the subtraction z_plus - z amplify in
    either form -- measured below: worst relative difference of tau 7e-13 (1e-10 where the parallax is below the pixel error and tau is
    huge or negative: no measurement the filter could use) over baselines of 5 cm - 1 m, depths of
    0.5 - 30 m and bearings up to 40 degrees off axis, five orders below the f32 the filter squares it into; NaN where the
    reference's acos is NaN (a bearing that is not a unit vector beyond rounding)."""
    lib = C.CDLL(pyoracle.lib()._name)
    lib.orc_compute_tau.restype = C.c_double
    lib.orc_compute_tau.argtypes = [D, D, C.c_double, C.c_double]
    hm.hm_compute_tau_algebraic.restype = C.c_double
    hm.hm_compute_tau_algebraic.argtypes = [D, D, C.c_double, C.c_double]
    rng = np.random.default_rng(23)
    worst = worst_degenerate = 0.0
    for k in range(4000):
        fx = rng.choice([160.0, 315.5, 400.0, 800.0])
        px_error_angle = np.arctan(1.0 / (2.0 * fx)) * 2.0
        T_ref_cur = _random_pose(rng, 0.4, rng.choice([0.05, 0.2, 1.0]))
        f = rng.normal(size=3) * 0.35 + np.array([0, 0, 1.0])
        f /= np.linalg.norm(f)
        z = rng.uniform(0.5, 30.0)
        a = hm.hm_compute_tau_algebraic(_p(T_ref_cur), _p(f), z, px_error_angle)
        b = lib.orc_compute_tau(_p(T_ref_cur), _p(f), z, px_error_angle)
        assert np.isnan(a) == np.isnan(b), (k, a, b)
        if np.isnan(b):
            continue
        rel = abs(a - b) / abs(b)
        if 0.0 < b < z:   # a measurement the filter can use: the depth interval is narrower than the depth itself
            worst = max(worst, rel)
            assert rel <= 1e-10, (k, a, b, z)
        else:             # parallax below the pixel error: tau huge or negative, the quotient ill-conditioned in both forms
            worst_degenerate = max(worst_degenerate, rel)
            assert rel <= 1e-6, (k, a, b, z)
    print('worst relative difference of tau:', worst, '(degenerate geometry:', worst_degenerate, ')')
    assert worst > 0.0   # (it IS a different evaluation; if this ever reads 0 the flag did not reach the build)
