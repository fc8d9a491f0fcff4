"""K1 -- the headline kernel -- without a GPU: rpg_svo_amd/csrc/sparse_align.hip in the host-emulated library
(tests/emu_build.py): svo_hip_sparse_align runs sia_kernel, one workgroup of 256 work-items per frame with its LDS tiles,
wave reductions (DPP, permlane swaps), the two-wave solve and its barriers, on the CPU, against the oracle's
SparseImgAlign::run with the requirements of the GPU test (tests/test_sparse_align_gpu.py): poses to 1e-4 in SE(3)
log-norm (measured here: see the assert), tracked counts identical, iteration counts per level equal for nearly every
frame."""
import ctypes as C

import numpy as np
import pytest

from helpers import make_batch, marshal_problem, run_oracle
from rpg_svo_amd import capi, se3, synth


@pytest.fixture(scope="module", params=[0], ids=["default"])
def emu(request):
    from emu_build import build_emulated
    from emu_build import BUILDS
    return build_emulated(BUILDS[request.param])


@pytest.fixture(scope="module")
def emu_default():
    """(the default build only: tests of kernels no queued flag touches)"""
    from emu_build import build_emulated
    return build_emulated(())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)  # (the pointer object keeps the array alive)


def run_emulated(emu, b, max_level, min_level, n_iter=30, entry="svo_hip_sparse_align"):
    imgs = np.ascontiguousarray(b.images)
    n, h, w = imgs.shape
    layout = capi.pyr_layout(w, h, b.n_levels)
    store = np.zeros(capi.pyr_store_bytes(layout, n), np.uint8)
    assert emu.svo_hip_pyramid_build_tiled(C.byref(layout), _p(store), 0, n, _p(imgs), C.c_longlong(h * w), w, capi.HALFSAMPLE_AUTO, 0, None) == 0
    T_cr, xyz = marshal_problem(b.T_ref_w, b.T_cur_w, b.f, b.pos)
    c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
    ref_slot, cur_slot, nn = c(b.ref_slot, np.int32), c(b.cur_slot, np.int32), c(b.n, np.int32)
    px, xyz, T_in, valid = c(b.px, np.float64), c(xyz, np.float64), c(T_cr, np.float64), c(b.has_point, np.uint8)
    B, ns = px.shape[0], px.shape[1]
    d = tuple(getattr(b.cam, "d", (0.0,) * 5))
    P = capi.SiaParams(b.cam.fx, b.cam.fy, b.cam.cx, b.cam.cy, max_level, min_level, n_iter, int(getattr(b.cam, "model", 0)), 1e-6,
                       (C.c_double * 5)(*d))
    T_out, H = np.zeros((B, 12)), np.zeros((B, 36))
    n_tracked, iters = np.zeros(B, np.int32), np.zeros((B, capi.MAX_LEVELS), np.int32)
    chi2, status = np.zeros(B), np.zeros(B, np.int32)
    rc = getattr(emu, entry)(C.byref(layout), _p(store), B, _p(ref_slot), _p(cur_slot), _p(nn), ns, _p(px), _p(xyz), _p(valid),
                                  C.byref(P), _p(T_in), _p(T_out), _p(H), _p(n_tracked), _p(iters), _p(chi2), _p(status), None)
    assert rc == 0, rc
    return se3.mul(T_out, b.T_ref_w), n_tracked, iters, H, status


def test_emulated_sparse_align_follows_the_oracle(emu, oracle):
    seq = synth.make_sequence(7, 200)
    b = make_batch(seq, [(i, i + 1) for i in range(6)], 4)
    T_h, n_tracked, iters, H, status = run_emulated(emu, b, 3, 0)
    T_o, res_o, _ = run_oracle(oracle, b, 3, 0)
    d = se3.log_norm(T_h, T_o)
    err = se3.log_norm(T_h, b.T_gt_w)
    assert d.max() <= 1e-5, d                      # (the GPU test's bound is 1e-4)
    assert err.max() < 5e-3, err                   # and the motion is recovered
    assert np.array_equal(n_tracked, np.array([r["n_tracked"] for r in res_o]))
    it_o = np.array([r["iters"][:4] for r in res_o])
    assert np.mean(np.all(iters[:, :4] == it_o, axis=1)) >= 0.8, (iters[:, :4], it_o)
    Ho = np.array([np.asarray(r["H"]).ravel() for r in res_o])
    assert np.abs(H - Ho).max() <= 1e-4 * np.abs(Ho).max()


def test_emulated_sparse_align_edge_cases(emu, oracle):
    """ragged feature counts, features without a point, an empty frame (pose untouched, nothing tracked), the reference's
    default schedule (levels 4 -> 2 of a 5-level pyramid)"""
    seq = synth.make_sequence(4, 120)
    rng = np.random.default_rng(3)
    hp = (rng.uniform(size=(3, 120)) > 0.2).astype(np.uint8)
    b = make_batch(seq, [(0, 1), (1, 2), (2, 3)], 5, n_valid=[120, 37, 0], has_point=hp)
    T_h, n_tracked, iters, H, status = run_emulated(emu, b, 4, 2)
    T_o, res_o, _ = run_oracle(oracle, b, 4, 2)
    assert se3.log_norm(T_h, T_o).max() <= 1e-5
    assert np.array_equal(n_tracked, np.array([r["n_tracked"] for r in res_o]))
    assert n_tracked[2] == 0 and np.abs(T_h[2] - b.T_cur_w[2]).max() < 1e-12  # (the prior, up to the product T_cur_ref * T_ref)


@pytest.mark.parametrize("n_patches", [60, 190])
def test_emulated_wave_per_frame_kernel(emu_default, oracle, n_patches):
    """sparse_align_wave.hip -- a frame per wave, one and three patches per lane, no workgroup barrier, the solve in the same
    wave behind a transposing wave reduction -- through a test-only entry (svo_hip_sparse_align gives it batches of >= 1024
    frames only): the oracle's poses, and the workgroup kernel's, with the bounds of tests/test_sparse_align_gpu.py."""
    emu = emu_default
    seq = synth.make_sequence(7, n_patches, seed=23)
    b = make_batch(seq, [(i, i + 1) for i in range(6)], 4)
    b.n[3] = n_patches - 7
    b.has_point[5, ::3] = 0
    T_w, ntr_w, it_w, H_w, st_w = run_emulated(emu, b, 3, 0, entry="emu_sparse_align_wave")
    T_g, ntr_g, it_g, _, _ = run_emulated(emu, b, 3, 0)
    T_o, res_o, _ = run_oracle(oracle, b, 3, 0)
    assert np.all(np.isfinite(T_w))
    d = se3.log_norm(T_w, T_o)
    assert d.max() <= 1e-4 and np.median(d) <= 1e-5, d
    assert se3.log_norm(T_w, T_g).max() <= 1e-4
    it_o = np.array([r["iters"][:4] for r in res_o])
    same = np.all(it_o == it_w[:, :4], axis=1)
    assert (~same).sum() <= 1, (it_w[:, :4], it_o)
    assert np.array_equal(ntr_w[same], np.array([r["n_tracked"] for r in res_o])[same])
    assert np.array_equal(st_w, np.array([r["stop"] for r in res_o]))


@pytest.mark.parametrize("kind", ["radtan", "atan"])
def test_emulated_distorted_camera_models(emu, oracle, kind):
    """the DIST instantiation of sia_kernel (the camera model's world2cam in the loop, no window cache) -- and, for 60 patches
    per frame, of the wave-per-frame kernel: the oracle's poses through the vikit models' projection"""
    from helpers import camera_models
    cam = camera_models()[kind]
    seq = synth.make_sequence(4, 60, cam=cam, seed=9, margin=56, cell=40)
    b = make_batch(seq, [(i, i + 1) for i in range(3)], 5)
    T_o, res_o, _ = run_oracle(oracle, b, 4, 2)
    T_h, ntr, _, _, _ = run_emulated(emu, b, 4, 2)
    assert se3.log_norm(T_h, T_o).max() <= 1e-4 and np.median(se3.log_norm(T_h, T_o)) <= 1e-5
    assert np.array_equal(ntr, np.array([r["n_tracked"] for r in res_o]))
    T_w, ntr_w, _, _, _ = run_emulated(emu, b, 4, 2, entry="emu_sparse_align_wave")
    assert se3.log_norm(T_w, T_o).max() <= 1e-4 and se3.log_norm(T_w, T_h).max() <= 1e-4
    assert se3.log_norm(T_h, b.T_gt_w).max() < 5e-3   # both solved the problem (60 patches, levels 4 -> 2)


def test_emulated_border_features_outside_patches_and_iteration_caps(emu, oracle):
    """features in the 3..30 px band next to a border (invisible at coarse levels, joining at finer ones: the H rebuild when
    the set of patches inside the image changes); a prior so wrong that nothing projects into the image (H = 0, x = 0, the
    pose kept); n_iter = 0, 1, 2"""
    import torch
    seq = synth.make_sequence(4, 200, seed=3)
    rng = np.random.default_rng(3)
    for i in range(4):
        k = rng.choice(200, 40, replace=False)
        side = rng.integers(0, 4, size=40)
        off = rng.uniform(3.0, 30.0, size=40)
        u, v = seq.px[i, k, 0].numpy().copy(), seq.px[i, k, 1].numpy().copy()
        u[side == 0] = off[side == 0]
        u[side == 1] = 639.0 - off[side == 1]
        v[side == 2] = off[side == 2]
        v[side == 3] = 479.0 - off[side == 3]
        seq.px[i, k, 0] = torch.from_numpy(u)
        seq.px[i, k, 1] = torch.from_numpy(v)
    seq.f, seq.pos = synth.features_3d(seq.T_f_w, seq.cam, seq.px)
    b = make_batch(seq, [(0, 1), (1, 2), (2, 3)], 4)
    for lo in (0, 2):   # the full schedule (every feature joins eventually) and a stop at level 2 (some never do)
        T_o, res_o, _ = run_oracle(oracle, b, 3, lo)
        T_h, ntr, _, _, _ = run_emulated(emu, b, 3, lo)
        assert se3.log_norm(T_h, T_o).max() <= 1e-4
        assert np.array_equal(ntr, np.array([r["n_tracked"] for r in res_o]))
    assert np.all(ntr < 200) and np.all(ntr > 100), ntr
    # nothing inside the image
    b2 = make_batch(seq, [(0, 1), (2, 3)], 4)
    b2.T_cur_w = se3.mul(se3.exp(np.array([[50.0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0.0, 0]])), b2.T_cur_w)
    T_o, res_o, _ = run_oracle(oracle, b2, 3, 0)
    T_h, ntr, _, _, _ = run_emulated(emu, b2, 3, 0)
    assert res_o[0]["n_tracked"] == 0 and ntr[0] == 0
    assert se3.log_norm(T_h[:1], T_o[:1]).max() < 1e-12 and se3.log_norm(T_h[1:], T_o[1:]).max() <= 1e-4
    # iteration caps
    for n_iter in (0, 1, 2):
        b3 = make_batch(seq, [(0, 1)], 4)
        T_o, res_o, _ = run_oracle(oracle, b3, 3, 0, n_iter=n_iter)
        T_h, ntr, iters, _, _ = run_emulated(emu, b3, 3, 0, n_iter=n_iter)
        assert se3.log_norm(T_h, T_o).max() <= 1e-4
        assert np.array_equal(iters[:, :4], np.array([r["iters"][:4] for r in res_o]))
