"""Shared builders for parity tests: a batch of (ref, cur) alignment problems in
both representations (host arrays for the oracle, device tensors for the HIP path)."""
from __future__ import annotations

import os

import numpy as np
import torch

from rpg_svo_amd import se3, synth
from rpg_svo_amd.sparse_img_align import marshal_problem


class Batch:
    pass


def make_batch(seq: synth.Sequence, pairs, n_levels: int, n_valid=None, prior="ref", has_point=None,
               prior_noise=None, seed=0):
    """pairs: list of (ref_idx, cur_idx).  prior: 'ref' = pose of the reference frame
    (the pipeline's constant-position prior, frame_handler_mono.cpp:132) or 'gt'."""
    b = Batch()
    B = len(pairs)
    N = seq.px.shape[1]
    b.B, b.N, b.n_levels = B, N, n_levels
    b.cam = seq.cam
    b.ref_slot = np.array([p[0] for p in pairs], dtype=np.int32)
    b.cur_slot = np.array([p[1] for p in pairs], dtype=np.int32)
    b.images = seq.images.cpu().numpy()
    b.T_ref_w = seq.T_f_w[b.ref_slot].copy()
    b.T_gt_w = seq.T_f_w[b.cur_slot].copy()
    b.T_cur_w = b.T_ref_w.copy() if prior == "ref" else b.T_gt_w.copy()
    if prior_noise is not None:
        rng = np.random.default_rng(seed)
        b.T_cur_w = se3.mul(se3.exp(rng.normal(size=(B, 6)) * prior_noise), b.T_cur_w)
    b.px = seq.px[b.ref_slot].cpu().numpy().copy()
    b.f = seq.f[b.ref_slot].cpu().numpy().copy()
    b.pos = seq.pos[b.ref_slot].cpu().numpy().copy()
    b.n = np.full(B, N, dtype=np.int32) if n_valid is None else np.asarray(n_valid, dtype=np.int32)
    b.has_point = np.ones((B, N), dtype=np.uint8) if has_point is None else np.asarray(has_point, dtype=np.uint8)
    return b


def run_oracle(oracle, b: Batch, max_level, min_level, n_iter=30, halfsample=None, n_threads=4, which="orc"):
    """which: "orc" = the C restatement, "ref" = the reference's own SparseImgAlign (oracle/_ref)."""
    mode = oracle.HALFSAMPLE_AUTO if halfsample is None else halfsample
    pyrs = [oracle.create_img_pyramid(im, b.n_levels, mode) for im in b.images]
    T, res = oracle.sparse_img_align_batch(pyrs, b.ref_slot, b.cur_slot, b.cam, b.T_ref_w, b.T_cur_w, b.n,
                                           b.px, b.f, b.has_point, b.pos, max_level, min_level, n_iter,
                                           n_threads=n_threads, which=which)
    return T, res, pyrs


def tile_batch(b: Batch, times: int) -> Batch:
    """The same problems `times` over (large-batch paths: svo_hip_sparse_align switches kernel with B)."""
    import copy
    t = copy.copy(b)
    t.B = b.B * times
    for k in ("ref_slot", "cur_slot", "T_ref_w", "T_gt_w", "T_cur_w", "px", "f", "pos", "n", "has_point"):
        a = getattr(b, k)
        setattr(t, k, np.concatenate([a] * times, axis=0))
    return t


def run_hip(b: Batch, max_level, min_level, n_iter=30, device="cuda:0", halfsample=None, kernel="auto"):
    from rpg_svo_amd import capi
    from rpg_svo_amd.pyramid import PyramidStore
    from rpg_svo_amd.sparse_img_align import SparseImgAlign
    mode = capi.HALFSAMPLE_AUTO if halfsample is None else halfsample
    store = PyramidStore(b.cam.width, b.cam.height, b.n_levels, len(b.images), device=device, halfsample=mode)
    store.load_images(torch.from_numpy(b.images).to(device))
    T_cr, xyz = marshal_problem(b.T_ref_w, b.T_cur_w, b.f, b.pos)
    dev = torch.device(device)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    sia = SparseImgAlign(max_level, min_level, n_iter)
    sia.kernel = kernel
    out = sia.run(store, b.cam, t(b.ref_slot, torch.int32), t(b.cur_slot, torch.int32), t(b.n, torch.int32),
                  t(b.px, torch.float64), t(xyz, torch.float64), t(T_cr, torch.float64),
                  valid=t(b.has_point, torch.uint8))
    torch.cuda.synchronize()
    T_cr_out = out.T_cur_from_ref.cpu().numpy()
    # cur_frame_->T_f_w_ = T_cur_from_ref * ref_frame_->T_f_w_ (sparse_img_align.cpp:70)
    T_cur_w = se3.mul(T_cr_out, b.T_ref_w)
    return T_cur_w, out, store


# ---- scenes for the steps after sparse alignment ---------------------------------------
def scene_store(scene, n_levels=5, device="cuda:0", T_override=None):
    """PyramidStore + FrameTable of a synth.TrackScene (slot i = frame i)."""
    from rpg_svo_amd.pyramid import PyramidStore
    from rpg_svo_amd.tracking import FrameTable
    dev = torch.device(device)
    n = scene.images.shape[0]
    store = PyramidStore(scene.cam.width, scene.cam.height, n_levels, n, device=device)
    store.load_images(scene.images.to(dev))
    T = scene.T_f_w if T_override is None else T_override
    frames = FrameTable(torch.arange(n, dtype=torch.int32, device=dev),
                        torch.as_tensor(np.ascontiguousarray(T), dtype=torch.float64, device=dev))
    return store, frames


def obs_csr(obs_lists, device="cuda:0"):
    """Point::obs_ lists [(frame, px, f, level, type, grad)] -> (obs_ptr, FeatureSet)."""
    from rpg_svo_amd.tracking import FeatureSet
    dev = torch.device(device)
    ptr = np.zeros(len(obs_lists) + 1, dtype=np.int32)
    flat = []
    for i, o in enumerate(obs_lists):
        ptr[i + 1] = ptr[i] + len(o)
        flat.extend(o)
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    fs = FeatureSet(frame=t([o[0] for o in flat], torch.int32), level=t([o[3] for o in flat], torch.int32),
                    px=t([o[1] for o in flat], torch.float64), f=t([o[2] for o in flat], torch.float64),
                    type=t([o[4] for o in flat], torch.uint8), grad=t([o[5] for o in flat], torch.float64))
    return t(ptr, torch.int32), fs


# ---- the camera models of the reference's launch files -------------------------------------
def camera_models():
    """name -> synth.Camera: the undistorted VGA pinhole of the benchmark, and the two cameras the
    reference ships (svo_ros/param/camera_pinhole.yaml: radial-tangential, camera_atan.yaml: ATAN)."""
    return {
        "pinhole": synth.Camera.vga(),
        "radtan": synth.Camera.radtan(752, 480, 414.536145, 414.284429, 348.804988, 240.076451,
                                      -0.283076, 0.066674, 0.000896, 0.000778),
        "atan": synth.Camera.atan(752, 480, 0.509326, 0.796651, 0.45905, 0.510056, 0.9320),
    }


CAMERA_KINDS = ("pinhole", "radtan", "atan")


# ---- random maps for Reprojector::reprojectMap (row N2) ------------------------------------------------------------
def random_map(cam, n_kfs=12, n_points=900, n_candidates=700, seed=0, n_overlap=10, cell_size=30, dead_frac=0.05):
    """A map in the plain-array form of svo_hip_map: keyframes on a random walk above the plane z = 0 looking down,
    map points (GOOD / UNKNOWN) observed in 1..n keyframes at random positions of their fts_ lists, candidates with
    one observation that is in no keyframe's list, some dead entries; the current frame is the last entry of the frame
    table; `n_overlap` keyframes ranked by distance, the rest not overlapping; a random visiting order of the cells."""
    rng = np.random.default_rng(seed)
    n_frames = n_kfs + 1
    # keyframe centres up to 3 m from the current one at ~2 m height: viewing directions of a point differ by up to
    # ~70 degrees between frames, so Point::getCloseViewObs (cos < 0.5) rejects some candidates
    R0 = np.diag([1.0, -1.0, -1.0])
    T = np.zeros((n_frames, 12))
    for i in range(n_frames):
        rv = rng.normal(0, 0.05, 3)
        R = se3.split(se3.exp(np.concatenate([np.zeros(3), rv])))[0] @ R0
        ctr = np.array([rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(1.6, 2.4)]) if i < n_kfs else np.array([0.1, -0.2, 2.0])
        T[i] = se3.join(R, -R @ ctr)
    c = se3.inv(T)[:, 9:]
    cur = n_frames - 1
    dist = np.linalg.norm(c[:n_kfs] - c[cur], axis=1)
    kf_rank = np.full(n_frames, -1, dtype=np.int32)
    for r, f in enumerate(np.argsort(dist, kind="stable")[:n_overlap]):
        kf_rank[f] = r
    P = n_points + n_candidates
    # points on and around the plane, spread over more than the field of view (some project outside)
    span = 1.3 * 2.0 * max(cam.width / cam.fx, cam.height / cam.fy) / 2
    pos = np.stack([rng.uniform(-span, span, P), rng.uniform(-span, span, P), rng.normal(0, 0.05, P)], -1)
    pos[:, :2] += c[cur, :2]
    type_ = np.zeros(P, dtype=np.int32)
    type_[:n_points] = rng.choice([2, 3], size=n_points)
    type_[n_points:] = 1
    type_[rng.uniform(size=P) < dead_frac] = 0
    order = np.zeros(P, dtype=np.int32)
    perm = rng.permutation(n_candidates)           # list order of the candidates is NOT entry order
    order[n_points:] = perm
    obs_begin, obs_count = np.zeros(P, dtype=np.int32), np.zeros(P, dtype=np.int32)
    obs_frame, obs_order = [], []
    next_ord = np.zeros(n_kfs, dtype=np.int64)
    slots = [rng.permutation(4000)[:P] for _ in range(n_kfs)]   # distinct fts_ positions per keyframe
    for p in range(P):
        obs_begin[p] = len(obs_frame)
        if p < n_points:
            k = rng.integers(1, min(6, n_kfs) + 1)
            for f in rng.choice(n_kfs, size=k, replace=False):
                obs_frame.append(f)
                obs_order.append(int(slots[f][next_ord[f]]))
                next_ord[f] += 1
            if rng.uniform() < 0.03:   # an observation whose Feature is in no keyframe's list
                obs_frame.append(int(rng.integers(n_kfs))); obs_order.append(-1)
        else:
            obs_frame.append(int(rng.integers(n_kfs))); obs_order.append(-1)
        obs_count[p] = len(obs_frame) - obs_begin[p]
    O = len(obs_frame)
    n_cols, n_rows = -(-cam.width // cell_size), -(-cam.height // cell_size)
    cell_order = rng.permutation(n_cols * n_rows)
    cell_rank = np.empty(n_cols * n_rows, dtype=np.int32)
    cell_rank[cell_order] = np.arange(n_cols * n_rows, dtype=np.int32)
    return dict(T=T, cur=cur, kf_rank=kf_rank, pos=pos, type=type_, order=order, obs_begin=obs_begin, obs_count=obs_count,
                obs_frame=np.array(obs_frame, dtype=np.int32), obs_order=np.array(obs_order, dtype=np.int32),
                obs_level=rng.integers(0, 3, O).astype(np.int32), obs_type=(rng.uniform(size=O) < 0.2).astype(np.uint8),
                obs_px=rng.uniform(20, 400, (O, 2)), obs_f=rng.normal(size=(O, 3)), obs_grad=rng.normal(size=(O, 2)),
                cell_size=cell_size, n_cols=n_cols, n_rows=n_rows, cell_order=cell_order, cell_rank=cell_rank)


def oracle_reproject_map(mp, cam, first_cell=0, max_cells=1 << 30):
    from oracle import pytrack
    return pytrack.reproject_map(cam, mp["T"], mp["cur"], mp["kf_rank"], mp["pos"], mp["type"], mp["order"], mp["obs_begin"],
                                 mp["obs_count"], mp["obs_frame"], mp["obs_order"], mp["cell_size"], mp["n_cols"],
                                 mp["n_cols"] * mp["n_rows"], mp["cell_rank"], first_cell, max_cells)


# SVO_TEST_FUZZ=<k> (default 0: the suites as committed) moves the track scene's trajectory, features and depth errors and
# every random draw of the tracking suites (tests/test_tracking_gpu.py, test_track_emulated.py, test_optimizers_emulated.py)
# to other seeds: scripts/fuzz_tracking.sh runs them over a range of k -- the bit-exact asserts on scenes nobody has looked at.
FUZZ = int(os.environ.get("SVO_TEST_FUZZ", "0"))


def fuzz_rng(k):
    return np.random.default_rng(k + 1000 * FUZZ)
