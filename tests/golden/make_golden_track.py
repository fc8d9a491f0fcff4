#!/usr/bin/env python
"""Generates tests/golden/track_qvga.npz: a small seeded scene (3 keyframes + a current frame,
320x240) with the outputs of the steps after sparse alignment -- Matcher::findMatchDirect per map
point, pose_optimizer::optimizeGaussNewton, FastDetector::detect -- as computed by THE REFERENCE'S
OWN translation units (oracle/_ref/libsvo_ref.so: svo/src/matcher.cpp, feature_alignment.cpp,
point.cpp, pose_optimizer.cpp, feature_detection.cpp compiled in place against oracle/shim).  The C
restatement is run on the same inputs and must agree bit for bit before anything is written.
Run from the repo root in the build container:  python tests/golden/make_golden_track.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import pytrack  # noqa: E402
from rpg_svo_amd import se3, synth  # noqa: E402

N_LEVELS = 4


def main():
    if not pytrack.ref_available() and not pytrack.build_ref():
        raise SystemExit("oracle/_ref is needed (reference checkout absent)")
    ref, orc = pytrack.Track("ref"), pytrack.Track("orc")
    cam = synth.Camera(320, 240, 200.0, 200.0, 160.0, 120.0)
    scene = synth.make_track_scene(n_kf=3, n_feat=40, cam=cam, seed=99)
    imgs = scene.images.cpu().numpy()
    T = scene.T_f_w.copy()
    T[scene.cur] = scene.T_cur_prior
    out = {}
    res = {}
    for name, trk in (("ref", ref), ("orc", orc)):
        pyrs = [trk.create_img_pyramid(im, N_LEVELS) for im in imgs]
        frames = pytrack.make_frames(pyrs, T)
        opt = pytrack.matcher_options(n_pyr_levels=N_LEVELS)
        ok, px, ro, sl, pw = [], [], [], [], []
        for i in range(len(scene.obs)):
            obs = [pytrack.make_feature(*o) for o in scene.obs[i]]
            o_ok, o_px, r = trk.find_match_direct(frames, cam, scene.cur, scene.pt_pos[i], obs, scene.px_init[i], opt)
            ok.append(o_ok); px.append(o_px); ro.append(r["ref_obs"]); sl.append(r["search_level"]); pw.append(r["patch_with_border"])
        rng = np.random.default_rng(5)
        P = len(scene.pt_pos)
        f = synth._bearing(cam, scene.px_true + rng.normal(size=(P, 2)) * 0.3)
        level = rng.integers(0, 3, size=P).astype(np.int32)
        pos = scene.pt_pos.copy()
        pos[::11] += rng.normal(size=pos[::11].shape) * 0.2
        hp = (rng.uniform(size=P) > 0.15).astype(np.uint8)
        po = trk.pose_optimize(cam, scene.T_cur_prior, f, level, hp, pos, 2.0, 10)
        res[name] = dict(ok=np.array(ok, dtype=np.uint8), px=np.array(px), ref_obs=np.array(ro, dtype=np.int32),
                         search_level=np.array(sl, dtype=np.int32), patch=np.array(pw, dtype=np.uint8),
                         po_T=po["T_f_w"], po_Cov=po["Cov"], po_hp=po["has_point"],
                         po_stats=np.array([po["estimated_scale"], po["error_init"], po["error_final"], po["num_obs"]]))
        out.update(po_f=f, po_level=level, po_pos=pos, po_hp_in=hp)
    for k in res["ref"]:
        assert np.array_equal(res["ref"][k], res["orc"][k]), f"restatement differs from the reference in {k}"
    # FastDetector::detect on the current frame (reference TU + restated FAST library)
    pyr = orc.create_img_pyramid(imgs[scene.cur], N_LEVELS)
    cell, lv = 20, 3
    cols, rows = -(-cam.width // cell), -(-cam.height // cell)
    xy, lvl, sc, n = pytrack.fast_detect_grid(pyr, lv, cell, cols, rows, None, 20, 20.0)
    px_ref, lvl_ref = pytrack.ref_fast_detect(pyr, cam, lv, cell, None, 20.0)
    sel = sc > 20.0
    assert np.array_equal(xy[sel].astype(np.float64), px_ref) and np.array_equal(lvl[sel], lvl_ref)
    ptr = np.zeros(len(scene.obs) + 1, dtype=np.int32)
    flat = []
    for i, o in enumerate(scene.obs):
        ptr[i + 1] = ptr[i] + len(o)
        flat.extend(o)
    np.savez_compressed(
        os.path.join(os.path.dirname(os.path.abspath(__file__)), "track_qvga.npz"),
        cam=np.array([cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy]), n_levels=N_LEVELS, images=imgs, T_f_w=T, cur=scene.cur,
        pt_pos=scene.pt_pos, px_init=scene.px_init, obs_ptr=ptr, obs_frame=np.array([o[0] for o in flat], dtype=np.int32),
        obs_px=np.array([o[1] for o in flat]), obs_f=np.array([o[2] for o in flat]), obs_level=np.array([o[3] for o in flat], dtype=np.int32),
        obs_type=np.array([o[4] for o in flat], dtype=np.uint8), obs_grad=np.array([o[5] for o in flat]),
        fast_cell=cell, fast_levels=lv, fast_xy=xy, fast_level=lvl, fast_score=sc,
        **{("m_" + k if not k.startswith("po_") else k): v for k, v in res["ref"].items()}, **out)
    print("wrote track_qvga.npz:", int(res["ref"]["ok"].sum()), "of", len(scene.obs), "matches;", int(n), "corners; pose-opt obs",
          int(res["ref"]["po_stats"][3]))


if __name__ == "__main__":
    main()
