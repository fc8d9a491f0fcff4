#!/usr/bin/env python
"""Run the reference's FrameHandlerMono twice on one synthetic sequence -- all-CPU reference
translation units vs. the same control plane with the drop-in HIP bodies -- and print the
per-frame SE(3) log-norm between the two trajectories, ATE, counters and timings.
(tests/dropin/_build/*.so must exist: `make -C tests/dropin`, needs /root/reference.)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
import numpy as np  # noqa: E402

import pypipeline as pp  # noqa: E402
from rpg_svo_amd import se3, synth  # noqa: E402


def horn_ate(P, Q):
    Pc, Qc = P - P.mean(0), Q - Q.mean(0)
    U, _, Vt = np.linalg.svd((Pc.T @ Qc).T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    t = Q.mean(0) - R @ P.mean(0)
    return float(np.sqrt((((R @ P.T).T + t - Q) ** 2).sum(1).mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--json", default=None)
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    cam = synth.Camera(752, 480, 315.5, 315.5, 376.0, 240.0)
    tex = synth.make_texture(seed=12345)
    T = synth.make_trajectory(args.frames, seed=args.seed, max_step=0.02, max_rot_deg=0.3)
    imgs = synth.render(tex, T, cam).numpy()
    out = {}
    for flv in ("ref", "hip"):
        t0 = time.time()
        out[flv] = pp.run_sequence(flv, cam, imgs, T)
        print(f"{flv}: {time.time() - t0:.3f} s for {len(imgs)} frames", file=sys.stderr)
    Tr = np.stack([r["T_f_w"] for r in out["ref"]])
    Th = np.stack([r["T_f_w"] for r in out["hip"]])
    d = se3.log_norm(Th, Tr)
    eg_r = se3.log_norm(Tr, T)
    eg_h = se3.log_norm(Th, T)
    pos_r, pos_h, pos_g = se3.inv(Tr)[:, 9:], se3.inv(Th)[:, 9:], se3.inv(T)[:, 9:]
    keys = ("n_obs", "is_keyframe", "n_kfs", "n_candidates", "n_seeds", "img_align_n_tracked", "repr_n_mps",
            "repr_n_new_references", "sfba_n_edges_final")
    same = {k: float(np.mean([a[k] == b[k] for a, b in zip(out["ref"], out["hip"])])) for k in keys}
    if args.verbose:
        for i, (a, b) in enumerate(zip(out["ref"], out["hip"])):
            print(i, "%.2e" % d[i], [(a[k], b[k]) for k in keys if a[k] != b[k]],
                  "ref %.2f ms hip %.2f ms" % (a["t_tot_time"] * 1e3, b["t_tot_time"] * 1e3))
    tm = lambda flv, k: float(np.median([r[k] for r in out[flv][1:]]) * 1e3)
    summary = {
        "frames": len(imgs), "se3_lognorm_hip_vs_ref_max": float(d.max()), "se3_lognorm_hip_vs_ref_median": float(np.median(d)),
        "ate_rmse_hip_vs_ref_m": horn_ate(pos_h, pos_r), "ate_rmse_ref_vs_gt_m": horn_ate(pos_r, pos_g),
        "ate_rmse_hip_vs_gt_m": horn_ate(pos_h, pos_g), "pose_err_vs_gt_median": {"ref": float(np.median(eg_r)), "hip": float(np.median(eg_h))},
        "identical_counter_fraction": same,
        "keyframes": {"ref": int(sum(r["is_keyframe"] for r in out["ref"])), "hip": int(sum(r["is_keyframe"] for r in out["hip"]))},
        "median_ms_per_frame": {flv: {k: tm(flv, "t_" + k) for k in ("tot_time", "sparse_img_align", "reproject", "pose_optimizer", "point_optimizer", "pyramid_creation")} for flv in ("ref", "hip")},
    }
    print(json.dumps(summary, indent=1))
    if args.json:
        json.dump(summary, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
