#!/usr/bin/env python
"""The drop-in against the all-CPU reference over SEVERAL synthetic trajectories (one sequence is an anecdote).

For each seed: the reference's FrameHandlerMono with its own translation units and with the drop-in HIP bodies on the same
images; the first frame at which ANY discrete decision differs, the first frame at which a TRACKER decision differs, the
trajectory agreement before those frames, and the accuracy of both runs against the ground truth over the whole sequence.
For the first sequence that has one, the depth filter's seed list of both runs is dumped at the frame before the first difference: which
seed sits on the convergence threshold (sqrt(sigma2) against z_range / 200, depth_filter.cpp:261) is then a recorded fact.
GPU box:  python scripts/dropin_many.py [frames=400] [seeds=5,11,23,37,41,59] > profiles/r05_dropin_many.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
import numpy as np  # noqa: E402

import pypipeline as pp  # noqa: E402
from rpg_svo_amd import se3, synth  # noqa: E402
from scripts.dropin_compare import horn_ate  # noqa: E402

DEC = ("is_keyframe", "n_obs", "repr_n_mps", "repr_n_new_references", "n_kfs", "stage", "img_align_n_tracked", "n_candidates", "n_seeds")
TRK = ("is_keyframe", "n_obs", "repr_n_mps", "repr_n_new_references", "n_kfs", "stage", "img_align_n_tracked")


def seeds_at(flavour, cam, imgs, T, frame):
    """the seed list after `frame` frames have been fed"""
    p = pp.Pipeline(flavour, cam)
    try:
        p.set_first_frame(imgs[0], 0.0, T[0], pp.range_map(cam, T[0]))
        for i in range(1, frame + 1):
            p.add_image(imgs[i], float(i))
        return p.seeds()
    finally:
        p.close()


def main():
    kv = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
    n_frames = int(kv.get("frames", 400))
    seeds = [int(s) for s in kv.get("seeds", "5,11,23,37,41,59").split(",")]
    flavour = kv.get("flavour", "hip")  # (hipmock: the drop-in's host code on the mock device, CPU only)
    cam = synth.Camera(752, 480, 315.5, 315.5, 376.0, 240.0)
    tex = synth.make_texture(seed=12345)
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)  # the reference logs every frame to stderr
    runs = []
    try:
        import torch
        dev = "cuda" if torch.cuda.is_available() else "cpu"
        for k, sd in enumerate(seeds):
            T = synth.make_trajectory(n_frames, seed=sd, max_step=0.02, max_rot_deg=0.3)
            imgs = synth.render(tex, T, cam, device=dev).cpu().numpy()
            ref = pp.run_sequence("ref", cam, imgs, T)
            hip = pp.run_sequence(flavour, cam, imgs, T)
            Tr, Th = np.stack([r["T_f_w"] for r in ref]), np.stack([r["T_f_w"] for r in hip])
            d = se3.log_norm(Th, Tr)
            first = lambda keys: next((i for i, (a, b) in enumerate(zip(ref, hip)) if any(a[q] != b[q] for q in keys)), None)
            f_any, f_trk = first(DEC), first(TRK)
            pos = lambda TT: se3.inv(TT)[:, 9:]
            run = {"trajectory_seed": sd, "frames": n_frames, "keyframes": {"ref": int(sum(r["is_keyframe"] for r in ref)), "hip": int(sum(r["is_keyframe"] for r in hip))},
                   "first_frame_with_a_different_decision": f_any, "first_frame_with_a_different_tracking_decision": f_trk,
                   "what_differs_first": None if f_any is None else {q: [ref[f_any][q], hip[f_any][q]] for q in DEC if ref[f_any][q] != hip[f_any][q]},
                   "se3_lognorm_max_before_the_first_difference": float(d[:f_any if f_any is not None else n_frames].max()),
                   "se3_lognorm_max_before_the_first_tracking_difference": float(d[:f_trk if f_trk is not None else n_frames].max()),
                   "se3_lognorm_max_whole_sequence": float(d.max()),
                   "ate_rmse_vs_ground_truth_m": {"cpu_reference": horn_ate(pos(Tr), pos(T)), "hip_dropin": horn_ate(pos(Th), pos(T))},
                   "ate_rmse_hip_vs_cpu_m": horn_ate(pos(Th), pos(Tr))}
            if f_any is not None and f_any >= 2 and not any("seed_lists_before_the_first_difference" in r for r in runs):
                # the seed lists of both runs after frame f_any - 1: identical decisions so far, so the lists line up
                sr, sh = seeds_at("ref", cam, imgs, T, f_any - 1), seeds_at(flavour, cam, imgs, T, f_any - 1)
                thr = lambda s: np.sqrt(np.maximum(s[:, 8], 0)) * 200.0 / s[:, 7]  # < 1: converged (depth_filter.cpp:261)
                rec = {"frame": f_any - 1, "n_seeds": [len(sr), len(sh)]}
                if len(sr) == len(sh) and len(sr):
                    m = np.abs(thr(sr) - 1.0)
                    order = np.argsort(m)[:5]
                    rec["seeds_nearest_the_convergence_threshold"] = [
                        {"batch_id": int(sr[i, 0]), "px": [float(sr[i, 2]), float(sr[i, 3])],
                         "sqrt_sigma2_times_200_over_z_range": {"cpu_reference": float(thr(sr)[i]), "hip_dropin": float(thr(sh)[i])},
                         "sigma2": {"cpu_reference": float(sr[i, 8]), "hip_dropin": float(sh[i, 8])},
                         "mu": {"cpu_reference": float(sr[i, 6]), "hip_dropin": float(sh[i, 6])},
                         "z_range": float(sr[i, 7])} for i in order]
                    rec["max_relative_sigma2_difference_over_all_seeds"] = float(np.max(np.abs(sr[:, 8] - sh[:, 8]) / np.maximum(np.abs(sr[:, 8]), 1e-30)))
                run["seed_lists_before_the_first_difference"] = rec
            runs.append(run)
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
    arr = lambda key, sub=None: np.array([(r[key][sub] if sub else r[key]) for r in runs if (r[key][sub] if sub else r[key]) is not None], dtype=float)
    ms = lambda a: {"mean": float(a.mean()), "std": float(a.std()), "min": float(a.min()), "max": float(a.max()), "n": int(a.size)} if a.size else None
    out = {"sequences": len(runs), "frames_each": n_frames,
           "first_frame_with_a_different_decision": ms(arr("first_frame_with_a_different_decision")),
           "first_frame_with_a_different_tracking_decision": ms(arr("first_frame_with_a_different_tracking_decision")),
           "se3_lognorm_max_before_the_first_difference": ms(arr("se3_lognorm_max_before_the_first_difference")),
           "ate_rmse_vs_ground_truth_m": {"cpu_reference": ms(arr("ate_rmse_vs_ground_truth_m", "cpu_reference")),
                                          "hip_dropin": ms(arr("ate_rmse_vs_ground_truth_m", "hip_dropin"))},
           "ate_rmse_hip_vs_cpu_m": ms(arr("ate_rmse_hip_vs_cpu_m")), "runs": runs}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
