#!/usr/bin/env python
"""ROS-free replay driver (the role of svo_ros/src/benchmark_node.cpp:178-256):
feeds a synthetic sequence through the reference's FrameHandlerMono -- with the drop-in HIP
bodies (default) or all-reference on the CPU -- and writes what the reference's benchmark
writes: traj_estimate.txt, traj_groundtruth.txt, the per-frame trace CSV with the reference's
column names, and the ATE summary.

    python scripts/replay.py --out /tmp/run1 [--flavour hip|ref] [--frames 200] [--noise 2]
    python scripts/replay.py --out /tmp/run2 --dataset /data/sin2_tex2_h1_v8_d --cam 752,480,315.5,315.5,376,240
    python scripts/replay.py --write-dataset /tmp/synth_ds --frames 100      (synthetic data in that layout)

A dataset directory has the layout of the reference's Blender datasets (rpg_svo_amd/dataset.py):
trajectory.txt, img/<name>_0.png, depth/<name>_0.depth (needed for the first frame only).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
import numpy as np  # noqa: E402

from rpg_svo_amd import dataset, se3, synth, trace  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--dataset", help="replay this dataset directory instead of a synthetic sequence")
    ap.add_argument("--cam", default="752,480,315.5,315.5,376,240", help="width,height,fx,fy,cx,cy of --dataset")
    ap.add_argument("--write-dataset", help="write the synthetic sequence in the dataset layout and exit")
    ap.add_argument("--flavour", default="hip", choices=["hip", "ref"])
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--noise", type=float, default=0.0, help="image noise sigma (benchmark_node.cpp:166-176)")
    ap.add_argument("--fps", type=float, default=30.0)
    args = ap.parse_args()
    cam = synth.Camera(752, 480, 315.5, 315.5, 376.0, 240.0)
    if args.dataset:
        w, h, fx, fy, cx, cy = [float(x) for x in args.cam.split(",")]
        cam = synth.Camera(int(w), int(h), fx, fy, cx, cy)
        ts, names, T_gt = dataset.read_trajectory_file(args.dataset)
        if args.frames < len(names):
            ts, names, T_gt = ts[:args.frames], names[:args.frames], T_gt[:args.frames]
        args.frames = len(names)
        imgs = np.stack([dataset.read_image(args.dataset, n) for n in names])
        range0 = dataset.load_blender_depthmap(os.path.join(args.dataset, "depth", names[0] + "_0.depth"), cam)
    else:
        T_gt = synth.make_trajectory(args.frames, seed=args.seed, max_step=0.02, max_rot_deg=0.3)
        imgs = synth.render(synth.make_texture(seed=12345), T_gt, cam).numpy()
        ts = np.arange(args.frames) / args.fps
        range0 = None
    if args.noise > 0:
        rng = np.random.default_rng(args.seed)
        imgs = np.clip(imgs.astype(np.float32) + rng.normal(0, args.noise, imgs.shape), 0, 255).round().astype(np.uint8)
    if args.write_dataset:
        sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
        R0 = T_gt[0, :9].reshape(3, 3)
        c0 = -R0.T @ T_gt[0, 9:]
        u, v = np.meshgrid(np.arange(cam.width, dtype=np.float64), np.arange(cam.height, dtype=np.float64))
        d = np.stack([(u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy, np.ones_like(u)], -1)
        z0 = (-c0[2] / (d @ R0)[..., 2]).astype(np.float32)  # z-depth of the plane z = 0 in frame 0
        dataset.write_dataset(args.write_dataset, imgs, T_gt, cam, ts, z_depth={0: z0})
        print(json.dumps({"written": args.write_dataset, "frames": int(args.frames)}))
        return
    if not args.out:
        raise SystemExit("--out is required for a replay")
    import pypipeline as pp
    if not pp.available(args.flavour):
        raise SystemExit(f"{pp.lib_path(args.flavour)} missing: build with `make -C tests/dropin` (needs the reference checkout)")
    os.makedirs(args.out, exist_ok=True)
    if range0 is None:
        range0 = pp.range_map(cam, T_gt[0])
    p = pp.Pipeline(args.flavour, cam)
    rows, T_est, ok = [], [], []
    try:
        n0, r = p.set_first_frame(imgs[0], ts[0], T_gt[0], range0)
        rows.append(r); T_est.append(r["T_f_w"]); ok.append(True)
        for i in range(1, args.frames):
            r = p.add_image(imgs[i], ts[i])
            rows.append(r); T_est.append(r["T_f_w"])
            ok.append(r["stage"] == pp.STAGE_DEFAULT_FRAME)
            if not ok[-1]:
                print(f"SVO failed at frame {i} before the entire dataset could be processed", file=sys.stderr)
                break
    finally:
        p.close()
    T_est = np.stack(T_est)
    n = len(T_est)
    trace.write_trajectory(os.path.join(args.out, "traj_estimate.txt"), ts[:n], T_est)
    trace.write_trajectory(os.path.join(args.out, "traj_groundtruth.txt"), ts[:n], T_gt[:n])
    csv_rows = []
    for r in rows:
        d = {k: r.get(k, 0.0) for k in trace.LOGS}
        d.update({k: r.get("t_" + k, 0.0) for k in trace.TIMERS})
        csv_rows.append(d)
    trace.write_trace_csv(os.path.join(args.out, "svo.csv"), csv_rows)
    stats = trace.ate(se3.inv(T_est)[:, 9:], se3.inv(T_gt[:n])[:, 9:])
    stats.update({"flavour": args.flavour, "frames": n, "first_frame_features": n0,
                  "keyframes": int(sum(r["is_keyframe"] for r in rows)),
                  "median_ms_per_frame": float(np.median([r["t_tot_time"] for r in rows[1:]]) * 1e3)})
    json.dump(stats, open(os.path.join(args.out, "ate.json"), "w"), indent=1)
    print(json.dumps(stats))


if __name__ == "__main__":
    main()
