"""On-disk formats either side of the path, as the reference writes/reads them:

  traj_estimate.txt   "timestamp tx ty tz qx qy qz qw" per frame, pose T_w_f, timestamp with 15
                      and the rest with 6 fixed decimals (svo_ros/src/benchmark_node.cpp:91-101)
  <trace_name>.csv    one row per frame; header = timer names then log names, each group in
                      alphabetical order, values with 15 fixed decimals -- the file
                      vk::PerformanceMonitor writes for FrameHandlerBase
                      (svo/src/frame_handler_base.cpp:46-74; cf. svo/test/benchmark.csv)
  ATE                 Horn-aligned absolute trajectory error as
                      svo_analysis/src/svo_analysis/tum_benchmark_tools/evaluate_ate.py:47-80

so that svo_analysis' scripts run unchanged on a replay of this implementation.
"""
from __future__ import annotations

import numpy as np

from . import se3

TIMERS = ("local_ba", "point_optimizer", "pose_optimizer", "pyramid_creation", "reproject", "sparse_img_align", "tot_time")
LOGS = ("dropout", "img_align_n_tracked", "loba_err_fin", "loba_err_init", "loba_n_erredges_fin", "loba_n_erredges_init",
        "n_candidates", "repr_n_mps", "repr_n_new_references", "sfba_error_final", "sfba_error_init",
        "sfba_n_edges_final", "sfba_thresh")
TRACE_COLUMNS = TIMERS + LOGS


def quat_from_R(R: np.ndarray) -> np.ndarray:
    """Unit quaternion (x, y, z, w) of a rotation matrix, w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    q /= np.linalg.norm(q)
    return q if q[3] >= 0 else -q


def R_from_quat(q: np.ndarray) -> np.ndarray:
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def write_trajectory(path: str, timestamps, T_f_w: np.ndarray) -> None:
    """T_f_w [n,12] (world -> frame); the file holds T_w_f like the reference's tracePose."""
    T_w_f = se3.inv(np.asarray(T_f_w, dtype=np.float64))
    with open(path, "w") as fh:
        for ts, T in zip(timestamps, T_w_f):
            q = quat_from_R(T[:9].reshape(3, 3))
            p = T[9:]
            fh.write("%.15f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n" % (ts, p[0], p[1], p[2], q[0], q[1], q[2], q[3]))


def read_trajectory(path: str):
    """-> (timestamps [n], T_w_f [n,12])"""
    ts, Ts = [], []
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        v = [float(x) for x in line.replace(",", " ").split()]
        ts.append(v[0])
        Ts.append(np.concatenate([R_from_quat(np.array(v[4:8])).reshape(9), v[1:4]]))
    return np.array(ts), np.array(Ts)


def write_trace_csv(path: str, rows: list[dict]) -> None:
    """rows: per frame a dict with (a subset of) TRACE_COLUMNS; timers in seconds."""
    with open(path, "w") as fh:
        fh.write(",".join(TRACE_COLUMNS) + "\n")
        for r in rows:
            fh.write(",".join("%.15f" % float(r.get(c, 0.0)) for c in TRACE_COLUMNS) + "\n")


def read_trace_csv(path: str) -> dict[str, np.ndarray]:
    lines = open(path).read().strip().splitlines()
    names = lines[0].split(",")
    data = np.array([[float(x) for x in l.split(",")] for l in lines[1:]]).reshape(-1, len(names))
    return {n: data[:, i] for i, n in enumerate(names)}


def align_horn(model: np.ndarray, data: np.ndarray):
    """Closed-form rigid alignment (Horn) of model [n,3] onto data [n,3]: (R, t, per-point error)."""
    mz, dz = model - model.mean(0), data - data.mean(0)
    W = mz.T @ dz
    U, _, Vt = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    t = data.mean(0) - R @ model.mean(0)
    err = np.sqrt((((R @ model.T).T + t - data) ** 2).sum(1))
    return R, t, err


def ate(pos_est: np.ndarray, pos_ref: np.ndarray) -> dict:
    """The statistics evaluate_ate.py prints (metres)."""
    _, _, e = align_horn(np.asarray(pos_est, float), np.asarray(pos_ref, float))
    return {"compared_pose_pairs": int(len(e)), "rmse": float(np.sqrt((e * e).mean())), "mean": float(e.mean()),
            "median": float(np.median(e)), "std": float(e.std()), "min": float(e.min()), "max": float(e.max())}


def associate(ts_a, ts_b, max_difference: float = 0.02):
    """Greedy nearest-timestamp association (tum_benchmark_tools/associate.py)."""
    cand = sorted((abs(a - b), i, j) for i, a in enumerate(ts_a) for j, b in enumerate(ts_b) if abs(a - b) < max_difference)
    used_a, used_b, out = set(), set(), []
    for _, i, j in cand:
        if i not in used_a and j not in used_b:
            used_a.add(i); used_b.add(j); out.append((i, j))
    return sorted(out)
