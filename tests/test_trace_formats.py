"""N3: trajectory / trace files in the reference's formats and the ATE evaluation (CPU only)."""
import os
import sys

import numpy as np

from rpg_svo_amd import se3, synth, trace


def test_trajectory_roundtrip(tmp_path):
    T = synth.make_trajectory(25, seed=2)
    ts = np.arange(25) / 30.0
    p = str(tmp_path / "traj_estimate.txt")
    trace.write_trajectory(p, ts, T)
    first = open(p).readline().split()
    assert len(first) == 8 and len(first[0].split(".")[1]) == 15 and all(len(x.split(".")[1]) == 6 for x in first[1:])
    ts2, T_w_f = trace.read_trajectory(p)
    assert np.allclose(ts2, ts)
    d = se3.log_norm(se3.inv(T_w_f), T)
    assert d.max() < 5e-6  # six decimals


def test_quaternion_conversions():
    rng = np.random.default_rng(0)
    for _ in range(50):
        R = se3.split(se3.exp(np.concatenate([np.zeros(3), rng.normal(size=3) * 2.0])))[0]
        q = trace.quat_from_R(R)
        assert abs(np.linalg.norm(q) - 1) < 1e-12 and q[3] >= 0
        assert np.allclose(trace.R_from_quat(q), R, atol=1e-12)


def test_trace_csv_matches_reference_header(tmp_path):
    ref_csv = "/root/reference/svo/test/benchmark.csv"
    p = str(tmp_path / "svo.csv")
    trace.write_trace_csv(p, [{"tot_time": 0.001, "repr_n_mps": 130, "dropout": 0}, {"sfba_thresh": 2.0}])
    if os.path.exists(ref_csv):  # same columns, same order, as the file the reference ships
        assert open(p).readline().strip() == open(ref_csv).readline().strip()
    d = trace.read_trace_csv(p)
    assert list(d) == list(trace.TRACE_COLUMNS) and d["repr_n_mps"][0] == 130 and d["sfba_thresh"][1] == 2.0
    assert open(p).read().splitlines()[1].split(",")[6] == "0.001000000000000"


def test_ate_is_invariant_to_rigid_motion_and_detects_error():
    rng = np.random.default_rng(1)
    P = rng.normal(size=(200, 3))
    R = se3.split(se3.exp(np.array([0, 0, 0, 0.3, -0.2, 0.9])))[0]
    Q = P @ R.T + np.array([1.0, -2.0, 0.5])
    s = trace.ate(P, Q)
    assert s["rmse"] < 1e-12 and s["compared_pose_pairs"] == 200
    Qn = Q + rng.normal(size=Q.shape) * 0.01
    s = trace.ate(P, Qn)
    assert 0.012 < s["rmse"] < 0.022 and s["min"] <= s["median"] <= s["max"]


def test_associate():
    a = [0.0, 0.1, 0.2, 0.3]
    b = [0.101, 0.199, 0.35, 0.0005]
    assert trace.associate(a, b) == [(0, 3), (1, 0), (2, 1)]
