"""N3 end to end: the product's ROS-free replay tool (rpg_svo_amd/host/tools/svo_replay.cpp) on a
dataset in the reference's on-disk layout, once linked against the reference's own translation units
(CPU) and once against the drop-in bodies + libsvo_hip.so (MI355X); the two outputs -- traj_estimate.txt
in the TUM format the reference's evaluation scripts read, and the per-frame trace CSV the reference
writes itself -- must tell the same story."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
import pypipeline as pp  # noqa: E402

from rpg_svo_amd import dataset, se3, synth, trace  # noqa: E402

BUILD = os.path.join(ROOT, "tests", "dropin", "_build")
CAM = "pinhole:752,480,315.5,315.5,376,240"


def _tool(flavour):
    return os.path.join(BUILD, f"svo_replay_{flavour}")


@pytest.fixture(scope="module")
def replay_dataset(tmp_path_factory):
    pp.build("all")  # no-op without the reference checkout
    if not (os.path.exists(_tool("ref")) and os.path.exists(_tool("hip"))):
        pytest.skip("tests/dropin/_build/svo_replay_{ref,hip} absent and no reference checkout to build them from")
    d = str(tmp_path_factory.mktemp("synth_ds"))
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "replay.py"), "--write-dataset", d, "--frames", "60"],
                   check=True, capture_output=True)
    return d


def _run(flavour, ds, out, mapper_thread=True):
    os.makedirs(out, exist_ok=True)
    p = subprocess.run([_tool(flavour), "--dataset", ds, "--out", out, "--cam", CAM, "--mapper-thread", "1" if mapper_thread else "0"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    traj = np.loadtxt(os.path.join(out, "traj_estimate.txt"))
    with open(os.path.join(out, "svo.csv")) as fh:
        header = fh.readline().strip().split(",")
    rows = np.loadtxt(os.path.join(out, "svo.csv"), delimiter=",", skiprows=1)
    return traj, header, rows


def test_dataset_replay_reference_flavour_tracks(replay_dataset, tmp_path):
    """CPU only: pins the tool itself (dataset reader, first-frame set-up, writers)."""
    traj, header, rows = _run("ref", replay_dataset, str(tmp_path / "ref"))
    ts, names, T_gt = dataset.read_trajectory_file(replay_dataset)
    assert traj.shape == (60, 8) and np.allclose(traj[:, 0], ts[:60], atol=1e-6)
    assert {"sparse_img_align", "reproject", "pose_optimizer", "tot_time", "img_align_n_tracked", "repr_n_new_references",
            "sfba_n_edges_final"} <= set(header)
    assert rows.shape == (59, len(header))          # one trace row per addImage call
    # the estimate follows the ground truth (position, metres)
    gt_pos = se3.inv(T_gt[:60])[:, 9:]
    assert np.abs(traj[:, 1:4] - gt_pos).max() < 0.02


def test_dataset_replay_drop_in_host_code_on_the_mock_device(replay_dataset, tmp_path):
    """The tool linked with the drop-in bodies and the mock device (tests/host/mock_svo_hip.cpp +
    tests/dropin/mock_compute_oracle.cpp, CPU suite): same files, same counters as the all-reference tool."""
    if not os.path.exists(_tool("hipmock")):
        pytest.skip("tests/dropin/_build/svo_replay_hipmock not built")
    # --mapper-thread 0: with DepthFilter's thread running, WHEN a seed converges -- hence which frame first sees its point --
    # depends on how far the thread got, in both flavours (the mock computes with the CPU oracle: on a loaded machine its
    # mapper lags by whole keyframes and the candidate counts of the two runs drift apart by dozens).  Without the thread
    # the replay is deterministic and the two flavours make the same decisions frame by frame.
    traj_r, header_r, rows_r = _run("ref", replay_dataset, str(tmp_path / "ref"), mapper_thread=False)
    traj_m, header_m, rows_m = _run("hipmock", replay_dataset, str(tmp_path / "mock"), mapper_thread=False)
    assert header_r == header_m and rows_r.shape == rows_m.shape and traj_r.shape == traj_m.shape
    assert np.abs(traj_r[:, 1:] - traj_m[:, 1:]).max() < 1e-4
    col = {n: i for i, n in enumerate(header_r)}
    for name in ("img_align_n_tracked", "repr_n_mps", "repr_n_new_references", "sfba_n_edges_final", "n_candidates", "dropout"):
        same = np.mean(rows_r[:, col[name]] == rows_m[:, col[name]])
        assert same >= 0.95, (name, same)
        if name == "n_candidates":
            assert np.abs(rows_r[:, col[name]] - rows_m[:, col[name]]).max() <= 3


@pytest.mark.gpu
def test_dataset_replay_hip_matches_reference(replay_dataset, tmp_path, gpu_device):
    traj_r, header_r, rows_r = _run("ref", replay_dataset, str(tmp_path / "ref"), mapper_thread=False)
    traj_h, header_h, rows_h = _run("hip", replay_dataset, str(tmp_path / "hip"), mapper_thread=False)
    assert header_r == header_h and rows_r.shape == rows_h.shape and traj_r.shape == traj_h.shape
    # TUM-format poses (timestamp tx ty tz qx qy qz qw).  Both flavours run with --mapper-thread 0 (with DepthFilter's thread
    # running, WHEN a seed converges also depends on how far the thread got: a flaky comparison).  What is left: which frame
    # first sees a new candidate point depends on
    # last-bit differences of the pose a seed is updated with, and a feature set that differs by one point moves a pose
    # by ~1e-4 m on this 0.4 m trajectory (measured maxima of single runs on the MI355X box: 0.6e-4 .. 1.04e-4 m; both
    # flavours stay within 1.4 mm RMSE of the ground truth, bench.py dropin_sequence).  Positions to 2.5e-4 m, orientation
    # to 2.5e-4 rad; the per-frame counters below are the sharper statement.
    assert np.abs(traj_r[:, 1:4] - traj_h[:, 1:4]).max() < 2.5e-4
    dq = np.abs(np.sum(traj_r[:, 4:8] * traj_h[:, 4:8], axis=1))
    assert (2 * np.arccos(np.clip(dq, -1, 1))).max() < 2.5e-4
    # the counters of the trace: same decisions frame by frame
    col = {n: i for i, n in enumerate(header_r)}
    for name in ("img_align_n_tracked", "repr_n_mps", "repr_n_new_references", "sfba_n_edges_final", "n_candidates", "dropout"):
        same = np.mean(rows_r[:, col[name]] == rows_h[:, col[name]])
        # a seed whose variance sits on the convergence threshold can converge one update later in one of the
        # runs (float-level differences of the pose it is updated with); the candidate count then differs by
        # one for a frame or two
        assert same >= (0.85 if name == "n_candidates" else 0.95), (name, same)
        if name == "n_candidates":
            assert np.abs(rows_r[:, col[name]] - rows_h[:, col[name]]).max() <= 3
