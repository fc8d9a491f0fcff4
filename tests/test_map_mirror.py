"""Row N2, CPU: the C restatement of Reprojector::reprojectMap (oracle/svo_oracle_track.c: orc_reproject_map) against
a literal Python walk of the reference's loops (svo/src/reprojector.cpp:64-142, 151-153, 206-217) over the same
plain-array map -- keyframes closest first, every Feature of a keyframe in list order, "project a point only once",
candidates afterwards in list order, push_back into grid cells, a stable sort per cell -- plus the properties the
batched drop-in relies on (one cell per point, cells in visiting order, the first-batch cut)."""
import numpy as np
import pytest

from helpers import camera_models, oracle_reproject_map, random_map
from oracle import pytrack
from rpg_svo_amd import se3


def literal_walk(mp, cam, orc):
    P = mp["pos"].shape[0]
    n_frames = mp["T"].shape[0]
    cells = [[] for _ in range(mp["n_cols"] * mp["n_rows"])]
    kf_count = np.zeros(n_frames, dtype=np.int32)
    last_projected = np.zeros(P, dtype=bool)
    point_cell = np.full(P, -2, dtype=np.int32)
    T_cur = mp["T"][mp["cur"]]

    def reproject_point(p):   # Reprojector::reprojectPoint (:206-217) through the oracle's own w2c / isInFrame
        k, _ = orc.reproject_point(cam, T_cur, mp["pos"][p], mp["cell_size"], mp["n_cols"])
        point_cell[p] = k
        if k >= 0:
            cells[k].append(p)
        return k >= 0

    for rank in range(n_frames):
        fs = np.nonzero(mp["kf_rank"] == rank)[0]
        if fs.size == 0:
            continue
        f = int(fs[0])
        fts = []   # the keyframe's fts_: (position, point)
        for p in range(P):
            if mp["type"][p] < 2:
                continue
            for o in range(mp["obs_begin"][p], mp["obs_begin"][p] + mp["obs_count"][p]):
                if mp["obs_frame"][o] == f and mp["obs_order"][o] >= 0:
                    fts.append((int(mp["obs_order"][o]), p))
        for _, p in sorted(fts):
            if last_projected[p]:
                continue
            last_projected[p] = True
            if reproject_point(p):
                kf_count[f] += 1
    cands = sorted((int(mp["order"][p]), p) for p in range(P) if mp["type"][p] == 1)
    for _, p in cands:
        reproject_point(p)
    cell_of_rank = np.argsort(mp["cell_rank"])
    visit = []
    for i in range(len(cells)):
        cell = sorted(cells[cell_of_rank[i]], key=lambda p: -mp["type"][p])  # list::sort is stable, so is sorted()
        visit += [(p, i) for p in cell]
    return point_cell, kf_count, visit


@pytest.mark.parametrize("kind", ["pinhole", "atan"])
def test_oracle_restatement_is_the_literal_walk(oracle, kind):
    cam = camera_models()[kind]
    orc = pytrack.Track("orc")
    for seed in range(3):
        mp = random_map(cam, n_kfs=8, n_points=250, n_candidates=200, seed=seed, n_overlap=5)
        r = oracle_reproject_map(mp, cam)
        point_cell, kf_count, visit = literal_walk(mp, cam, orc)
        assert np.array_equal(r["point_cell"], point_cell)
        assert np.array_equal(r["kf_count"], kf_count)
        assert [(int(p), int(c)) for p, c in zip(r["visit_point"], r["visit_cell"])] == visit
        assert r["header"][1] == (point_cell >= 0).sum() == len(visit) and r["header"][4] == mp["n_cols"] * mp["n_rows"]
        # a trial is a visit with a close view; its observation is one of the point's
        t = r["visit_trial"]
        assert np.array_equal(t[t >= 0], np.arange((t >= 0).sum()))
        for v in np.nonzero(t >= 0)[0]:
            p, o = r["visit_point"][v], r["trial_obs"][t[v]]
            assert mp["obs_begin"][p] <= o < mp["obs_begin"][p] + mp["obs_count"][p]
            assert r["trial_cell"][t[v]] == r["visit_cell"][v] and np.array_equal(r["trial_pos"][t[v]], mp["pos"][p])
        assert (t >= 0).sum() > 20 and (t < 0).sum() > 0   # both kinds occur


def test_first_batch_cut_and_continuation(oracle):
    """Cells are taken until max_cells_with_trials of them hold a trial; a second call from end_cell continues the walk:
    together they are the uncut walk (the drop-in's two batches)."""
    cam = camera_models()["pinhole"]
    mp = random_map(cam, seed=7)
    full = oracle_reproject_map(mp, cam)
    a = oracle_reproject_map(mp, cam, 0, 40)
    end = int(a["header"][4])
    cells_with_trials = len(set(a["trial_cell"].tolist()))
    assert cells_with_trials == 40 and a["trial_cell"].max() == end - 1      # the walk stops right after the 40th such cell
    b = oracle_reproject_map(mp, cam, end, 1 << 30)
    assert np.array_equal(np.concatenate([a["visit_point"], b["visit_point"]]), full["visit_point"])
    assert np.array_equal(np.concatenate([a["trial_obs"], b["trial_obs"]]), full["trial_obs"])
    assert np.array_equal(np.concatenate([a["visit_trial"], np.where(b["visit_trial"] >= 0, b["visit_trial"] + a["header"][3], -1)]),
                          full["visit_trial"])
    z = oracle_reproject_map(mp, cam, 0, 0)
    assert z["header"][2] == 0 and z["header"][3] == 0 and z["header"][4] == 0
