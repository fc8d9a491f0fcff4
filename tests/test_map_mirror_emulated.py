"""Row N2 without a GPU: rpg_svo_amd/csrc/map_mirror.hip compiled for the host through tests/host/hip_emu.h -- the C-ABI
entry point svo_hip_reproject_map itself, its one-workgroup kernel run by 1024 host threads (barriers, LDS, the scan's
__shfl_up, LDS atomics emulated) -- against the oracle's restatement of Reprojector::reprojectMap on random maps, exactly
as tests/test_map_mirror_gpu.py does on the device."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import camera_models, FUZZ, fuzz_rng, oracle_reproject_map, random_map
from rpg_svo_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=[0], ids=["default"])
def emu(request):
    from emu_build import build_emulated
    from emu_build import BUILDS
    return build_emulated(BUILDS[request.param])


def _ptr(a):
    return None if a is None else a.ctypes.data


class HostMirror:
    """the records of svo_hip_map in host arrays (what rpg_svo_amd.map_mirror.MapMirror keeps in device tensors)"""

    def __init__(self, lib, P, O):
        self.lib = lib
        self.n_points = self.n_obs = 0
        self.a = dict(d_pos=np.zeros((P, 3)), d_type=np.zeros(P, np.int32), d_order=np.zeros(P, np.int32), d_obs_begin=np.zeros(P, np.int32),
                      d_obs_count=np.zeros(P, np.int32), d_obs_frame=np.zeros(O, np.int32), d_obs_order=np.zeros(O, np.int32),
                      d_obs_level=np.zeros(O, np.int32), d_obs_type=np.zeros(O, np.uint8), d_obs_px=np.zeros((O, 2)),
                      d_obs_f=np.zeros((O, 3)), d_obs_grad=np.zeros((O, 2)))
        self.pending = None

    def patch(self, index, pos, type, order, obs_begin, obs_count, obs_index=None, obs_frame=None, obs_order=None, obs_level=None,
              obs_type=None, obs_px=None, obs_f=None, obs_grad=None):
        c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        index = c(index, np.int32)
        p = dict(index=index, pos=c(pos, np.float64).reshape(-1, 3), type=c(type, np.int32), order=c(order, np.int32),
                 obs_begin=c(obs_begin, np.int32), obs_count=c(obs_count, np.int32))
        if index.size:
            self.n_points = max(self.n_points, int(index.max()) + 1)
        n_obs = 0 if obs_index is None else len(obs_index)
        if n_obs:
            oi = c(obs_index, np.int32)
            self.n_obs = max(self.n_obs, int(oi.max()) + 1)
            p.update(obs_index=oi, obs_frame=c(obs_frame, np.int32), obs_order=c(obs_order, np.int32), obs_level=c(obs_level, np.int32),
                     obs_type=c(obs_type, np.uint8), obs_px=c(obs_px, np.float64).reshape(-1, 2), obs_f=c(obs_f, np.float64).reshape(-1, 3),
                     obs_grad=c(obs_grad, np.float64).reshape(-1, 2))
        self.pending = (p, int(index.size), n_obs)

    def reproject(self, cam, T, cur, kf_rank, mp, first_cell=0, max_cells_with_trials=1 << 30, max_visits=4096, max_trials=4096):
        n_frames = T.shape[0]
        T = np.ascontiguousarray(T, np.float64)
        kf_rank = np.ascontiguousarray(kf_rank, np.int32)
        cell_rank = np.ascontiguousarray(mp["cell_rank"], np.int32)
        zi = lambda n: np.zeros(max(n, 1), np.int32)
        zd = lambda n, k: np.zeros((max(n, 1), k))
        P = max(self.n_points, 1)
        out = dict(d_header=zi(capi.REPROJ_HEADER), d_point_cell=zi(P), d_point_px=zd(P, 2), d_kf_count=zi(n_frames), d_visit_point=zi(max_visits),
                   d_visit_cell=zi(max_visits), d_visit_trial=zi(max_visits), d_trial_cur=zi(max_trials), d_trial_pos=zd(max_trials, 3),
                   d_trial_obs_begin=zi(max_trials), d_trial_obs_end=zi(max_trials), d_trial_cell=zi(max_trials), d_trial_px=zd(max_trials, 2))
        patch = None
        if self.pending is not None:
            p, n_pts, n_obs = self.pending
            self.pending = None
            g = lambda k: _ptr(p.get(k))
            patch = capi.MapPatch(n_pts, n_obs, g("index"), g("pos"), g("type"), g("order"), g("obs_begin"), g("obs_count"), g("obs_index"),
                                  g("obs_order"), capi.Features(g("obs_frame"), g("obs_level"), g("obs_type"), g("obs_px"), g("obs_f"), g("obs_grad")))
        frames = capi.Frames(n_frames, 0, None, T.ctypes.data)
        m = capi.Map(self.n_points, self.n_obs, *[self.a[n].ctypes.data for n, _ in capi.Map._fields_[2:]])
        grid = capi.Grid(mp["cell_size"], mp["n_cols"], mp["n_rows"], mp["n_cols"] * mp["n_rows"], cell_rank.ctypes.data)
        rs = capi.Reprojection(*[out[n].ctypes.data for n, _ in capi.Reprojection._fields_])
        c = capi.camera(cam)
        rc = self.lib.svo_hip_reproject_map(C.byref(c), C.byref(frames), C.c_int(cur), C.c_void_p(kf_rank.ctypes.data), C.byref(m),
                                            C.byref(patch) if patch is not None else None, C.byref(grid), C.c_int(first_cell),
                                            C.c_int(min(max_cells_with_trials, 1 << 30)), C.c_int(max_visits), C.c_int(max_trials), C.byref(rs), None)
        assert rc == 0, rc
        return out


def upload(lib, mp, capacity=None):
    P, O = mp["pos"].shape[0], mp["obs_frame"].shape[0]
    m = HostMirror(lib, capacity or max(P, 1), max(O, 1))
    m.patch(np.arange(P), mp["pos"], mp["type"], mp["order"], mp["obs_begin"], mp["obs_count"], obs_index=np.arange(O),
            obs_frame=mp["obs_frame"], obs_order=mp["obs_order"], obs_level=mp["obs_level"], obs_type=mp["obs_type"],
            obs_px=mp["obs_px"], obs_f=mp["obs_f"], obs_grad=mp["obs_grad"])
    return m


def compare(r, o, P, px_tol):
    st, E, V, M, end = [int(x) for x in r["d_header"][:5]]
    assert st == 0 and (E, V, M, end) == tuple(int(x) for x in o["header"][1:5])
    assert np.array_equal(r["d_point_cell"][:P], o["point_cell"])
    assert np.array_equal(r["d_kf_count"], o["kf_count"])
    seen = o["point_cell"] >= -1
    dpx = np.abs(r["d_point_px"][:P][seen] - o["point_px"][seen])
    assert dpx.size == 0 or dpx.max() <= px_tol, dpx.max()
    for a, b in (("d_visit_point", "visit_point"), ("d_visit_cell", "visit_cell"), ("d_visit_trial", "visit_trial")):
        assert np.array_equal(r[a][:V], o[b]), a
    assert np.array_equal(r["d_trial_obs_begin"][:M], o["trial_obs"])
    assert np.array_equal(r["d_trial_obs_end"][:M], o["trial_obs"] + 1)
    assert np.array_equal(r["d_trial_cell"][:M], o["trial_cell"])
    assert np.array_equal(r["d_trial_pos"][:M], o["trial_pos"])
    assert M == 0 or np.abs(r["d_trial_px"][:M] - o["trial_px"]).max() <= px_tol
    return V, M


@pytest.mark.parametrize("kind", ["pinhole", "atan"])
def test_emulated_kernel_is_the_reference_walk(emu, oracle, kind):
    """Integer results (cells, counts, visiting order, trials, chosen observations) identical to the oracle; projections,
    host-compiled without contraction like the oracle, to 1e-12 px."""
    cam = camera_models()[kind]
    for seed, (n_points, n_cand) in enumerate([(900, 700), (2500, 1900), (40, 0), (0, 300), (0, 0)]):
        mp = random_map(cam, n_kfs=12, n_points=n_points, n_candidates=n_cand, seed=seed + 100 * FUZZ)
        m = upload(emu, mp)
        r = m.reproject(cam, mp["T"], mp["cur"], mp["kf_rank"], mp)
        V, M = compare(r, oracle_reproject_map(mp, cam), n_points + n_cand, 1e-12)
        if n_points + n_cand > 1000:
            assert V > 500 and 0 < M < V


def test_emulated_batches_and_patches(emu, oracle):
    cam = camera_models()["pinhole"]
    mp = random_map(cam, seed=11 + 100 * FUZZ)
    P = mp["pos"].shape[0]
    m = upload(emu, mp, capacity=P + 50)
    a = m.reproject(cam, mp["T"], mp["cur"], mp["kf_rank"], mp, max_cells_with_trials=40)
    oa = oracle_reproject_map(mp, cam, 0, 40)
    compare(a, oa, P, 1e-12)
    end = int(oa["header"][4])
    compare(m.reproject(cam, mp["T"], mp["cur"], mp["kf_rank"], mp, first_cell=end), oracle_reproject_map(mp, cam, end), P, 1e-12)
    # an incremental patch: positions move, types change, a new candidate with a new observation record is appended
    rng = fuzz_rng(5)
    idx = rng.choice(P, 60, replace=False)
    mp["pos"][idx] += rng.normal(0, 0.02, (60, 3))
    promoted = idx[mp["type"][idx] == 2][:5]
    mp["type"][promoted] = 3
    dead = idx[mp["type"][idx] != 0][-4:]
    mp["type"][dead] = 0
    O = mp["obs_frame"].shape[0]
    new_p = P
    mp["pos"] = np.vstack([mp["pos"], mp["pos"][idx[0]] + [0.05, 0.02, 0.0]])
    for k, v in (("type", 1), ("order", int(mp["order"].max()) + 1), ("obs_begin", O), ("obs_count", 1)):
        mp[k] = np.append(mp[k], v).astype(np.int32)
    for k, v in (("obs_frame", 3), ("obs_order", -1), ("obs_level", 1)):
        mp[k] = np.append(mp[k], v).astype(np.int32)
    mp["obs_type"] = np.append(mp["obs_type"], 0).astype(np.uint8)
    mp["obs_px"] = np.vstack([mp["obs_px"], [100.0, 120.0]])
    mp["obs_f"] = np.vstack([mp["obs_f"], [0.0, 0.0, 1.0]])
    mp["obs_grad"] = np.vstack([mp["obs_grad"], [1.0, 0.0]])
    touched = np.unique(np.concatenate([idx, [new_p]]))
    m.a["d_obs_frame"] = np.append(m.a["d_obs_frame"], 0).astype(np.int32)  # room for the appended record
    for k, w in (("d_obs_order", np.int32), ("d_obs_level", np.int32), ("d_obs_type", np.uint8)):
        m.a[k] = np.append(m.a[k], 0).astype(w)
    for k, w in (("d_obs_px", 2), ("d_obs_f", 3), ("d_obs_grad", 2)):
        m.a[k] = np.vstack([m.a[k], np.zeros((1, w))])
    m.patch(touched, mp["pos"][touched], mp["type"][touched], mp["order"][touched], mp["obs_begin"][touched], mp["obs_count"][touched],
            obs_index=[O], obs_frame=mp["obs_frame"][O:], obs_order=mp["obs_order"][O:], obs_level=mp["obs_level"][O:],
            obs_type=mp["obs_type"][O:], obs_px=mp["obs_px"][O:], obs_f=mp["obs_f"][O:], obs_grad=mp["obs_grad"][O:])
    compare(m.reproject(cam, mp["T"], mp["cur"], mp["kf_rank"], mp), oracle_reproject_map(mp, cam), P + 1, 1e-12)
