"""Row N2, GPU: svo_hip_reproject_map (the device-resident map mirror, csrc/map_mirror.hip) against the oracle's
restatement of Reprojector::reprojectMap (svo/src/reprojector.cpp:64-142, 151-153, 206-217; point.cpp:97-117) on
random maps, through the C ABI.  Integer results (cells, counts, visiting order, trials, chosen observations) are
required to be IDENTICAL; the projections are f64 arithmetic (quaternion rotation, the camera model): 1e-9 px."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import CAMERA_KINDS, camera_models, FUZZ, fuzz_rng, oracle_reproject_map, random_map
from rpg_svo_amd import capi
from rpg_svo_amd.map_mirror import Grid, MapMirror

pytestmark = pytest.mark.gpu


def upload(mp, device, capacity=None):
    P, O = mp["pos"].shape[0], mp["obs_frame"].shape[0]
    m = MapMirror(capacity or max(P, 1), max(O, 1), device)
    m.patch(np.arange(P), mp["pos"], mp["type"], mp["order"], mp["obs_begin"], mp["obs_count"], obs_index=np.arange(O),
            obs_frame=mp["obs_frame"], obs_order=mp["obs_order"], obs_level=mp["obs_level"], obs_type=mp["obs_type"],
            obs_px=mp["obs_px"], obs_f=mp["obs_f"], obs_grad=mp["obs_grad"])
    return m


def run(m, mp, cam, device, **kw):
    T = torch.as_tensor(mp["T"], dtype=torch.float64, device=device)
    rank = torch.as_tensor(mp["kf_rank"], dtype=torch.int32, device=device)
    grid = Grid.for_camera(cam.width, cam.height, mp["cell_size"], mp["cell_order"], device)
    return m.reproject(cam, T, mp["cur"], rank, grid, **kw)


def compare(r, o, P, px_tol):
    st, E, V, M, end = r.counts()
    assert st == 0 and (E, V, M, end) == tuple(int(x) for x in o["header"][1:5])
    assert np.array_equal(r.point_cell.cpu().numpy()[:P], o["point_cell"])
    assert np.array_equal(r.kf_count.cpu().numpy(), o["kf_count"])
    seen = o["point_cell"] >= -1
    dpx = np.abs(r.point_px.cpu().numpy()[:P][seen] - o["point_px"][seen])
    assert dpx.size == 0 or dpx.max() <= px_tol, dpx.max()
    assert np.array_equal(r.visit_point.cpu().numpy()[:V], o["visit_point"])
    assert np.array_equal(r.visit_cell.cpu().numpy()[:V], o["visit_cell"])
    assert np.array_equal(r.visit_trial.cpu().numpy()[:V], o["visit_trial"])
    assert np.array_equal(r.trial_obs_begin.cpu().numpy()[:M], o["trial_obs"])
    assert np.array_equal(r.trial_obs_end.cpu().numpy()[:M], o["trial_obs"] + 1)
    assert np.array_equal(r.trial_cell.cpu().numpy()[:M], o["trial_cell"])
    assert np.array_equal(r.trial_pos.cpu().numpy()[:M], o["trial_pos"])
    assert M == 0 or np.abs(r.trial_px.cpu().numpy()[:M] - o["trial_px"]).max() <= px_tol
    assert np.all(r.trial_cur.cpu().numpy()[:M] == r.trial_cur.cpu().numpy()[0]) if M else True
    return V, M


@pytest.mark.parametrize("kind", CAMERA_KINDS)
def test_reproject_map_is_the_reference_walk(gpu_device, oracle, kind):
    cam = camera_models()[kind]
    tol = 1e-9   # (tests/test_tracking_gpu.py::test_reproject_points: same arithmetic, same bound)
    for seed, (n_points, n_cand) in enumerate([(900, 700), (2500, 1900), (40, 0), (0, 300), (0, 0)]):
        mp = random_map(cam, n_kfs=12, n_points=n_points, n_candidates=n_cand, seed=seed + 100 * FUZZ)
        m = upload(mp, gpu_device)
        r = run(m, mp, cam, gpu_device)
        V, M = compare(r, oracle_reproject_map(mp, cam), n_points + n_cand, tol)
        if n_points + n_cand > 1000:
            assert V > 500 and 0 < M < V


def test_batches_patches_and_capacities(gpu_device, oracle):
    cam = camera_models()["pinhole"]
    mp = random_map(cam, seed=11 + 100 * FUZZ)
    P = mp["pos"].shape[0]
    m = upload(mp, gpu_device, capacity=P + 50)
    # the first batch (cells until 40 of them hold a trial), then the rest from end_cell
    a = run(m, mp, cam, gpu_device, max_cells_with_trials=40)
    oa = oracle_reproject_map(mp, cam, 0, 40)
    compare(a, oa, P, 1e-9)
    end = int(oa["header"][4])
    compare(run(m, mp, cam, gpu_device, first_cell=end), oracle_reproject_map(mp, cam, end), P, 1e-9)
    # an incremental patch: positions move, types change (a point promoted, one deleted, a candidate gone), a new candidate
    # with a new observation record appended
    rng = fuzz_rng(5)
    idx = rng.choice(P, size=60, replace=False)
    mp["pos"][idx] += rng.normal(0, 0.05, (60, 3))
    mp["type"][idx[:10]] = 3
    mp["type"][idx[10:20]] = 0
    O = mp["obs_frame"].shape[0]
    new_p = P
    mp["pos"] = np.concatenate([mp["pos"], mp["pos"][idx[:1]] + 0.01])
    mp["type"] = np.append(mp["type"], 1).astype(np.int32)
    mp["order"] = np.append(mp["order"], mp["order"].max() + 1).astype(np.int32)
    mp["obs_begin"] = np.append(mp["obs_begin"], O).astype(np.int32)
    mp["obs_count"] = np.append(mp["obs_count"], 1).astype(np.int32)
    for k, v in (("obs_frame", 3), ("obs_order", -1), ("obs_level", 1), ("obs_type", 0)):
        mp[k] = np.append(mp[k], v).astype(mp[k].dtype)
    for k, n in (("obs_px", 2), ("obs_f", 3), ("obs_grad", 2)):
        mp[k] = np.concatenate([mp[k], rng.normal(size=(1, n))])
    which = np.append(idx, new_p)
    m.obs_frame = torch.cat([m.obs_frame, m.obs_frame[:8]]); m.obs_order = torch.cat([m.obs_order, m.obs_order[:8]])
    m.obs_level = torch.cat([m.obs_level, m.obs_level[:8]]); m.obs_type = torch.cat([m.obs_type, m.obs_type[:8]])
    m.obs_px = torch.cat([m.obs_px, m.obs_px[:8]]); m.obs_f = torch.cat([m.obs_f, m.obs_f[:8]]); m.obs_grad = torch.cat([m.obs_grad, m.obs_grad[:8]])
    m.patch(which, mp["pos"][which], mp["type"][which], mp["order"][which], mp["obs_begin"][which], mp["obs_count"][which],
            obs_index=[O], obs_frame=mp["obs_frame"][O:], obs_order=mp["obs_order"][O:], obs_level=mp["obs_level"][O:],
            obs_type=mp["obs_type"][O:], obs_px=mp["obs_px"][O:], obs_f=mp["obs_f"][O:], obs_grad=mp["obs_grad"][O:])
    compare(run(m, mp, cam, gpu_device), oracle_reproject_map(mp, cam), P + 1, 1e-9)
    # capacities: fewer visit / trial slots than needed is reported, nothing is written past them
    r = run(m, mp, cam, gpu_device, max_visits=10, max_trials=10)
    assert r.counts()[0] == 1 and r.counts()[2] == 0 and r.counts()[3] == 0


def test_indirect_match_batch_is_the_direct_one(gpu_device, oracle):
    """svo_hip_find_match_direct_indirect / svo_hip_select_matches_indirect (batch size read on the device, observation
    ranges instead of CSR offsets) against the plain entry points on the same trials: identical outputs, nothing
    written beyond the device-side batch size."""
    from helpers import obs_csr, scene_store
    from rpg_svo_amd import synth, tracking
    from rpg_svo_amd.pyramid import _stream_ptr
    scene = synth.make_track_scene(n_kf=4, n_feat=100, cam=camera_models()["pinhole"])
    T = scene.T_f_w.copy()
    T[scene.cur] = scene.T_cur_prior
    store, frames = scene_store(scene, T_override=T)
    dev = gpu_device
    t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    M = len(scene.obs)
    obs_ptr, obs = obs_csr(scene.obs)
    cur_frame = torch.full((M,), scene.cur, dtype=torch.int32, device=dev)
    pt_pos, px_cur = t(scene.pt_pos, torch.float64), t(scene.px_init, torch.float64)
    matcher = tracking.Matcher(align_max_iter=10, n_pyr_levels=5)
    ref = matcher.find_match_direct(store, scene.cam, frames, cur_frame, pt_pos, obs_ptr, obs, px_cur)
    lib = capi.load()
    cap = M + 37
    pad = lambda x: torch.cat([x, torch.zeros((cap - M,) + tuple(x.shape[1:]), dtype=x.dtype, device=dev)]).contiguous()
    res = matcher.alloc_result(cap, dev)
    res.px_cur.copy_(pad(px_cur))
    res.ok.fill_(-7)
    d_M = t([M], torch.int32)
    ws = torch.empty(lib.svo_hip_match_workspace_bytes(cap), dtype=torch.uint8, device=dev)
    c, fr, ob = capi.camera(scene.cam), frames.struct(), obs.struct()
    ob_begin, ob_end = pad(obs_ptr[:-1].contiguous()), pad(obs_ptr[1:].contiguous())
    cur_p, pos_p = pad(cur_frame), pad(pt_pos)
    capi.check(lib.svo_hip_find_match_direct_indirect(C.byref(store.layout), store.ptr, C.byref(c), C.byref(fr), cap, d_M.data_ptr(),
                                                      cur_p.data_ptr(), pos_p.data_ptr(), ob_begin.data_ptr(), ob_end.data_ptr(),
                                                      C.byref(ob), 5, 10, res.px_cur.data_ptr(), res.ok.data_ptr(), res.ref_obs.data_ptr(),
                                                      res.search_level.data_ptr(), res.A_cur_ref.data_ptr(), res.patch_with_border.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), _stream_ptr(dev)), "find_match_direct_indirect")
    torch.cuda.synchronize()
    for k in ("ok", "px_cur", "ref_obs", "search_level", "A_cur_ref", "patch_with_border"):
        assert torch.equal(getattr(res, k)[:M], getattr(ref, k)), k
    assert (res.ok[M:] == -7).all()   # nothing beyond the device-side batch size is touched
    assert ref.ok.sum().item() > M // 2
    # the selection rule on the same trials, batch size on the device
    cell = t(np.arange(M) // 3, torch.int32)   # three adjacent trials per cell
    outs = []
    for indirect in (False, True):
        n, sel = torch.zeros(1, dtype=torch.int32, device=dev), torch.full((121,), -1, dtype=torch.int32, device=dev)
        f, pos_o = torch.zeros(121, 3, dtype=torch.float64, device=dev), torch.zeros(121, 3, dtype=torch.float64, device=dev)
        lvl_o, has = torch.zeros(121, dtype=torch.int32, device=dev), torch.zeros(121, dtype=torch.uint8, device=dev)
        if indirect:
            capi.check(lib.svo_hip_select_matches_indirect(C.byref(c), cap, d_M.data_ptr(), pad(cell).data_ptr(), res.ok.data_ptr(),
                                                           res.px_cur.data_ptr(), res.search_level.data_ptr(), pos_p.data_ptr(), 120,
                                                           n.data_ptr(), sel.data_ptr(), f.data_ptr(), lvl_o.data_ptr(), pos_o.data_ptr(),
                                                           has.data_ptr(), None, 0, _stream_ptr(dev)))
        else:
            capi.check(lib.svo_hip_select_matches(C.byref(c), M, cell.data_ptr(), ref.ok.data_ptr(), ref.px_cur.data_ptr(),
                                                  ref.search_level.data_ptr(), pt_pos.data_ptr(), 120, n.data_ptr(), sel.data_ptr(),
                                                  f.data_ptr(), lvl_o.data_ptr(), pos_o.data_ptr(), has.data_ptr(), None, 0, _stream_ptr(dev)))
        torch.cuda.synchronize()
        outs.append((n.clone(), sel.clone(), f.clone(), lvl_o.clone(), pos_o.clone(), has.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][0].item() > 20
