"""bench.py's bookkeeping (CPU only): the algorithmic-byte formula of SURVEY 8(d) and the ATE helper."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_algorithmic_bytes_formula():
    # SURVEY 8(d) worked example: VGA-4, N = 200 all visible, 4 levels, 10 iterations per level:
    # 4*200*49 (7x7 reference windows) + 4*200*10*25 (5x5 current windows) + geometry + fixed I/O
    B = 3
    n = np.full(B, 200.0)
    iters = np.zeros((B, 8))
    iters[:, :4] = 10
    got = bench.algorithmic_bytes(n, n, iters, 3, 0)
    per_frame = 4 * 200 * 49 + 4 * 200 * 10 * 25 + 41 * 200 + 540
    assert got == B * per_frame
    # only the levels of the schedule count; untracked patches fetch no windows
    iters2 = np.zeros((1, 8)); iters2[0, 2:5] = [3, 2, 4]
    got = bench.algorithmic_bytes(np.array([120.0]), np.array([100.0]), iters2, 4, 2)
    assert got == 100 * ((49 + 25 * 3) + (49 + 25 * 2) + (49 + 25 * 4)) + 41 * 120 + 540


def test_horn_ate_invariance():
    rng = np.random.default_rng(0)
    P = rng.normal(size=(50, 3))
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    Q = P @ R.T + np.array([3.0, -1.0, 2.0])
    assert bench.horn_ate(P, Q) < 1e-12
    assert abs(bench.horn_ate(P, Q + np.array([0.0, 0.0, 0.0])) - 0.0) < 1e-12
    Qn = Q.copy(); Qn[0] += 0.5   # one outlier of 0.5*sqrt(3) m among 50 poses: rmse ~ 0.87/sqrt(50) = 0.12
    assert 0.08 < bench.horn_ate(P, Qn) < 0.16


def test_workloads_match_baseline_configs():
    w = bench.WORKLOADS
    assert w["vga4_n200_sparse_align"][:7] == (640, 480, 400.0, 4, 3, 0, 200)          # configs[1]
    assert w["xga5_n1000_sparse_align"][:7] == (1280, 960, 800.0, 5, 4, 0, 1000)       # configs[3]
    assert w["svo_default_752_l4to2_n120"][:7] == (752, 480, 315.5, 5, 4, 2, 120)      # configs[0]/[4] geometry
    assert bench.HBM_PEAK_GBS == 8000.0


def test_valu_roofline_pricing():
    """roofline_valu: class counters x measured issue cost over the SIMD cycles of the launch."""
    n_cu, n_simd = bench.N_CU, bench.N_SIMD
    raw = {"SQ_INSTS_VALU": 1000.0 * n_simd, "SQ_BUSY_CU_CYCLES": 10000.0 * n_cu,        # 1000 VALU per SIMD in 10000 cycles
           "SQ_INSTS_VALU_FMA_F32": 400.0 * n_simd, "SQ_INSTS_VALU_FMA_F64": 100.0 * n_simd, "SQ_INSTS_VALU_TRANS_F64": 10.0 * n_simd,
           "SQ_WAVE_CYCLES": 100.0, "SQ_WAIT_ANY": 50.0}
    r = bench.valu_roofline(raw, kernel_ms=1.0)
    busy = 400 * 2.4 + 100 * 3.5 + 10 * 9.6 + (1000 - 510) * 3.0
    assert abs(r["frac"] - busy / 10000.0) < 1e-12 and r["achieved"] == r["frac"]
    assert abs(r["lower_bound_2_cycles_per_instruction"] - 0.2) < 1e-12
    assert abs(r["classified_by_counters_frac"] - 0.51) < 1e-12
    assert r["wave_cycles_waiting_frac"] == 0.5 and r["wave_cycles_issuing_frac"] is None
    # without the busy counter the nominal clock stands in
    raw.pop("SQ_BUSY_CU_CYCLES")
    r2 = bench.valu_roofline(raw, kernel_ms=1.0)
    assert abs(r2["cu_cycles_per_launch"] - 1e-3 * bench.CLOCK_GHZ * 1e9) < 1e-6


def test_last_stdout_line_is_compact_and_round_trips():
    """the driver could not parse a 21 KB line.  The last stdout line is compact_line(result): below 4 KB
    whatever the legs put into the full object, valid JSON, and carrying the contract keys + roofline (with counter
    traffic) + cpu_baseline + parity."""
    import json
    full = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_default.json")))   # a real 21 KB result object
    assert len(json.dumps(full)) > 15000
    full["dtype_short"] = "f32 pixels, f64 pose"
    full["cpu_baseline"]["sample_short"] = "2048 frame pairs, 1 thread"
    line = json.dumps(bench.compact_line(full, "bench_details.json"))
    assert len(line) < bench.COMPACT_LIMIT and "\n" not in line
    c = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "details"):
        assert k in c, k
    assert c["config"]["workload"] == "vga4_n200_sparse_align" and "model" not in c["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "ms", "kernel"):
        assert k in c["roofline"], k
    assert abs(c["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5 * full["roofline"]["frac"]
    assert abs(c["value"] - full["value"]) <= 1e-5 * full["value"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    assert c["legs"]["full_track"]["ms_per_step"] > 0 and c["legs"]["dropin_sequence"]["ms_hip"] > 0
    # an absurdly large leg cannot push the line over the limit: optional blocks are dropped, the contract keys stay
    full["full_track"]["stages_ms"] = {f"stage{i}": float(i) for i in range(2000)}
    c2 = bench.compact_line(full, None)
    assert len(json.dumps(c2)) < bench.COMPACT_LIMIT and "roofline" in c2 and "cpu_baseline" in c2
    # NaN / inf never reach the line as bare tokens json.loads of a strict parser would refuse
    assert "NaN" not in line and "Infinity" not in line


def test_lds_port_use_ratios_and_missing_counters():
    """bench.lds_port_use: LDS port busy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES (both summed over the CUs); a counter that was
    not collected gives None, never an exception (the leg runs unattended under rocprofv3)."""
    raw = {"GRBM_GUI_ACTIVE": 8000.0, "SQ_LDS_IDX_ACTIVE": 128000.0, "SQ_LDS_BANK_CONFLICT": 32000.0, "SQ_INSTS_LDS": 50000.0,
           "SQ_WAVE_CYCLES": 1e6, "SQ_WAIT_INST_LDS": 100.0, "SQ_ACTIVE_INST_LDS": 2e5, "SQ_BUSY_CU_CYCLES": 256000.0}
    r = bench.lds_port_use(raw)
    assert r["port_busy_frac"] == 0.5
    assert r["bank_conflict_frac_of_port_cycles"] == 0.25
    assert r["port_cycles_per_lds_instruction"] == 2.56
    assert r["wave_cycles_waiting_for_lds_issue_frac"] == 4.0 * 100.0 / 1e6  # the counter is in units of 4 cycles
    assert abs(r["sdk_formula_idx_active_over_gui_active_x_cus"] - 128000.0 / (8000.0 * 256)) < 1e-12
    empty = bench.lds_port_use({})
    assert set(empty) == set(r) and all(v is None for v in empty.values())
    __import__("json").dumps(r)


def test_time_budget_arithmetic():
    """--time-budget: seconds since process start; 0 = no limit"""
    import bench
    t0 = bench.T_PROCESS_START
    assert bench.time_left(0.0) == float("inf") and bench.time_left(-1.0) == float("inf")
    assert abs(bench.time_left(300.0, now=t0 + 100.0) - 200.0) < 1e-9
    assert bench.time_left(300.0, now=t0 + 400.0) < 0
    # the optional passes stand back for the full-track counter leg: with 100 s left none of them starts, the three the
    # headline needs still do (pmc_leg's condition)
    left = bench.time_left(300.0, now=t0 + 200.0)
    assert left < bench.PMC_PASS_RESERVE_S + bench.FULL_TRACK_PMC_RESERVE_S and left > bench.PMC_PASS_RESERVE_S


def test_dispatches_are_attributed_to_their_stage_by_their_place_in_the_step():
    """align_kernel serves findMatchDirect AND the depth filter; bench.attribute_dispatches tells its
    launches apart by where they stand in the step (the kernel that opens a stage), per dispatch, for the kernel trace and
    for every counter pass alike; the workload's set-up launches (the same kernels, before the first step) are dropped."""
    import bench
    rows, k = [], [0]

    def d(name, **q):
        k[0] += 1
        rows.append((k[0], name, q))

    for _ in range(3):  # set-up: update_seeds passes without a matcher
        d("seed_prepare_kernel", ms=9.0); d("epi_scan_kernel", ms=9.0); d("align_kernel", ms=9.0)
    for step in range(2):
        d("match_prepare_kernel", ms=0.2); d("warp_kernel", ms=0.5)
        for _ in range(3):
            d("align_kernel", ms=0.3, C=10.0)
        d("pose_opt_wave_kernel", ms=0.25); d("pose_opt_kernel", ms=0.01)
        d("seed_prepare_kernel", ms=0.7); d("epi_scan_kernel", ms=2.0 + step)
        for _ in range(3):
            d("align_kernel", ms=0.6, C=1.0)
    import random
    random.Random(3).shuffle(rows)  # (any order in: sorted by the order key)
    out = bench.attribute_dispatches(rows, 2)
    assert set(out) == {"find_match_direct/match_prepare_kernel", "find_match_direct/warp_kernel", "find_match_direct/align_kernel",
                        "pose_optimize/pose_opt_wave_kernel", "pose_optimize/pose_opt_kernel", "update_seeds/seed_prepare_kernel",
                        "update_seeds/epi_scan_kernel", "update_seeds/align_kernel"}
    assert abs(out["find_match_direct/align_kernel"]["ms"] - 0.9) < 1e-12 and out["find_match_direct/align_kernel"]["C"] == 30.0
    assert abs(out["update_seeds/align_kernel"]["ms"] - 1.8) < 1e-12 and out["update_seeds/align_kernel"]["C"] == 3.0
    assert out["update_seeds/align_kernel"]["launches_per_step"] == 3 and out["update_seeds/epi_scan_kernel"]["ms"] == 2.5
    assert bench.attribute_dispatches(rows, 5) == {}  # fewer steps in the run than asked for
