#!/usr/bin/env python
"""bench.py -- frames/s of sparse image alignment (+ the Gauss-Newton pose
refinement it contains) on synthetic VGA pyramid batches, plus the read-outs the
scope table asks for next to it.

A "step" is one pass of the hot path (svo_hip_sparse_align, K1) over one batch
of B independent (reference frame, current frame) problems per GPU, with the
image pyramids and feature arrays already resident in HBM.  Workload at N=1 is
BASELINE.json configs[1]: 640x480 mono, 4 pyramid levels (3 -> 0), 200 reference
patches, SparseImgAlign only.  `value` is that and nothing else.

    python bench.py --gpus N --steps K --warmup W        (N > 1: spawns one rank per GPU itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Extra keys of that line (N = 1 only, each leg is
independent and never changes `value`):
  roofline / parity / cpu_baseline   the headline kernel (K1)
  align_plus_refine                  K1 + pose_optimizer (K4) timed together
  full_track                         BASELINE configs[2]: K1 + findMatchDirect + K4 + updateSeeds,
                                     per-stage times and rooflines, parity against the CPU chain
  noise_sigma2                       the headline workload with image noise (benchmark_node.cpp:166-176)
  config3_xga5_b64                   BASELINE configs[3]: 1280x960, 5 levels, 1000 patches, 64 frames
  rig_replay                         BASELINE configs[4] shape: one camera stream per rank, a pose gather
                                     per frame (also reported for N > 1)
  k0_pyramid / dropin_sequence       pyramid builder roofline; the reference pipeline with HIP bodies
The oracle (oracle/, CPU restatement + the reference's own translation units) is used here only
for the `cpu_baseline` and `parity` legs; it is never the thing measured.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from rpg_svo_amd import capi, se3, synth  # noqa: E402
from rpg_svo_amd.pyramid import PyramidStore  # noqa: E402
from rpg_svo_amd.sparse_img_align import SparseImgAlign, marshal_problem  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
N_CU = 256
PMC_PASS_TIMEOUT_S = 90  # per rocprofv3 --pmc child run of the pmc leg
T_PROCESS_START = time.time()
# --time-budget: seconds (since this process started) after which no further OPTIONAL work is begun: extra legs, the counter
# passes beyond the three the headline line needs (FETCH_SIZE, WRITE_SIZE, the SQ group), the full-track counter leg.  The
# headline measurement, its roofline / traffic and the CPU baseline are never skipped; whatever is skipped says so.
FULL_TRACK_PMC_RESERVE_S = 130.0
PMC_PASS_RESERVE_S = 25.0


def time_left(budget_s: float, now: float | None = None) -> float:
    if not budget_s or budget_s <= 0:
        return float("inf")
    return budget_s - ((time.time() if now is None else now) - T_PROCESS_START)
N_SIMD = 1024          # 256 CUs x 4 SIMD-32
CLOCK_GHZ = 2.4

WORKLOADS = {
    # name: (width, height, f, n_levels, max_level, min_level, n_patches, margin, cell)
    "vga4_n200_sparse_align": (640, 480, 400.0, 4, 3, 0, 200, 28, 32),
    "svo_default_752_l4to2_n120": (752, 480, 315.5, 5, 4, 2, 120, 56, 40),
    "xga5_n1000_sparse_align": (1280, 960, 800.0, 5, 4, 0, 1000, 56, 32),
    # configs[1]'s shape (4 levels, 3 -> 0, 200 patches) on the image size of the cameras the reference ships (svo_ros/param)
    "ref752_4_n200_sparse_align": (752, 480, 414.5, 4, 3, 0, 200, 28, 32),
}


def reference_cameras() -> dict:
    """The two calibrations the reference ships -- svo_ros/param/camera_pinhole.yaml (vk::PinholeCamera with radial-
    tangential distortion, cam_d0 = -0.283076) and camera_atan.yaml (vk::ATANCamera, cam_d0 = 0.9320) -- and, as the
    yardstick, camera_pinhole.yaml's intrinsics without its distortion (the model every other number of this file is on)."""
    return {"pinhole_undistorted": synth.Camera(752, 480, 414.536145, 414.284429, 348.804988, 240.076451),
            "radtan": synth.Camera.radtan(752, 480, 414.536145, 414.284429, 348.804988, 240.076451, -0.283076, 0.066674, 0.000896, 0.000778),
            "atan": synth.Camera.atan(752, 480, 0.509326, 0.796651, 0.45905, 0.510056, 0.9320)}
EXTRA_KEYS = {"f64": "f64_partials", "refine": "align_plus_refine", "full": "full_track", "full_easy": "full_track_easy", "long_scan": "full_track_long_scan", "stream": "stream_replay", "noise": "noise_sigma2", "config3": "config3_xga5_b64",
              "k0": "k0_pyramid", "dropin": "dropin_sequence", "cameras": "reference_cameras"}


def algorithmic_bytes(n_patches: np.ndarray, n_tracked: np.ndarray, iters: np.ndarray, max_level: int, min_level: int) -> float:
    """per frame  sum_l N_l*(49 + 25*I_l) + per-patch geometry + pose/H I/O.
    N_l is taken as the tracked-patch count; geometry is 41 B/patch here (px 2xf64,
    xyz_ref 3xf64, valid u8) and the fixed I/O is 540 B (poses in/out 2x96, H 288,
    counters 60)."""
    lv = slice(min_level, max_level + 1)
    per_level = n_tracked[:, None] * (49.0 + 25.0 * iters[:, lv])
    return float(per_level.sum() + 41.0 * n_patches.sum() + 540.0 * len(n_patches))


def horn_ate(P: np.ndarray, Q: np.ndarray) -> float:
    """ATE RMSE after Horn alignment (svo_analysis/.../evaluate_ate.py:47-80)."""
    Pc, Qc = P - P.mean(0), Q - Q.mean(0)
    W = Pc.T @ Qc
    U, _, Vt = np.linalg.svd(W.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    t = Q.mean(0) - R @ P.mean(0)
    err = (R @ P.T).T + t - Q
    return float(np.sqrt((err ** 2).sum(1).mean()))


def roofline(kernel: str, alg_bytes: float, ms: float, **extra) -> dict:
    gbs = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else float("nan")
    d = {"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": gbs / HBM_PEAK_GBS, "ms": ms, "algorithmic_bytes_per_launch": alg_bytes}
    d.update(extra)
    return d


DETAILS_FILE = "bench_details.json"   # the full result object (every leg); the LAST stdout line is the compact summary
COMPACT_LIMIT = 6000                  # bytes: the driver parses the tail of stdout (BENCH_r03: a 21 KB line was not parsed -- its record
                                      # holds the last 14 161 bytes of stdout; rounds 4-5 stayed below 4 000, this round's line carries the
                                      # per-kernel table AND the reference's cameras)


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def _num(x, nd=6):
    """floats to `nd` significant digits (the compact line only; the details file keeps full precision)"""
    if isinstance(x, float) and x == x and abs(x) != float("inf"):
        return float(f"{x:.{nd}g}")
    if isinstance(x, dict):
        return {k: _num(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_num(v, nd) for v in x]
    return x


def compact_line(result: dict, details_path: str | None) -> dict:
    """The one line the driver parses: the contract keys, `roofline` (with counter traffic), `cpu_baseline`,
    `parity`, one headline number per extra leg, and where the full object went.  Kept below COMPACT_LIMIT bytes."""
    c = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "data", "value_sparse_align_only"))
    c["dtype"] = result.get("dtype_short", result.get("dtype"))
    c["config"] = _pick(result.get("config", {}), ("workload", "image", "pyr_levels", "schedule", "patches_per_frame",
                                                   "frames_per_step_per_gpu", "n_iter_cap", "image_noise_sigma", "parallelism",
                                                   "k1_kernel", "hip_graph", "mean_gn_iterations_per_frame", "mean_tracked_patches",
                                                   "k1_launches_before_timed_region", "timed_launches", "median_pose_error_vs_gt",
                                                   "matches_per_frame", "pose_refine_obs_after_pruning",
                                                   "median_pose_error_vs_gt_after_refine"))
    c["roofline"] = _pick(result.get("roofline", {}), ("bound", "kernel", "achieved", "peak", "unit", "frac", "ms", "traffic",
                                                       "traffic_over_algorithmic", "algorithmic_bytes_per_launch",
                                                       "algorithmic_bytes_per_frame", "ms_last_10_launches", "kernel_ms_avg"))
    if "frac_uses" in result.get("roofline", {}):
        c["roofline"]["frac_uses"] = "kernel_ms_avg"  # (= ms: HIP events over the timed launches; ms_last_10_launches is the settled figure)
    if isinstance(result.get("roofline_valu"), dict):
        c["roofline"]["valu_busy_frac"] = result["roofline_valu"].get("frac")
    if isinstance(result.get("roofline_pose_optimize"), dict):
        c["roofline_pose_optimize"] = _pick(result["roofline_pose_optimize"], ("kernel", "achieved", "frac", "ms", "algorithmic_bytes_per_frame"))
    f64 = result.get("f64_partials")
    if isinstance(f64, dict) and isinstance(f64.get("roofline"), dict):
        # the reference-width build of the same kernel (-DSIA_F64_PARTIALS), measured in the same run: its own line
        c["roofline_f64_build"] = dict(_pick(f64["roofline"], ("achieved", "peak", "frac", "ms", "traffic", "traffic_over_algorithmic")),
                                       value=f64.get("frames_per_s"), unit="frames/s",
                                       **({"value_sparse_align_only": f64["frames_per_s_sparse_align_only"]} if f64.get("frames_per_s_sparse_align_only") else {}))
    if "cpu_baseline" in result:
        c["cpu_baseline"] = _pick(result["cpu_baseline"], ("value", "unit", "cores", "kind", "sample_short", "cpu_model", "value_release_flags",
                                                           "value_sparse_align_only", "pose_optimize_us_per_frame",
                                                           "value_best_threads", "value_release_flags_best_threads", "best_threads",
                                                           "host_logical_cpus", "skipped"))
        if "sample_short" in c["cpu_baseline"]:
            c["cpu_baseline"]["sample"] = c["cpu_baseline"].pop("sample_short")
    if "parity" in result:
        c["parity"] = _pick(result["parity"], ("frames_compared", "se3_lognorm_max", "se3_lognorm_median", "ate_rmse_vs_cpu_m",
                                               "same_iteration_counts_frac", "against", "refined_pose_se3_lognorm_max"))
    legs = {}
    ft = result.get("full_track")
    if isinstance(ft, dict):
        legs["full_track"] = _pick(ft, ("ms_per_step", "frames_per_s", "skipped"))
        if isinstance(ft.get("stages_ms"), dict):
            legs["full_track"]["stages_ms"] = _num({k: v for k, v in ft["stages_ms"].items() if v >= 0.05}, 4)
        if isinstance(ft.get("mapper_on_its_own_stream"), dict) and "ms_per_step" in ft["mapper_on_its_own_stream"]:
            legs["full_track"]["ms_per_step_mapper_on_its_own_stream"] = _num(ft["mapper_on_its_own_stream"]["ms_per_step"], 4)
        if isinstance(ft.get("rooflines"), dict):
            legs["full_track"]["frac"] = _num({k: v.get("frac") for k, v in ft["rooflines"].items() if isinstance(v, dict)}, 3)
        if isinstance(ft.get("kernels"), dict):  # per kernel: [rocprof ms per step, fraction of the HBM roofline by compulsory bytes]
            short = lambda k: k.replace("find_match_direct/", "fm/").replace("update_seeds/", "us/").replace("pose_optimize/", "po/").replace("_kernel", "")
            legs["full_track"]["kernels_ms_frac"] = {short(k): [_num(v.get("ms"), 3), _num(v.get("frac"), 3)] for k, v in ft["kernels"].items()
                                                     if isinstance(v, dict) and (v.get("ms") or 0) >= 0.05}
    for key, keys in (("noise_sigma2", ("frames_per_s", "ms_per_step")),
                      ("f64_partials", ("frames_per_s", "slowdown")),
                      ("k0_pyramid", ("ms", "achieved", "frac")),
                      ("config3_xga5_b64", ("frames_per_s", "ms_per_step", "frames_per_s_at_batch_1024")),
                      ("stream_replay", ("frames_per_s", "host_link_GBs", "overlap_frac")),
                      ("full_track_long_scan", ("ms_per_step", "scanned_positions_per_seed", "epi_scan_ns_per_position"))):
        if isinstance(result.get(key), dict):
            legs[key] = _pick(result[key], keys + ("skipped",))
    ds = result.get("dropin_sequence")
    if isinstance(ds, dict):
        legs["dropin_sequence"] = _pick(ds, ("frames", "keyframes", "first_frame_with_a_different_decision", "se3_lognorm_max_before_it",
                                             "first_frame_with_a_different_tracking_decision", "se3_lognorm_max_before_the_first_tracking_difference",
                                             "ate_rmse_vs_cpu_m", "ate_rmse_vs_ground_truth_m", "skipped"))
        lw = ds.get("median_ms_per_frame_hip_dropin_list_walking_reprojector")
        if isinstance(lw, dict) and "tot_time" in lw:
            legs["dropin_sequence"]["ms_hip_map_mirror_off"] = lw["tot_time"]
            legs["dropin_sequence"]["map_mirror_same_trajectory"] = lw.get("trajectory_identical_to_the_mirror_path")
        wp = ds.get("median_ms_per_frame_hip_dropin_with_the_host_pyramid_built")
        if isinstance(wp, dict) and "tot_time" in wp:
            legs["dropin_sequence"]["ms_hip_host_pyramid_built"] = wp["tot_time"]
        for k_out, k_in in (("ms_cpu_ref", "median_ms_per_frame_cpu_reference"), ("ms_hip", "median_ms_per_frame_hip_dropin"),
                            ("ms_hip_deferred_mapper", "median_ms_per_frame_hip_dropin_deferred_mapper")):
            if isinstance(ds.get(k_in), dict):
                legs["dropin_sequence"][k_out] = ds[k_in].get("tot_time")
        op = ds.get("median_ms_per_frame_in_a_process_of_its_own")
        if isinstance(op, dict) and "hip_dropin" in op:
            legs["dropin_sequence"]["ms_hip_own_process"] = op["hip_dropin"]
            legs["dropin_sequence"]["ms_hip_deferred_mapper_own_process"] = op["hip_dropin_deferred_mapper"]
            if "hip_dropin_bound_to_the_gpus_numa_node" in op:
                legs["dropin_sequence"]["ms_hip_own_process_bound_to_gpu_numa_node"] = op["hip_dropin_bound_to_the_gpus_numa_node"]
                legs["dropin_sequence"]["ms_hip_deferred_mapper_own_process_bound_to_gpu_numa_node"] = op.get("hip_dropin_deferred_mapper_bound_to_the_gpus_numa_node")
        if isinstance(ds.get("early_mapper"), dict):
            legs["dropin_sequence"]["early_mapper"] = ds["early_mapper"]
        if isinstance(ds.get("map_size"), dict):
            legs["dropin_sequence"]["map_size"] = _pick(ds["map_size"], ("n_kfs", "n_candidates", "kf_points_in_frame", "trials", "matches"))
    rc = result.get("reference_cameras")
    if isinstance(rc, dict) and isinstance(rc.get("cameras"), dict):
        # per camera: [K1 M frames/s, K1 frac, K1 counter traffic / algorithmic, full-track ms per step, drop-in ms per frame]
        g = lambda d, *ks: (g(d.get(ks[0]), *ks[1:]) if len(ks) > 1 else d.get(ks[0])) if isinstance(d, dict) else None
        legs["reference_cameras"] = {"frames_per_step": rc.get("frames_per_step"),
                                     "keys": ["k1_frames_per_s", "k1_frac", "k1_traffic_over_algorithmic", "k1_gn_iterations_per_frame",
                                              "k1_time_per_iteration_over_undistorted_pinhole", "full_track_ms", "dropin_ms_per_frame"]}
        for name, r in rc["cameras"].items():
            legs["reference_cameras"][name] = _num([g(r, "sparse_align", "frames_per_s"), g(r, "sparse_align", "roofline", "frac"),
                                                    g(r, "sparse_align", "roofline", "traffic_over_algorithmic"),
                                                    g(r, "sparse_align", "mean_gn_iterations_per_frame"),
                                                    g(r, "sparse_align", "per_iteration_over_undistorted_pinhole"),
                                                    g(r, "full_track", "ms_per_step"), g(r, "dropin", "tot_time")], 4) if "skipped" not in r else "skipped"
    for key in ("gather", "stages_ms"):  # small objects of the multi-GPU / --pipeline full runs: whole
        if isinstance(result.get(key), dict):
            c[key] = result[key]
    if isinstance(result.get("rig_replay"), dict):
        c["rig_replay"] = _pick(result["rig_replay"], ("cameras", "rig_frames_per_s", "us_per_frame_set", "us_per_frame_set_without_gather",
                                                        "gather_bytes_per_frame_set", "skipped"))
    if legs:
        c["legs"] = legs
    c["details"] = details_path
    c = _num(c)
    # never exceed the limit: drop the optional blocks, least important first
    order = ("stream_replay", "k0_pyramid", "align_plus_refine", "config3_xga5_b64", "noise_sigma2", "f64_partials", "full_track_long_scan",
             "reference_cameras", "dropin_sequence", "full_track")
    for victim in ("rig_replay", "legs", "stages_ms", "parity"):
        if len(json.dumps(c)) <= COMPACT_LIMIT:
            break
        if victim == "legs" and "legs" in c:
            for k in [k for k in order if k in c["legs"]] + [k for k in c["legs"] if k not in order]:
                if len(json.dumps(c)) <= COMPACT_LIMIT:
                    break
                c["legs"].pop(k, None)
        else:
            c.pop(victim, None)
    return c


def write_details(result: dict) -> str | None:
    """the full object: next to bench.py and, on a gpurun box, also under gpurun_out/ (which is what travels back)"""
    where = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if d != ROOT and not os.path.isdir(d):
                continue
            with open(os.path.join(d, DETAILS_FILE), "w") as f:
                json.dump(result, f)
            where = where or os.path.join(os.path.relpath(d, ROOT), DETAILS_FILE).replace("./", "")
        except OSError:
            pass
    return where


class MutedStderr:
    """fd 2 -> bench_stderr.log while the extra legs run: the reference's own translation units (oracle/_ref, the
    drop-in pipelines) log every frame and every DepthFilter construction / destruction to stderr (SVO_INFO_STREAM),
    hundreds of lines that pushed the result out of the driver's view in round 3.  Exceptions of a leg are reported in
    the result itself ("skipped"), so nothing is lost."""

    def __init__(self, path=os.path.join(ROOT, "bench_stderr.log")):
        self.path, self.saved = path, None

    def __enter__(self):
        try:
            flush_c_stdio()
            sys.stderr.flush()
            fd = os.open(self.path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
            self.saved = os.dup(2)
            os.dup2(fd, 2)
            os.close(fd)
        except OSError:
            self.saved = None
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            flush_c_stdio()
            sys.stderr.flush()
            os.dup2(self.saved, 2)
            os.close(self.saved)
            self.saved = None
        return False


# ---- launch plumbing -----------------------------------------------------------------------------
def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def flush_c_stdio() -> None:
    """fflush(NULL): libraries below us (librccl, the reference's logging in the drop-in leg) write through C stdio."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def spawn_command(n_gpus: int, argv: list[str], port: int) -> list[str]:
    """The launch line of the contract (one rank per GPU of one node, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def spawn_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` from a plain shell: re-launch under torch.distributed.run."""
    cmd = spawn_command(n_gpus, sys.argv[1:], free_port())
    if os.environ.get("SVO_BENCH_DRY_SPAWN") == "1":
        print(json.dumps({"spawn": cmd}))
        return 0
    return subprocess.call(cmd)


class Events:
    """HIP-event timing on torch's current stream through the C ABI (the stream the kernels of
    libsvo_hip.so are enqueued on)."""

    def __init__(self, lib, dev):
        self.lib, self.dev = lib, dev

    def stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def mark(self):
        e = C.c_void_p()
        capi.check(self.lib.svo_hip_event_create(C.byref(e)))
        self.lib.svo_hip_event_record(e, self.stream())
        return e

    def ms(self, e0, e1) -> float:
        m = C.c_float()
        capi.check(self.lib.svo_hip_event_elapsed_ms(e0, e1, C.byref(m)))
        return m.value

    def time(self, fn, reps: int, warmup: int = 1) -> float:
        """mean milliseconds of fn() over `reps` back-to-back runs"""
        for _ in range(warmup):
            fn()
        marks = []
        for _ in range(reps):
            e0 = self.mark()
            fn()
            marks.append((e0, self.mark()))
        out = float(np.mean([self.ms(a, b) for a, b in marks]))
        for a, b in marks:
            self.lib.svo_hip_event_destroy(a)
            self.lib.svo_hip_event_destroy(b)
        return out


class Workload:
    """Synthetic replay sequence of B+1 frames on one GPU: problem b = (frame b -> frame b+1)."""

    def __init__(self, name: str, B: int, dev, rank: int = 0, noise: float = 0.0, images: torch.Tensor | None = None,
                 T_gt: np.ndarray | None = None, n_patches: int | None = None, cam=None):
        (self.width, self.height, self.focal, self.n_levels, self.max_level, self.min_level, self.n_patches,
         margin, cell) = WORKLOADS[name]
        if cam is not None:  # one of the reference's own cameras (REFERENCE_CAMERAS): its image size and model
            self.width, self.height, self.focal = cam.width, cam.height, cam.fx
        if n_patches is not None:  # side measurements only (config3_leg's latency floor), never the headline
            self.n_patches = n_patches
        if os.environ.get("SVO_BENCH_PATCHES"):  # kernel experiments only (scripts/): NOT the configuration the metric names
            self.n_patches = int(os.environ["SVO_BENCH_PATCHES"])
        self.name, self.B, self.dev, self.noise = name, B, dev, noise
        w, h, f = self.width, self.height, self.focal
        self.cam = synth.Camera(w, h, f, f, w / 2.0, h / 2.0) if cam is None else cam
        self.T_gt = synth.make_trajectory(B + 1, seed=12345 + rank) if T_gt is None else T_gt
        if images is None:
            images = synth.render(synth.make_texture(seed=12345), self.T_gt, self.cam, device=dev, chunk=32 if w <= 800 else 8)
        self.clean_images = images
        if noise > 0:  # svo_ros/src/benchmark_node.cpp:166-176: N(0, sigma) on every pixel, saturated
            g = torch.Generator(device=dev).manual_seed(99 + rank)
            out = torch.empty_like(images)
            for i0 in range(0, images.shape[0], 1024):
                blk = images[i0:i0 + 1024].float()
                out[i0:i0 + 1024] = (blk + noise * torch.randn(blk.shape, generator=g, device=dev)).round().clamp(0, 255).to(torch.uint8)
            images = out
        self.images = images
        self.px_all = synth.select_features(self.clean_images[:B], self.n_patches, margin=margin, cell=cell)
        g = torch.Generator().manual_seed(777 + rank)
        self.px_all = (self.px_all + (torch.rand(self.px_all.shape, generator=g, dtype=torch.float64) - 0.5).to(dev)).contiguous()
        self.f_all, self.pos_all = synth.features_3d(self.T_gt[:B], self.cam, self.px_all)
        self.store = PyramidStore(w, h, self.n_levels, B + 1, device=dev)
        self.store.load_images(self.images)  # level 0 copy + K0 pyramid build (untimed here)
        self.T_ref_w = self.T_gt[:B]
        self.T_prior_w = self.T_ref_w.copy()  # constant-position prior, frame_handler_mono.cpp:132
        T_cr, xyz_ref = marshal_problem(self.T_ref_w, self.T_prior_w, self.f_all.cpu().numpy(), self.pos_all.cpu().numpy())
        tdev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        self.ref_slot = torch.arange(0, B, dtype=torch.int32, device=dev)
        self.cur_slot = torch.arange(1, B + 1, dtype=torch.int32, device=dev)
        self.n_t = torch.full((B,), self.n_patches, dtype=torch.int32, device=dev)
        self.xyz_t = tdev(xyz_ref, torch.float64)
        self.T_in = tdev(T_cr, torch.float64)

    def run_align(self, sia, out=None):
        return sia.run(self.store, self.cam, self.ref_slot, self.cur_slot, self.n_t, self.px_all, self.xyz_t, self.T_in, out=out)

    def align_stats(self, out) -> dict:
        n_tracked = out.n_tracked.cpu().numpy().astype(np.float64)
        iters = out.iters.cpu().numpy().astype(np.float64)
        alg = algorithmic_bytes(np.full(self.B, self.n_patches, dtype=np.float64), n_tracked, iters, self.max_level, self.min_level)
        T_est_w = se3.mul(out.T_cur_from_ref.cpu().numpy(), self.T_ref_w)
        return {"alg_bytes": alg, "iters": iters, "n_tracked": n_tracked, "T_est_w": T_est_w,
                "gt_err": se3.log_norm(T_est_w, self.T_gt[1:self.B + 1])}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16384, help="frames per step per GPU")
    ap.add_argument("--workload", default="vga4_n200_sparse_align", choices=sorted(WORKLOADS))
    ap.add_argument("--noise", type=float, default=0.0, help="image noise sigma (gray levels) of the headline workload")
    ap.add_argument("--camera", default="default", choices=["default", "pinhole_undistorted", "radtan", "atan"],
                    help="default: the workload's own undistorted pinhole; else one of the reference's calibrations "
                         "(svo_ros/param/camera_pinhole.yaml = radtan, camera_atan.yaml = atan, or camera_pinhole.yaml without its "
                         "distortion) -- the image size follows the camera (752x480)")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="frames timed on the host for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n-iter", type=int, default=30, help="Gauss-Newton iteration cap per level (30 in the pipeline)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step (all kernel launches of the pipeline) in a HIP graph and replay it: "
                         "small batches -- e.g. one frame per camera of a rig -- are launch-bound")
    ap.add_argument("--pipeline", default="refine", choices=["refine", "align", "full"],
                    help="what the TIMED step runs.  refine (default, the headline): the metric as worded -- SparseImgAlign "
                         "followed by the Gauss-Newton pose refinement (K1 + K4) on BASELINE configs[1]'s frames, K4 fed by the "
                         "matches K2 / K3 produced for them in set-up; align: SparseImgAlign only (configs[1] read literally; the "
                         "default line carries it as value_sparse_align_only); full: configs[2] -- the whole track is the step "
                         "(the default run reports it as the extra key full_track instead)")
    ap.add_argument("--extras", default="all",
                    help="comma list of the extra legs to run at N=1 (all, none, or any of: f64, refine, full, full_easy, noise, "
                         "config3, stream, rig, k0, dropin, pmc)")
    ap.add_argument("--k1-kernel", default="auto", choices=["auto", "workgroup"],
                    help="auto: svo_hip_sparse_align (the wave-per-frame kernel for batches >= 1024 of <= 192 patches, else the "
                         "workgroup-per-frame kernel); workgroup: always the workgroup-per-frame kernel")
    ap.add_argument("--full-line", action="store_true",
                    help="print the FULL result object as the last stdout line (scripts/); default: the compact summary "
                         f"(< {COMPACT_LIMIT} B) with the full object in {DETAILS_FILE}")
    ap.add_argument("--time-budget", type=float, default=330.0,
                    help="seconds since process start after which no further optional leg or counter pass is begun (0: no limit); "
                         "the default run takes about four minutes on an MI355X box and stays below this")
    ap.add_argument("--pmc-child", default="", help=argparse.SUPPRESS)
    ap.add_argument("--dump-result", default="", help=argparse.SUPPRESS)  # child of the f64_partials leg: poses + iteration counts
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))  # plain `python bench.py --gpus N`: one rank per GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible); "
                         "one process per GPU is the only supported mapping")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # SVO_BENCH_FORCE_DIST=1 runs the RCCL code path (process group, overlapped gather) even with a
    # single rank: the 1-GPU box can then exercise it (tests/test_bench_dist_gpu.py)
    use_dist = world > 1 or os.environ.get("SVO_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        # librccl announces itself through C stdio, which is block-buffered on a pipe and would otherwise come out
        # at process exit, AFTER the JSON line: create the communicator now and push the banner out now
        dist.barrier()
        flush_c_stdio()
    lib = capi.load()
    ev = Events(lib, dev)
    if args.extras == "all":
        extras = {"f64", "refine", "full", "full_easy", "long_scan", "noise", "config3", "stream", "rig", "k0", "dropin", "pmc", "cameras"}
    elif args.extras == "none":
        extras = set()
    else:
        extras = set(args.extras.split(","))
    if args.no_cpu_baseline:
        extras.discard("dropin")
    if args.pmc_child or world > 1:
        extras = set()

    B = args.batch
    t_gen = time.time()
    W = Workload(args.workload, B, dev, rank, noise=args.noise, cam=None if args.camera == "default" else reference_cameras()[args.camera])
    store = W.store
    sia = SparseImgAlign(W.max_level, W.min_level, args.n_iter)
    sia.kernel = args.k1_kernel
    out = sia.alloc_result(B, dev)
    # N>1: the only exchange is the gather of the [B,12] poses.  It is double-buffered and issued
    # asynchronously (RCCL's own stream) so that it overlaps the next step's kernels; the timed
    # region ends only after the last gather has completed.  SVO_BENCH_SYNC_GATHER=1: blocking gather.
    gather = None
    outs = [out]
    if use_dist:
        from rpg_svo_amd.dist import OverlappedPoseGather
        gather = OverlappedPoseGather(B, 12, torch.float64, dev)
        if os.environ.get("SVO_BENCH_SYNC_GATHER") != "1":
            outs = [sia.alloc_result(B, dev) for _ in range(gather.depth)]
        for k, o in enumerate(outs):
            o.T_cur_from_ref = gather._local[k]
        # set-up, not a step: the first collective on a buffer pays for RCCL's lazy channel / registration work;
        # run it once per buffer here so that a short --warmup does not leave it inside the timed region
        for k in range(gather.depth):
            gather.submit(k)
        gather.drain()
    full = FullTrack(W, dev, rank) if args.pipeline == "full" else None
    refine = RefineStep(W, sia, dev, rank) if args.pipeline == "refine" else None
    pos = None
    if refine is not None:
        # the step's result is the REFINED pose: with N > 1 that is what the ranks exchange
        pos = [refine.alloc_result(gather._local[k] if gather is not None else None) for k in range(len(outs))]
        if gather is not None:
            for o in outs:  # (K1's relative pose stays local)
                o.T_cur_from_ref = torch.empty(B, 12, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    marks = []
    marks4 = []
    graph = None
    counter = [0]

    def step_compute(timed: bool) -> None:
        k = counter[0] % len(outs)
        o = outs[k]
        if gather is not None:
            gather.local(counter[0] if len(outs) > 1 else 0)  # waits for the gather that last read this buffer
        e0 = ev.mark() if timed else None
        W.run_align(sia, out=o)
        e1 = ev.mark() if timed else None
        if timed:
            marks.append((e0, e1))
        if refine is not None:
            refine.run(o.T_cur_from_ref, pos[k])
            if timed:
                marks4.append((e1, ev.mark()))
        if full is not None:
            full.step(o.T_cur_from_ref, ev if timed else None)

    def step(timed: bool) -> None:
        if graph is not None:
            e0 = ev.mark() if timed else None
            graph.replay()
            if timed:
                marks.append((e0, ev.mark()))
        else:
            step_compute(timed)
        if gather is not None:  # RCCL gather of the SE(3) results (the only exchange step)
            if len(outs) > 1:
                gather.submit(counter[0])
            else:
                gather.submit(0)
                gather.result(0)
        counter[0] += 1

    for _ in range(args.warmup):
        step(False)
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()
    if args.graph and use_dist:
        raise SystemExit("--graph replays fixed buffers; combine it with the overlapped gather only at --gpus 1")
    if args.graph:
        # torch's graph object is the capture front end (private allocator pool for the tensors the
        # host mirrors create); what gets captured are the launches libsvo_hip.so enqueues on the
        # capture stream.  C++ hosts use svo_hip_graph_begin_capture / end_capture / launch.
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            step_compute(False)
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.cuda.graph(graph):
            step_compute(False)
        torch.cuda.synchronize()
    # ---- the timed region: exactly K steps between barrier + synchronize on both sides ----
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(True)
    if gather is not None:
        gather.drain()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gather_stats = None
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        gather_stats = time_gather(gather, dist, dev, world, B, len(outs) > 1)

    # per-launch kernel duration from HIP events on the launch stream
    kernel_ms = float(np.mean([ev.ms(a, b) for a, b in marks])) if marks else float("nan")

    rig = None
    if ("rig" in extras or use_dist) and not args.pmc_child:
        try:
            rig = rig_replay(ev, dev, dist if use_dist else None, world, rank)
        except Exception as e:  # never lose the headline line to an extra leg
            rig = {"skipped": repr(e)}

    if rank != 0:
        dist.destroy_process_group()
        flush_c_stdio()
        return

    out = outs[(counter[0] - 1) % len(outs)]  # the result block of the last step
    st = W.align_stats(out)
    iters, n_tracked, alg_bytes = st["iters"], st["n_tracked"], st["alg_bytes"]
    k4_ms = float(np.mean([ev.ms(a, b) for a, b in marks4])) if marks4 else None
    po_last = pos[(counter[0] - 1) % len(outs)] if refine is not None else None
    workload_name = args.workload
    if full is not None:
        workload_name = args.workload.replace("sparse_align", "full_track")
    elif refine is not None:
        workload_name = args.workload.replace("sparse_align", "sparse_align_plus_pose_refine")

    result = {
        "metric": "frames/sec sparse-align+pose-refine (VGA, 4 pyr lvls); ATE vs CPU ref",
        "value": world * B * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # what the arithmetic is carried out in (the reference: f32 pixels / chi2, f64 everything else).  `value` is the DEFAULT
        # build's; the same step on the build that is f64 wherever the reference is: roofline_f64_build.value
        "dtype": ("K1: " if refine is not None else "") +
                 "f32 pixels, residuals, chi2 and per-pixel Jacobian products (16-term patch sums); f64 projection, per-patch "
                 "Jacobian rows, per-lane Jres/H partials (tree-reduced per wave; reference: sequential), cross-wave sums, 6x6 "
                 "solve and pose; f32 series for SE3::exp (reference: f64 sin/cos) -- see f64_partials / roofline_f64_build for "
                 "the build that is f64 throughout" + ("; K4 (pose refine): f64 throughout, Tukey weight / MAD scale in f32 as vikit has them" if refine is not None else ""),
        "dtype_short": ("value = default build: K1 f32 pixels/residuals/chi2/per-pixel products, f64 Jacobian rows/partials/reductions/"
                        "projection/solve/pose; K4 f64 (reference-width build: roofline_f64_build.value)" if refine is not None else
                        "f32 pixels/residuals/chi2/per-pixel products, f64 Jacobian rows/partials/reductions/projection/solve/pose"),
        "data": "synthetic",
        "config": {
            "workload": workload_name,
            "image": f"{W.width}x{W.height}", "pyr_levels": W.n_levels,
            "schedule": f"levels {W.max_level}->{W.min_level}", "patches_per_frame": W.n_patches,
            "frames_per_step_per_gpu": B, "n_iter_cap": args.n_iter, "image_noise_sigma": args.noise,
            "timed_region": "svo_hip_sparse_align (SparseImgAlign::run incl. its Gauss-Newton pose solve) over the batch"
                            + (" + svo_hip_compose_poses + svo_hip_pose_optimize (pose_optimizer::optimizeGaussNewton from K1's pose of the "
                               "same step, on the matches K2 / K3 produced for these frames in set-up)" if refine is not None else "")
                            + (" + the rest of the track" if full is not None else "")
                            + (" + RCCL all_gather of the poses" if use_dist else ""),
            "parallelism": f"frames sharded 1 rank/GPU x{world}" + (", RCCL all_gather of poses, double-buffered and overlapped with the next step" if use_dist else ""),
            "hip_graph": bool(args.graph), "k1_kernel": args.k1_kernel,
            # the first dozen launches of a process run while the clocks settle (1.31 -> 1.14 ms);
            # which launches of K1 the timed region holds
            "k1_launches_before_timed_region": args.warmup + (2 if args.graph else 0),
            "timed_launches": f"launches {args.warmup + 1}..{args.warmup + args.steps} of this process",
            "mean_gn_iterations_per_frame": float(iters.sum(1).mean()),
            "mean_tracked_patches": float(n_tracked.mean()),
            "median_pose_error_vs_gt": float(np.median(st["gt_err"])),
        },
        "roofline": roofline(("sia_wave_kernel" if args.k1_kernel == "auto" and W.n_patches <= 192 and B >= 1024 else "sia_kernel") + " (svo_hip_sparse_align)", alg_bytes, kernel_ms, traffic=None, kernel_ms_avg=kernel_ms,
                             algorithmic_bytes_per_frame=alg_bytes / B,
                             frac_uses="kernel_ms_avg: HIP events on the launch stream around every timed launch of the kernel, averaged "
                                       "(the launches sit inside the clock ramp of a fresh process; ms_last_10_launches is the settled figure)",
                             ms_last_10_launches=float(np.mean([ev.ms(a, b) for a, b in marks[-10:]])) if marks else None,
                             # iterations/s and per-iteration time of the batch
                             gn_iterations_per_s=float(iters.sum()) / (kernel_ms * 1e-3),
                             us_per_gn_iteration_of_the_batch=kernel_ms * 1e3 / max(float(iters.sum(1).mean()), 1e-9)),
        "setup_s": t_gen,
    }
    if refine is not None:
        T_ref_w = po_last.T_f_w.cpu().numpy()
        result["stages_ms"] = {"sparse_align": kernel_ms, "compose_plus_pose_optimize": k4_ms}
        # configs[1] read literally ("SparseImgAlign only"): the same launches, K1 on its own
        result["value_sparse_align_only"] = world * B / (kernel_ms * 1e-3)
        result["config"].update(
            matches="K2 / K3 output of the representative full-track workload for the same frames (set-up, untimed)",
            match_trials_per_frame=refine.trials_per_frame, matches_per_frame=refine.matches_per_frame,
            pose_refine_obs_after_pruning=float(po_last.stats[:, 3].mean().item()),
            median_pose_error_vs_gt_after_refine=float(np.median(se3.log_norm(T_ref_w, W.T_gt[1:B + 1]))))
        result["roofline_pose_optimize"] = roofline("compose_kernel + pose_opt_wave_kernel (svo_hip_pose_optimize)",
                                                    refine.algorithmic_bytes(), k4_ms,
                                                    algorithmic_bytes_per_frame=refine.algorithmic_bytes() / B)
    if args.dump_result:
        np.savez(args.dump_result, T_est_w=st["T_est_w"], iters=out.iters.cpu().numpy())
    if args.pmc_child:  # child of the PMC leg: nothing else is needed from this process
        print(json.dumps(result))
        return
    # from here on only extra legs run: their stderr chatter (the reference's SVO_INFO_STREAM in oracle/_ref and the
    # drop-in pipelines, rocprofv3 children) goes to bench_stderr.log, and stays there until the process exits
    # (destructors of the reference's objects log at exit too)
    mute = MutedStderr()
    mute.__enter__()
    try:
        _extras_and_print(args, result, extras, full, ev, W, sia, out, st, store, lib, dev, rank, world, use_dist, dist,
                          gather_stats, rig, kernel_ms, alg_bytes, B, refine, po_last)
    except BaseException:
        mute.__exit__()
        raise


def _extras_and_print(args, result, extras, full, ev, W, sia, out, st, store, lib, dev, rank, world, use_dist, dist,
                      gather_stats, rig, kernel_ms, alg_bytes, B, refine=None, po_last=None) -> None:
    if full is not None:
        d = full.describe()
        d.pop("_T_refined")
        result["config"].update(d)
        result["stages_ms"] = full.stage_ms(ev) if not args.graph else {}
        result["stages_ms"]["sparse_align" if not args.graph else "whole_graph"] = kernel_ms
    if gather_stats is not None:
        result["gather"] = gather_stats
    if rig is not None:
        result["rig_replay"] = rig

    def leg(name, fn):
        if name not in extras:
            return
        if time_left(args.time_budget) < 10.0:
            result[EXTRA_KEYS[name]] = {"skipped": f"time budget ({args.time_budget:.0f} s since process start) used up"}
            return
        t = time.time()
        try:
            r = fn()
        except Exception as e:  # an extra leg must never cost the headline line
            r = {"skipped": repr(e)}
        if isinstance(r, dict):
            r["leg_seconds"] = time.time() - t
        result[EXTRA_KEYS[name]] = r

    if not args.no_cpu_baseline and world == 1:
        try:
            result["cpu_baseline"] = cpu_baseline(args, W, st["T_est_w"], result, out.iters.cpu().numpy(), refine, po_last)
        except Exception as e:
            result["cpu_baseline"] = {"skipped": repr(e)}
    leg("f64", lambda: f64_partials_leg(args, st["T_est_w"], out.iters.cpu().numpy(), result))
    leg("refine", lambda: align_plus_refine(W, sia, ev, dev, args.steps))
    leg("full", lambda: full_track_leg(W, sia, ev, dev, rank, lib, not args.no_cpu_baseline))
    leg("full_easy", lambda: full_track_leg(W, sia, ev, dev, rank, lib, False, steps=3, mode="easy"))
    leg("long_scan", lambda: long_scan_leg(W, ev, dev, lib))
    leg("k0", lambda: pyramid_roofline(ev, store, W.images))
    if args.noise == 0:
        leg("noise", lambda: noise_leg(W, sia, ev, dev, rank, args.steps))
    leg("config3", lambda: config3_leg(ev, dev, rank, args.n_iter, not args.no_cpu_baseline))
    leg("stream", lambda: stream_replay_leg(W, sia, ev, dev))
    leg("dropin", dropin_sequence)   # (before the counter passes, which are what gives way to --time-budget)
    leg("cameras", lambda: reference_cameras_leg(args, ev, dev, rank, lib))
    if "pmc" in extras:
        t = time.time()
        want_full = "full" in extras and isinstance(result.get("full_track"), dict) and "rooflines" in result["full_track"]
        try:
            pm = pmc_leg(args, kernel_ms, reserve_s=FULL_TRACK_PMC_RESERVE_S if want_full else 0.0)
        except Exception as e:
            pm = {"skipped": repr(e)}
        pm["leg_seconds"] = time.time() - t
        if "traffic_bytes_per_launch" in pm:
            result["roofline"]["traffic"] = pm["traffic_bytes_per_launch"]
            result["roofline"]["traffic_over_algorithmic"] = pm["traffic_bytes_per_launch"] / alg_bytes
            try:
                floor = line_floor_bytes(W) * B
                result["roofline"]["cache_line_floor_bytes_per_launch"] = floor
                result["roofline"]["traffic_over_cache_line_floor"] = pm["traffic_bytes_per_launch"] / floor
            except Exception as e:
                result["roofline"]["cache_line_floor_bytes_per_launch"] = repr(e)
        if "roofline_valu" in pm:
            result["roofline_valu"] = pm.pop("roofline_valu")
        if want_full:
            try:
                if time_left(args.time_budget) < FULL_TRACK_PMC_RESERVE_S - 30.0:
                    pf = {"skipped": f"time budget ({args.time_budget:.0f} s since process start)"}
                else:
                    pf = pmc_full_track_leg(args)
            except Exception as e:
                pf = {"skipped": repr(e)}
            pm["full_track"] = pf
            try:
                result["full_track"]["kernels"] = full_track_kernel_rooflines(result["full_track"], pf, B, W.n_patches)
            except Exception as e:
                result["full_track"]["kernels"] = {"skipped": repr(e)}
            for stage, st_ in pf.get("stages", {}).items():  # next to the stage's algorithmic bytes
                rl_ = result["full_track"]["rooflines"].get(stage)
                if rl_ and st_["traffic_bytes_per_step"] == st_["traffic_bytes_per_step"]:
                    rl_["traffic"] = st_["traffic_bytes_per_step"]
                    rl_["traffic_over_algorithmic"] = st_["traffic_bytes_per_step"] / rl_["algorithmic_bytes_per_launch"]
                    rl_["traffic_by_kernel"] = st_["by_kernel"]
            pm["leg_seconds"] = time.time() - t
        result["pmc"] = pm
        if isinstance(result.get("reference_cameras"), dict) and "cameras" in result["reference_cameras"]:
            t = time.time()
            reference_cameras_traffic(args, result["reference_cameras"])
            result["reference_cameras"]["traffic_seconds"] = time.time() - t
    if use_dist:
        dist.destroy_process_group()  # (the other ranks have left already)
    flush_c_stdio()
    # rank 0: the full object goes to bench_details.json, the LAST stdout line is its compact summary (< 4 KB)
    where = write_details(result)
    print(json.dumps(result if args.full_line else compact_line(result, where)), flush=True)


def time_gather(gather, dist, dev, world, B, overlapped) -> dict:
    """the exchange step on its own (outside the timed region): blocking all-gathers of one pose block"""
    reps = 20
    for _ in range(3):
        gather.submit(0)
        gather.result(0)
    torch.cuda.synchronize()
    dist.barrier()
    tg = time.perf_counter()
    for _ in range(reps):
        gather.submit(0)
        gather.result(0)
    torch.cuda.synchronize()
    tg = torch.tensor([(time.perf_counter() - tg) / reps], dtype=torch.float64, device=dev)
    dist.all_reduce(tg, op=dist.ReduceOp.MAX)
    return {"collective": "all_gather_into_tensor (RCCL)", "bytes_per_rank_per_step": int(B * 12 * 8),
            "bytes_gathered_per_step": int(world * B * 12 * 8), "ms_blocking_avg": float(tg.item()) * 1e3,
            "overlapped_in_timed_region": overlapped}


# ---- extra legs ------------------------------------------------------------------------------------
def rig_replay(ev, dev, dist, world, rank, n_frames: int = 200) -> dict:
    """BASELINE configs[4] shape: every rank is one camera of a rig (752x480, the reference's default
    schedule: 5 levels, 4 -> 2, 120 patches) tracking ITS OWN stream frame by frame -- frame k+1 needs
    frame k's pose (frame_handler_mono.cpp:85,132), so the per-camera batch is 1 -- with one all-gather
    of the SE(3) results per frame set.  Latency-bound by construction; reported as rig frames/s
    (all cameras) and the per-frame split."""
    W = Workload("svo_default_752_l4to2_n120", n_frames, dev, rank + 100)
    sia = SparseImgAlign(W.max_level, W.min_level, 30)
    out = sia.alloc_result(1, dev)
    allT = torch.zeros(max(world, 1), 12, dtype=torch.float64, device=dev)
    views = [(W.ref_slot[i:i + 1], W.cur_slot[i:i + 1], W.n_t[i:i + 1], W.px_all[i:i + 1], W.xyz_t[i:i + 1], W.T_in[i:i + 1])
             for i in range(n_frames)]

    def frame(i, do_gather=True):
        r, c, n, px, xyz, T = views[i]
        sia.run(W.store, W.cam, r, c, n, px, xyz, T, out=out)
        if dist is not None and do_gather:
            dist.all_gather_into_tensor(allT, out.T_cur_from_ref)
        # the host consumes the pose before the next frame is handed in (live tracking)
        torch.cuda.current_stream(dev).synchronize()

    for i in range(10):
        frame(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n_frames):
        frame(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for i in range(n_frames):
        frame(i, do_gather=False)
    torch.cuda.synchronize()
    dt_nogather = time.perf_counter() - t1
    if dist is not None:
        tt = torch.tensor([dt, dt_nogather], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_nogather = float(tt[0].item()), float(tt[1].item())
    return {"workload": "svo_default_752_l4to2_n120, one camera stream per rank, batch 1 per camera, pose gather per frame set",
            "cameras": world, "frames_per_camera": n_frames, "rig_frames_per_s": world * n_frames / dt,
            "us_per_frame_set": dt / n_frames * 1e6, "us_per_frame_set_without_gather": dt_nogather / n_frames * 1e6,
            "gather_bytes_per_frame_set": int(world * 96) if dist is not None else 0}


def refine_inputs(W: Workload, dev, rank: int, px_sigma: float = 0.3):
    """Synthetic matches for pose refinement: the reference frame's points observed in the current
    frame at their true projection + px_sigma pixels (what findMatchDirect would deliver)."""
    B, N, cam = W.B, W.n_patches, W.cam
    T = torch.as_tensor(W.T_gt[1:B + 1], dtype=torch.float64, device=dev)
    p = (T[:, :9].reshape(B, 1, 3, 3) @ W.pos_all[..., None])[..., 0] + T[:, None, 9:]
    g = torch.Generator().manual_seed(31 + rank)
    px = torch.stack([cam.fx * p[..., 0] / p[..., 2] + cam.cx, cam.fy * p[..., 1] / p[..., 2] + cam.cy], -1)
    px = px + px_sigma * torch.randn(px.shape, generator=g, dtype=torch.float64).to(dev)
    d = torch.stack([(px[..., 0] - cam.cx) / cam.fx, (px[..., 1] - cam.cy) / cam.fy, torch.ones_like(px[..., 0])], -1)
    f_cur = (d / d.norm(dim=-1, keepdim=True)).contiguous()
    level = torch.zeros(B, N, dtype=torch.int32, device=dev)
    has = torch.ones(B, N, dtype=torch.uint8, device=dev)
    return f_cur, level, has


class RefineStep:
    """The metric as it is worded -- "sparse-align + pose-refine" -- as the timed step: K1 (SparseImgAlign::run) followed
    by pose_optimizer::optimizeGaussNewton (K4) started from K1's pose of the same step (frame_handler_mono.cpp:137-165).
    K4's observations are the matches the pipeline itself produces for these frames: in set-up (untimed) the
    representative full-track workload (FullTrack) runs K1 -> Reprojector::reprojectPoint -> Matcher::findMatchDirect
    (K2 + K3) -> cam2world once; what it matched (~170 of ~200 trials per frame, 35 % of the points with grossly wrong
    depth, refined pixels off by what alignment leaves) is what every timed step refines on -- not synthetic matches at
    the true projection."""

    def __init__(self, W: Workload, sia, dev, rank: int):
        from rpg_svo_amd import tracking as tr
        self.tr, self.W, self.dev = tr, W, dev
        B, N = W.B, W.n_patches
        full = FullTrack(W, dev, rank, with_seeds=False)
        out = sia.alloc_result(B, dev)
        W.run_align(sia, out=out)
        m, _ = full.match_stage(out.T_cur_from_ref)
        torch.cuda.synchronize()
        self.f_new = full.f_new.view(B, N, 3)
        self.level = m.search_level.view(B, N).clone()
        self.okb = full.okb.clone()
        self.pt_pos = full.pt_pos.view(B, N, 3)
        self.n, self.T_ref = full.n, full.T_ref
        self.T_cur = torch.empty(B, 12, dtype=torch.float64, device=dev)
        self.trials_per_frame = float(full.in_cur.float().sum().item() / B)
        self.matches_per_frame = float(self.okb.float().sum().item() / B)
        self.n_obs = float(self.okb.float().sum().item())
        del full, out, m
        torch.cuda.empty_cache()

    def alloc_result(self, T_f_w: torch.Tensor | None = None):
        B, N, dev = self.W.B, self.W.n_patches, self.dev
        return self.tr.PoseOptResult(torch.empty(B, 12, dtype=torch.float64, device=dev) if T_f_w is None else T_f_w,
                                     torch.zeros(B, 36, dtype=torch.float64, device=dev), torch.zeros(B, 4, dtype=torch.float64, device=dev),
                                     torch.zeros(B, dtype=torch.int32, device=dev), torch.empty(B, N, dtype=torch.uint8, device=dev))

    def run(self, T_cur_from_ref, po):
        tr = self.tr
        tr.compose_poses(T_cur_from_ref, self.T_ref, out=self.T_cur)
        tr.optimize_gauss_newton(self.W.cam, self.n, self.f_new, self.level, self.pt_pos, self.okb, self.T_cur, 2.0, 10, out=po)

    def algorithmic_bytes(self) -> float:
        """52 B per observation (f, level, pos) + the flag byte of every slot + 416 B of pose / covariance / stats"""
        return self.n_obs * 52.0 + self.W.B * (self.W.n_patches * 1.0 + 416.0)


def align_plus_refine(W: Workload, sia, ev: Events, dev, steps: int) -> dict:
    """The metric's "+pose-refine" read literally: SparseImgAlign followed by
    pose_optimizer::optimizeGaussNewton (K1 -> compose -> K4) on the headline batch."""
    from rpg_svo_amd import tracking as tr
    B, N = W.B, W.n_patches
    f_cur, level, has = refine_inputs(W, dev, 0)
    T_ref = torch.as_tensor(W.T_ref_w, dtype=torch.float64, device=dev)
    T_cur = torch.empty(B, 12, dtype=torch.float64, device=dev)
    out = sia.alloc_result(B, dev)
    po = tr.PoseOptResult(torch.empty(B, 12, dtype=torch.float64, device=dev), torch.zeros(B, 36, dtype=torch.float64, device=dev),
                          torch.zeros(B, 4, dtype=torch.float64, device=dev), torch.zeros(B, dtype=torch.int32, device=dev),
                          torch.empty(B, N, dtype=torch.uint8, device=dev))

    def refine():
        tr.compose_poses(out.T_cur_from_ref, T_ref, out=T_cur)
        tr.optimize_gauss_newton(W.cam, W.n_t, f_cur, level, W.pos_all, has, T_cur, 2.0, 10, out=po)

    def both():
        W.run_align(sia, out=out)
        refine()

    ms_both = ev.time(both, steps, warmup=2)
    ms_refine = ev.time(refine, steps, warmup=1)
    torch.cuda.synchronize()
    err = se3.log_norm(po.T_f_w.cpu().numpy(), W.T_gt[1:B + 1])
    k4_bytes = float(B) * (N * 52 + 416)
    return {"frames_per_s": B / ms_both * 1e3, "ms_per_step": ms_both, "ms_pose_optimize": ms_refine,
            "frames_per_step": B, "matches": f"{N} synthetic matches per frame at the true projection + 0.3 px",
            "median_pose_error_vs_gt_after_refine": float(np.median(err)),
            "mean_obs_after_pruning": float(po.stats[:, 3].mean().item()),
            "roofline_pose_optimize": roofline("pose_opt_wave_kernel (svo_hip_pose_optimize)", k4_bytes, ms_refine,
                                               algorithmic_bytes_per_frame=N * 52 + 416)}


def noise_leg(W: Workload, sia, ev: Events, dev, rank: int, steps: int) -> dict:
    """The headline workload with N(0, 2) image noise (svo_ros/src/benchmark_node.cpp:166-176): more
    Gauss-Newton evaluations per frame, same kernel."""
    Wn = Workload(W.name, W.B, dev, rank, noise=2.0, images=W.clean_images, T_gt=W.T_gt)
    out = sia.alloc_result(W.B, dev)
    ms = ev.time(lambda: Wn.run_align(sia, out=out), max(steps // 2, 3), warmup=2)
    torch.cuda.synchronize()
    st = Wn.align_stats(out)
    r = {"image_noise_sigma": 2.0, "frames_per_s": W.B / ms * 1e3, "ms_per_step": ms,
         "mean_gn_iterations_per_frame": float(st["iters"].sum(1).mean()), "mean_tracked_patches": float(st["n_tracked"].mean()),
         "median_pose_error_vs_gt": float(np.median(st["gt_err"])),
         "roofline": roofline("sia_kernel", st["alg_bytes"], ms, gn_iterations_per_s=float(st["iters"].sum()) / (ms * 1e-3))}
    del Wn
    torch.cuda.empty_cache()
    return r


def config3_leg(ev: Events, dev, rank: int, n_iter: int, with_cpu: bool = True) -> dict:
    """BASELINE configs[3]: 1280x960, 5 levels (4 -> 0), 1000 patches, 64 frames at once.  A frame is one workgroup,
    so 64 frames occupy 64 of the 256 CUs and the step time IS the latency of one frame's alignment: `ms_per_step`
    is reported as that latency, the throughput of the configuration is the batch that fills the GPU
    (`frames_per_s_at_batch_1024`), and the reference's own code on the host cores runs beside both."""
    keep = {}

    def run(B, n_patches=None, reps=20):
        W = Workload("xga5_n1000_sparse_align", B, dev, rank + 7, n_patches=n_patches)
        sia = SparseImgAlign(W.max_level, W.min_level, n_iter)
        out = sia.alloc_result(W.B, dev)
        ms = ev.time(lambda: W.run_align(sia, out=out), reps, warmup=3)
        torch.cuda.synchronize()
        if B == 64 and n_patches is None:
            keep["W"], keep["out"] = W, out
        return ms, W.align_stats(out)
    ms, st = run(64)
    cpu = None
    if with_cpu:
        try:
            cpu = config3_cpu(keep["W"], st, keep["out"], n_iter)
        except Exception as e:
            cpu = {"skipped": repr(e)}
    keep.clear()
    ms_big, st_big = run(1024, reps=10)
    ms_q, st_q = run(64, n_patches=250)
    return {"workload": "xga5_n1000_sparse_align", "frames_per_step": 64, "frames_per_s": 64 / ms * 1e3, "ms_per_step": ms,
            "what_ms_per_step_is": "the latency of ONE 1000-patch frame (64 workgroups on 256 CUs run side by side)",
            "cpu_baseline": cpu,
            "mean_gn_iterations_per_frame": float(st["iters"].sum(1).mean()), "mean_tracked_patches": float(st["n_tracked"].mean()),
            "median_pose_error_vs_gt": float(np.median(st["gt_err"])),
            "roofline": roofline("sia_kernel", st["alg_bytes"], ms),
            # 64 frames occupy 64 of 256 CUs: the step time is the latency of ONE frame's alignment
            "frames_per_s_at_batch_1024": 1024 / ms_big * 1e3, "ms_per_step_at_batch_1024": ms_big,
            "roofline_at_batch_1024": roofline("sia_kernel", st_big["alg_bytes"], ms_big),
            # VERDICT r01 item 8 proposed splitting a frame over 4 workgroups with a global exchange of the sums:
            # a quarter of the patches per workgroup is this measurement, BEFORE any exchange cost
            "split4_latency_floor": {"patches_per_workgroup": 250, "ms_per_step": ms_q, "frames_per_s_upper_bound": 64 / ms_q * 1e3,
                                     "mean_gn_iterations_per_frame": float(st_q["iters"].sum(1).mean())}}


def config3_cpu(W: Workload, st: dict, out, n_iter: int) -> dict:
    """configs[3] on the host: the reference's own SparseImgAlign (oracle/_ref; the C port without it) on the same 64
    frames, one thread and all threads, and its agreement with the device result."""
    from oracle import pyoracle
    S = W.B
    pyrs = [pyoracle.create_img_pyramid(im, W.n_levels, pyoracle.HALFSAMPLE_AUTO) for im in W.images[:S + 1].cpu().numpy()]
    rs = np.arange(S, dtype=np.int32)
    nn = np.full(S, W.n_patches, dtype=np.int32)
    px, f, pos = W.px_all[:S].cpu().numpy(), W.f_all[:S].cpu().numpy(), W.pos_all[:S].cpu().numpy()
    hp = np.ones((S, W.n_patches), dtype=np.uint8)
    which = "ref" if pyoracle.ref_available() else "orc"
    cores = os.cpu_count() or 1

    def timed(k, threads):
        tm = {}
        t0 = time.perf_counter()
        T, r = pyoracle.sparse_img_align_batch(pyrs, rs[:k], rs[:k] + 1, W.cam, W.T_ref_w[:k], W.T_prior_w[:k], nn[:k], px[:k], f[:k],
                                               hp[:k], pos[:k], W.max_level, W.min_level, n_iter, n_threads=threads, which=which, timing=tm)
        return T, r, tm.get("run_seconds", time.perf_counter() - t0)
    _, _, t1 = timed(16, 1)
    T_cpu, res, tn = timed(S, min(cores, S))
    d = se3.log_norm(st["T_est_w"], T_cpu)
    it = out.iters.cpu().numpy()
    return {"kind": "reference" if which == "ref" else "port", "frames_per_s_1core": 16 / t1, "ms_per_frame_1core": t1 / 16 * 1e3,
            "frames_per_s_all_threads": S / tn, "threads": min(cores, S), "sample": f"{S} frames (16 for the 1-thread figure)",
            "parity": {"se3_lognorm_max": float(d.max()), "se3_lognorm_median": float(np.median(d)),
                       "same_iteration_counts_frac": float(np.mean([np.array_equal(r["iters"], x) for r, x in zip(res, it)]))}}


def stream_replay_leg(W: Workload, sia, ev: Events, dev, n_frames: int = 4096, chunk: int = 256) -> dict:
    """Image in -> pose out for an offline replay whose images live in HOST memory (second half): level-0
    images in pinned memory, copied H2D chunk by chunk on a copy stream into one of two packed staging buffers while
    the compute stream runs K0 (tiled level 0 + pyramid, one kernel) and K1 on the previous chunk.  The pyramid
    store is a ring of two chunks, so the frame pair across a chunk boundary finds both pyramids resident.  The
    bound is the host link: a VGA frame is 307 200 bytes."""
    n_frames = min(n_frames, W.B)
    chunk = min(chunk, n_frames // 2)
    n_chunks = n_frames // chunk
    n_frames = n_chunks * chunk
    h, w = W.height, W.width
    frame_bytes = h * w
    host = W.images[:n_frames].cpu().pin_memory()                      # [n, h, w] u8, page-locked
    staging = [torch.empty(chunk, h, w, dtype=torch.uint8, device=dev) for _ in range(2)]
    store = PyramidStore(w, h, W.n_levels, 2 * chunk, device=dev)
    copy_stream = torch.cuda.Stream(dev)
    comp = torch.cuda.current_stream(dev)
    out = sia.alloc_result(chunk, dev)
    # problem of frame k*chunk + i: reference = the frame before it (the other half of the ring for i = 0), current = itself;
    # features / priors of the frame pairs are the workload's own (device-resident, a few KB per frame)
    slots = torch.arange(chunk, dtype=torch.int32, device=dev)
    prob = []
    for k in range(n_chunks):
        base = (k & 1) * chunk
        cur = (slots + base).contiguous()
        ref = (cur - 1).clone()
        ref[0] = ((k + 1) & 1) * chunk + chunk - 1 if k > 0 else cur[0]  # frame 0 has no predecessor: aligned against itself
        j = (torch.arange(k * chunk, (k + 1) * chunk, device=dev) - 1).clamp(min=0)
        prob.append((ref.contiguous(), cur, W.px_all[j].contiguous(), W.xyz_t[j].contiguous(), W.T_in[j].contiguous()))
    n_t = W.n_t[:chunk].contiguous()
    torch.cuda.synchronize()

    def run(do_copy: bool, do_compute: bool) -> float:
        copied = [torch.cuda.Event() for _ in range(n_chunks)]
        consumed = [torch.cuda.Event() for _ in range(n_chunks)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_chunks):
            if do_copy:
                with torch.cuda.stream(copy_stream):
                    if k >= 2 and do_compute:
                        copy_stream.wait_event(consumed[k - 2])          # K0 of chunk k-2 has read this staging buffer
                    staging[k & 1].copy_(host[k * chunk:(k + 1) * chunk], non_blocking=True)
                    copied[k].record(copy_stream)
            if do_compute:
                if do_copy:
                    comp.wait_event(copied[k])
                store.load_images(staging[k & 1], first_slot=(k & 1) * chunk)  # K0: tiled level 0 + levels 1.. in one kernel
                consumed[k].record(comp)
                ref, cur, px, xyz, T_in = prob[k]
                sia.run(store, W.cam, ref, cur, n_t, px, xyz, T_in, out=out)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(True, True)  # warm-up (allocator, first-touch of the pinned pages)
    t_stream = min(run(True, True) for _ in range(2))
    t_copy = min(run(True, False) for _ in range(2))
    t_comp = min(run(False, True) for _ in range(2))
    gbps = n_frames * frame_bytes / t_copy / 1e9
    bound = gbps * 1e9 / frame_bytes
    res = {"frames": n_frames, "chunk_frames": chunk, "frames_per_s": n_frames / t_stream,
           "h2d_GBps": gbps, "pcie_bound_frames_per_s": bound, "frac_of_pcie_bound": (n_frames / t_stream) / bound,
           "seconds": {"streamed": t_stream, "copies_only": t_copy, "compute_only_K0_K1": t_comp},
           # 1 = the shorter of the two activities is hidden completely behind the longer one, 0 = they run back to back
           "overlap_frac": (t_copy + t_comp - t_stream) / min(t_copy, t_comp),
           "resident_frames_per_s_K0_K1": n_frames / t_comp,
           "what": "pinned host images -> H2D (copy stream, two staging buffers) -> K0 pyramid_fused_kernel -> K1 sia_kernel, per chunk; "
                   "poses stay on the device"}
    del host, staging, store
    torch.cuda.empty_cache()
    return res


def line_floor_bytes(W: Workload, n_sample: int = 64) -> float:
    """What the gathers of K1 cost at cache-line granularity, per frame: the DISTINCT 128-byte lines of a
    pyramid touched by the 7-row x 12-byte windows the kernel fetches -- as reference frame (its own
    features) and as current frame (the previous frame's features, projected) -- on every level of the
    schedule, in the store's own layout (capi.pyr_px_offset: 16 x 8 pixel tiles of one line each; a 7 x 12
    window touches 2.6 tiles on average where a row-major level costs it 7..14 lines).  The algorithmic byte
    count (49 + 25 I bytes per patch and level) is below that floor because a line is fetched whole."""
    lay = W.store.layout
    cam = W.cam
    B = W.B
    idx = np.unique(np.linspace(1, B - 1, n_sample).astype(int))
    px_all = W.px_all.cpu().numpy()
    pos_all = W.pos_all.cpu().numpy()
    total = 0.0
    for b in idx:
        lines = set()
        T = W.T_gt[b]
        p = pos_all[b - 1] @ T[:9].reshape(3, 3).T + T[9:]
        px_cur = np.stack([cam.fx * p[:, 0] / p[:, 2] + cam.cx, cam.fy * p[:, 1] / p[:, 2] + cam.cy], -1)
        for role_px in (px_all[b], px_cur):
            for l in range(W.min_level, W.max_level + 1):
                u = np.floor(role_px[:, 0] / (1 << l)).astype(np.int64)
                v = np.floor(role_px[:, 1] / (1 << l)).astype(np.int64)
                ok = (u - 3 >= 0) & (v - 3 >= 0) & (u + 3 < lay.w[l]) & (v + 3 < lay.h[l])
                u, v = u[ok], v[ok]
                c0 = (u - 3) & ~3
                if lay.tile == capi.PYR_TILED:  # svo_pyr::run_start: the run stays inside a tile row when the 7 bytes do
                    c0 = np.where(((c0 & 15) == 8) & (((u - 3) & 15) + 7 <= 16), c0 - 4, c0)
                for r in range(-3, 4):
                    for cb in (c0, c0 + 4, c0 + 8):  # the three aligned dwords of a window row
                        lines.update((capi.pyr_px_offset(lay, l, cb, v + r) // 128).tolist())
        total += 128.0 * len(lines)
    return total / len(idx)


# Issue cost per wave-instruction and SIMD with 4 waves resident, measured by scripts/valu_ubench.hip on this part
# (profiles/r02_valu_ubench.json): f32 add/mul/fma 2.4, f64 3.5, conversions 3.0, f32 transcendentals 3.9,
# v_rcp_f64 & co 9.6, 32-bit integer/logic 2.5; whatever the class counters do not cover (moves, selects, DPP,
# lane reads) is priced at 3.0.
VALU_CLASS_COST = {"SQ_INSTS_VALU_ADD_F32": 2.4, "SQ_INSTS_VALU_MUL_F32": 2.4, "SQ_INSTS_VALU_FMA_F32": 2.4,
                   "SQ_INSTS_VALU_ADD_F64": 3.5, "SQ_INSTS_VALU_MUL_F64": 3.5, "SQ_INSTS_VALU_FMA_F64": 3.5,
                   "SQ_INSTS_VALU_CVT": 3.0, "SQ_INSTS_VALU_INT32": 2.5, "SQ_INSTS_VALU_INT64": 3.0,
                   "SQ_INSTS_VALU_TRANS_F32": 3.9, "SQ_INSTS_VALU_TRANS_F64": 9.6}
VALU_OTHER_COST = 3.0


def valu_roofline(raw: dict, kernel_ms: float) -> dict:
    """VALU issue utilisation of one K1 launch from the SQ counters (per-launch averages in `raw`): the instruction
    classes priced with their measured issue cost over the SIMD cycles of the launch.  The cycles come from
    SQ_BUSY_CU_CYCLES (summed over the CUs, i.e. this run's clock) when present, else from the kernel time at the
    nominal clock."""
    cu_cycles = raw["SQ_BUSY_CU_CYCLES"] / N_CU if raw.get("SQ_BUSY_CU_CYCLES") else kernel_ms * 1e-3 * CLOCK_GHZ * 1e9
    simd_cycles = N_SIMD * cu_cycles
    n_valu = raw["SQ_INSTS_VALU"]
    classified = sum(raw.get(k, 0.0) for k in VALU_CLASS_COST)
    busy = sum(raw.get(k, 0.0) * c for k, c in VALU_CLASS_COST.items()) + max(n_valu - classified, 0.0) * VALU_OTHER_COST
    wc = raw.get("SQ_WAVE_CYCLES")
    frac_of_wave = lambda k: (raw[k] / wc) if (wc and k in raw) else None
    return {"bound": "valu-issue", "kernel": "sia_kernel", "peak": 1.0, "achieved": busy / simd_cycles, "frac": busy / simd_cycles,
            "unit": "fraction of SIMD cycles the VALU is issuing: sum over instruction classes of (class count x measured issue cost) "
                    "/ (1024 SIMDs x SQ_BUSY_CU_CYCLES/256)",
            "lower_bound_2_cycles_per_instruction": n_valu * 2.0 / simd_cycles,
            "valu_instructions_per_launch": n_valu, "classified_by_counters_frac": classified / n_valu if n_valu else None,
            "simd_cycles_per_valu_instruction": simd_cycles / n_valu if n_valu else None,
            "all_instructions_per_launch": raw.get("SQ_INSTS"), "cu_cycles_per_launch": cu_cycles, "kernel_ms": kernel_ms,
            "wave_cycles_issuing_frac": frac_of_wave("SQ_ACTIVE_INST_ANY"), "wave_cycles_waiting_frac": frac_of_wave("SQ_WAIT_ANY"),
            "wave_cycles_at_waitcnt_frac": frac_of_wave("SQ_WAIT_INST_ANY")}


def lds_port_use(raw: dict) -> dict:
    """Use of the LDS port from per-launch counter averages: SQ_LDS_IDX_ACTIVE are the cycles the LDS arrays of all CUs
    work on indexed operations, SQ_LDS_BANK_CONFLICT the part of them lost to bank conflicts, SQ_BUSY_CU_CYCLES the busy
    cycles summed over the CUs (for K1 it equals 256 x kernel time x clock within 7 %).  rocprofiler-sdk's derived LDS
    utilisation divides by GRBM_GUI_ACTIVE x CU_NUM instead; on this box that counter comes out about eight times the
    kernel's cycles (one count per XCD), so that form is reported under its own name and not used.  Every ratio is None
    when its denominator was not collected."""
    gui = raw.get("GRBM_GUI_ACTIVE")
    idx, conf, n = raw.get("SQ_LDS_IDX_ACTIVE"), raw.get("SQ_LDS_BANK_CONFLICT"), raw.get("SQ_INSTS_LDS")
    wc = raw.get("SQ_WAVE_CYCLES")
    ratio = lambda a, b: (a / b) if (a is not None and b) else None
    return {"port_busy_frac": ratio(idx, raw.get("SQ_BUSY_CU_CYCLES")),
            "bank_conflict_frac_of_port_cycles": ratio(conf, idx),
            "port_cycles_per_lds_instruction": ratio(idx, n),
            "lds_instructions_per_launch": n,
            "wave_cycles_waiting_for_lds_issue_frac": ratio(4.0 * raw["SQ_WAIT_INST_LDS"] if "SQ_WAIT_INST_LDS" in raw else None, wc),
            "wave_cycles_in_lds_instructions_frac": ratio(raw.get("SQ_ACTIVE_INST_LDS"), wc),
            "sdk_formula_idx_active_over_gui_active_x_cus": ratio(idx, gui * N_CU if gui else None)}


def pmc_leg(args, kernel_ms: float, reserve_s: float = 0.0, only: tuple | None = None, lib_path: str | None = None,
            k1_kernel: str | None = None) -> dict:
    """HBM traffic and VALU issue of the headline kernel, measured on THIS box by re-running this
    command (headline leg only, 3 steps) under rocprofv3 --pmc, one counter group per pass as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one
    pass; no trace domain is combined with --pmc).  only: the passes to run (default all); lib_path: the library
    variant the children load (SVO_HIP_LIB; default: the one this process runs on)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return {"skipped": "rocprofv3 not on PATH"}
    base = [sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1", "--batch", str(args.batch),
            "--workload", args.workload, "--noise", str(args.noise), "--n-iter", str(args.n_iter), "--no-cpu-baseline",
            "--extras", "none", "--pmc-child", "1", "--k1-kernel", k1_kernel or args.k1_kernel, "--pipeline", "align",
            "--camera", getattr(args, "camera", "default")]
    passes = {"fetch": ["FETCH_SIZE"], "write": ["WRITE_SIZE"],
              "sq": ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                     "SQ_INSTS_VALU", "SQ_BUSY_CU_CYCLES"],
              # the VALU instruction mix, priced below with the per-class issue costs of scripts/valu_ubench.hip
              "mix": ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F64",
                      "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32"],
              "mix2": ["SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT64", "SQ_INSTS", "SQ_INSTS_SALU",
                       "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_BRANCH"],
              # the LDS port of the CU (lds_port_use); last: a failure here costs nothing that came before
              "lds": ["SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS"]}
    if only is not None:
        passes = {k: v for k, v in passes.items() if k in only}
    status, raw = {}, {}
    env = dict(os.environ, TMPDIR="/tmp")
    if lib_path:
        env["SVO_HIP_LIB"] = lib_path
    env.pop("SVO_BENCH_FORCE_DIST", None)
    # a pass takes ~16 s; a profiler that hangs or fails must not hold the headline line back: one strike and the
    # remaining passes are skipped (worst case PMC_PASS_TIMEOUT_S on top of the run)
    broken = False
    for name, ctrs in passes.items():
        if broken:
            status[name] = "skipped (an earlier pass failed)"
            continue
        # (the instruction-mix and LDS passes give way to what is still to come: --time-budget)
        if name not in ("fetch", "write", "sq") and time_left(getattr(args, "time_budget", 0.0)) < PMC_PASS_RESERVE_S + reserve_s:
            status[name] = "skipped (time budget)"
            continue
        d = tempfile.mkdtemp(prefix=f"svo_pmc_{name}_", dir="/tmp")
        cmd = [exe, "--pmc", *ctrs, "--kernel-include-regex", "sia_(wave_)?kernel", "--output-format", "csv", "-d", d, "-o", name, "--", *base]
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=PMC_PASS_TIMEOUT_S)
        except subprocess.TimeoutExpired:
            status[name] = "timeout"
            broken = True
            continue
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            status[name] = f"rc={p.returncode}: {p.stderr[-200:]}"
            broken = True
            continue
        acc: dict[str, list[float]] = {}
        with open(files[0]) as fh:
            for row in csv.DictReader(fh):
                if "sia_kernel" not in row.get("Kernel_Name", "") and "sia_wave_kernel" not in row.get("Kernel_Name", ""):
                    continue
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in acc.items():
            raw[k] = float(np.mean(v))  # per launch
        status[name] = "ok"
        shutil.rmtree(d, ignore_errors=True)
    out: dict = {"passes": status, "counters_per_launch": raw,
                 "how": "child runs of this command (3 steps) under rocprofv3 --pmc, one counter group per pass (FETCH_SIZE and "
                        "WRITE_SIZE cannot share one), averaged per sia_kernel launch; KiB units; FETCH_SIZE doubled (gfx950 "
                        "correction of the guide)"}
    if "FETCH_SIZE" in raw and "WRITE_SIZE" in raw:
        # MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies a
        # 128-byte fabric request as 64 bytes -> doubled (the guide's correction for reads)
        out["traffic_bytes_per_launch"] = (2.0 * raw["FETCH_SIZE"] + raw["WRITE_SIZE"]) * 1024.0
        out["fetch_bytes_per_launch_corrected_x2"] = 2.0 * raw["FETCH_SIZE"] * 1024.0
        out["fetch_bytes_per_launch_raw_counter"] = raw["FETCH_SIZE"] * 1024.0
        out["write_bytes_per_launch"] = raw["WRITE_SIZE"] * 1024.0
    if "SQ_INSTS_VALU" in raw:
        out["roofline_valu"] = valu_roofline(raw, kernel_ms)
        if "SQ_LDS_IDX_ACTIVE" in raw:
            try:
                out["roofline_valu"]["lds"] = lds_port_use(raw)
            except Exception as e:  # (never in the way of the line)
                out["roofline_valu"]["lds"] = {"skipped": repr(e)}
    return out


def full_track_kernel_rooflines(ft: dict, pf: dict, frames_per_step: int, patches: int) -> dict:
    """Per kernel (and use) of the full-track step: rocprofv3's duration (kernel-trace pass of pmc_full_track_leg), counter
    traffic, and the bytes the kernel has to move AS IT IS CUT -- its inputs once, its outputs once, windows by their
    footprint -- with the fraction of the HBM roofline those bytes reach.  Formulas (bytes per unit):
      match_prepare   candidate: 52 in (frame, position, observation range, projection); observation record: 52;
                      tried trial: 62 of warp / alignment parameters + 8 of results
      warp_kernel     (the matcher's) trial: 37 parameters + 121 footprint of the 10 x 10 template in the reference level + 100 written
      align_kernel    trial: 100 template + 50 parameters / results; evaluation: 81 (the 9 x 9 window); as the depth filter's
                      last kernel (round 6: seed_finish is its epilogue) + 96 per seed (state in / out, bearing, flags, status)
      pose_opt        frame: 52 per observation + 416
      seed_prepare    seed: 76 in (seed state, its feature, the frame indices) + 64 of warp / scan parameters
      epi_scan        (round 6: with the affine warp) warped seed: 37 parameters + 121 footprint of the template; scanning seed:
                      32 (segment) + 64 + 7.13 per scanned position: the UNION of the 8 x 8 windows along the segment (0.7 px apart;
                      8 * 0.7 * (|cos| + |sin|), 4 / pi on average) -- the 64 bytes per position of SURVEY 8(d) count every window
                      in full although neighbours overlap by 7/8; a seed that goes on to the alignment: 100 written + 36 of results
      seed_finish     (batches up to 8192 seeds only: a launch of its own) seed: 96"""
    rl = ft.get("rooflines", {})
    fm, us = rl.get("find_match_direct", {}), rl.get("update_seeds", {})
    n_tried, M, n_obs = fm.get("trials", 0.0), fm.get("candidates", 0.0), fm.get("observations", 0.0)
    n_eval = fm.get("alignment_evaluations_per_trial", 0.0) * n_tried
    S, n_pos, S_scan = us.get("seeds", 0.0), us.get("scanned_positions_per_seed", 0.0) * us.get("seeds", 0.0), us.get("seeds_scanning", 0.0)
    al = us.get("alignment") if isinstance(us.get("alignment"), dict) else {}
    S_al, us_eval = al.get("aligned_seeds"), al.get("evaluations")
    st = ft.get("seed_status_per_frame", {})
    S_warp = frames_per_step * sum(v for k, v in st.items() if k in ("updated", "converged", "no_match", "nan")) if st else S
    fused_finish = "update_seeds/seed_finish_kernel" not in pf.get("kernels", {})
    need = {"find_match_direct/match_prepare_kernel": M * 52.0 + n_obs * 52.0 + n_tried * 70.0,
            "find_match_direct/warp_kernel": n_tried * 258.0,
            "find_match_direct/align_kernel": n_tried * 150.0 + n_eval * 81.0,
            "pose_optimize/pose_opt_wave_kernel": frames_per_step * (patches * 52.0 + 416.0),
            "update_seeds/seed_prepare_kernel": S * 140.0,
            "update_seeds/epi_scan_kernel": S_warp * 158.0 + S_scan * 96.0 + n_pos * 7.13 + (S_al or 0.0) * 136.0,
            "update_seeds/align_kernel": (S_al * 150.0 + us_eval * 81.0 + (S * 96.0 if fused_finish else 0.0)) if S_al is not None and us_eval is not None else None,
            "update_seeds/seed_finish_kernel": S * 96.0}
    out = {}
    for kn, kd in pf.get("kernels", {}).items():
        e = dict(kd)
        b = need.get(kn)
        e["compulsory_bytes"] = b
        if b and kd.get("ms", 0) > 0:
            e["achieved_GBs"] = b / (kd["ms"] * 1e-3) / 1e9
            e["frac"] = e["achieved_GBs"] / HBM_PEAK_GBS
            if kd.get("traffic_bytes"):
                e["traffic_over_compulsory"] = kd["traffic_bytes"] / b
        out[kn] = e
    return out


FULL_TRACK_KERNELS = {"find_match_direct": ("match_prepare_kernel", "warp_kernel", "align_kernel"),
                      "update_seeds": ("seed_prepare_kernel", "epi_scan_kernel", "align_kernel", "seed_finish_kernel"),
                      "pose_optimize": ("pose_opt_wave_kernel", "pose_opt_kernel")}
# the kernel that OPENS a stage of the step (FullTrack.step): whatever is dispatched after it belongs to that stage until the
# next opener.  align_kernel (and, in round 5, warp_kernel) serves two stages: its launches are told apart by where they
# stand in the dispatch order, per launch, not by name -- also for the counter passes
STAGE_OPENERS = {"match_prepare_kernel": "find_match_direct", "pose_opt_wave_kernel": "pose_optimize", "seed_prepare_kernel": "update_seeds"}


def attribute_dispatches(rows: list, n_steps: int) -> dict:
    """rows: (order key, kernel short name, {quantity: value}) of ONE child run of `bench.py --pipeline full`, any order.
    Returns {"stage/kernel": {quantity: mean over the last n_steps steps of the per-step SUM, "launches_per_step": ...}}.
    A step starts at its match_prepare_kernel; what precedes the first of the last n_steps steps (the workload's set-up
    launches the same kernels) is dropped."""
    rows = sorted(rows, key=lambda r: r[0])
    starts = [i for i, r in enumerate(rows) if r[1] == "match_prepare_kernel"]
    if len(starts) < n_steps:
        return {}
    first = starts[-n_steps]
    acc, stage, step = {}, None, -1
    for _, short, q in rows[first:]:
        if short == "match_prepare_kernel":
            step += 1
        stage = STAGE_OPENERS.get(short, stage)
        if stage is None or short not in FULL_TRACK_KERNELS[stage]:
            continue
        e = acc.setdefault(f"{stage}/{short}", {})
        for k, v in q.items():
            e[k] = e.get(k, 0.0) + v
        e["_launches"] = e.get("_launches", 0) + 1
    out = {}
    for kn, e in acc.items():
        out[kn] = {k: v / n_steps for k, v in e.items() if k != "_launches"}
        out[kn]["launches_per_step"] = e["_launches"] / n_steps
    return out


def pmc_full_track_leg(args, n_steps: int = 2) -> dict:
    """The kernels of the full-track step (configs[2], representative workload), per STAGE and KERNEL: child runs of
    `bench.py --pipeline full` under rocprofv3 -- a kernel trace (durations), FETCH_SIZE and WRITE_SIZE (one counter per
    pass, corrected as for K1), an SQ pass (VALU issue, waiting) and an LDS pass -- each attributed per dispatch by its
    place in the step (attribute_dispatches), so that the two uses of align_kernel have their own rows in every pass."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return {"skipped": "rocprofv3 not on PATH"}
    base = [sys.executable, os.path.abspath(__file__), "--steps", str(n_steps), "--warmup", "0", "--batch", str(args.batch),
            "--workload", args.workload, "--n-iter", str(args.n_iter), "--no-cpu-baseline", "--extras", "none",
            "--pmc-child", "1", "--pipeline", "full"]
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("SVO_BENCH_FORCE_DIST", None)
    names = sorted({k for ks in FULL_TRACK_KERNELS.values() for k in ks}, key=len, reverse=True)  # (longest first: pose_opt_wave_kernel / pose_opt_kernel)
    regex = "|".join(names)
    short_of = lambda kn: next((k for k in names if k in kn), None)
    status = {}

    def counter_pass(tag, ctrs, timeout_s=2 * PMC_PASS_TIMEOUT_S):
        """one child under --pmc: {stage/kernel: {counter: per-step sum}}"""
        d = tempfile.mkdtemp(prefix=f"svo_pmc_full_{tag}_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", *ctrs, "--kernel-include-regex", regex, "--output-format", "csv", "-d", d, "-o", tag, "--", *base]
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                status[tag] = f"rc={p.returncode}: {p.stderr[-200:]}"
                return None
            by_dispatch = {}
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    short = short_of(row["Kernel_Name"])
                    if short and row.get("Counter_Name") in ctrs:
                        e = by_dispatch.setdefault(int(row["Dispatch_Id"]), (short, {}))
                        e[1][row["Counter_Name"]] = e[1].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            status[tag] = "ok"
            return attribute_dispatches([(did, sh, q) for did, (sh, q) in by_dispatch.items()], n_steps)
        except subprocess.TimeoutExpired:
            status[tag] = "timeout"
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)

    fetch = counter_pass("fetch", ["FETCH_SIZE"])
    write = counter_pass("write", ["WRITE_SIZE"]) if fetch is not None else None
    if fetch is None or write is None:
        return {"passes": status}
    stages = {}
    traffic = {}
    for kn in sorted(set(fetch) | set(write)):
        t = (2.0 * fetch.get(kn, {}).get("FETCH_SIZE", 0.0) + write.get(kn, {}).get("WRITE_SIZE", 0.0)) * 1024.0  # KiB; FETCH doubled (gfx950)
        traffic[kn] = t
        stage, short = kn.split("/")
        st = stages.setdefault(stage, {"traffic_bytes_per_step": 0.0, "by_kernel": {}})
        st["by_kernel"][short] = t
        st["traffic_bytes_per_step"] += t

    # ---- per kernel and use: duration (a kernel-trace pass of the same child) ------------------------------------------
    kernels = {}
    try:
        d = tempfile.mkdtemp(prefix="svo_trace_full_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "trace", "--", *base]
        p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=2 * PMC_PASS_TIMEOUT_S)
        files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if p.returncode != 0 or not files:
            status["trace"] = f"rc={p.returncode}: {p.stderr[-200:]}"
        else:
            rows = []
            with open(files[0]) as fh:
                for row in csv.DictReader(fh):
                    short = short_of(row["Kernel_Name"])
                    if short:
                        rows.append((int(row["Start_Timestamp"]), short, {"ms": (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6}))
            kernels = attribute_dispatches(rows, n_steps)
            status["trace"] = "ok"
        shutil.rmtree(d, ignore_errors=True)
    except subprocess.TimeoutExpired:
        status["trace"] = "timeout"
    # ---- VALU issue and waiting; the LDS port (the four SIMDs of a CU share it) -----------------------------------------
    sq = counter_pass("sq", ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                             "SQ_INSTS_VALU", "SQ_BUSY_CU_CYCLES"]) if kernels else None
    for kn, raw in (sq or {}).items():
        if kn in kernels and raw.get("SQ_BUSY_CU_CYCLES", 0) > 0 and raw.get("SQ_INSTS_VALU", 0) > 0:
            vr = valu_roofline(raw, 1.0)
            kernels[kn]["valu"] = {"busy_frac_at_3_cycles_per_instruction": vr["frac"],
                                   "busy_frac_lower_bound_2_cycles": vr["lower_bound_2_cycles_per_instruction"],
                                   "wave_cycles_waiting_frac": vr["wave_cycles_waiting_frac"],
                                   "wave_cycles_at_waitcnt_frac": vr["wave_cycles_at_waitcnt_frac"],
                                   "valu_instructions_per_step": raw["SQ_INSTS_VALU"]}
    lds = counter_pass("lds", ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_LDS",
                               "SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"]) if kernels else None
    for kn, raw in (lds or {}).items():
        if kn in kernels:
            try:
                kernels[kn]["lds"] = lds_port_use(raw)
            except Exception as e:
                kernels[kn]["lds"] = {"skipped": repr(e)}
    for kn, kd in kernels.items():
        t = traffic.get(kn)
        if t is not None and t == t:
            kd["traffic_bytes"] = t
            kd["traffic_GBs"] = t / (kd["ms"] * 1e-3) / 1e9 if kd.get("ms", 0) > 0 else None
    return {"passes": status, "stages": stages, "kernels": kernels,
            "how": f"child runs of `bench.py --pipeline full` ({n_steps} steps) under rocprofv3: kernel trace, then --pmc passes "
                   "(FETCH_SIZE and WRITE_SIZE separately, KiB, FETCH_SIZE doubled on gfx950; SQ; LDS), every dispatch attributed to "
                   "its stage by its place in the step, counters summed per step and averaged over the steps"}


def pyramid_roofline(ev: Events, store, images, reps: int = 5) -> dict:
    """K0, the one HBM-streaming kernel of the path: image pyramids of the whole
    replay batch rebuilt from the packed images in a single fused pass.  Algorithmic bytes per
    frame = w*h read + every level written once."""
    n = images.shape[0]
    per_frame = images.shape[1] * images.shape[2] + store.bytes_per_pyramid()
    tiles = {str(tw): ev.time(lambda: store.load_images(images, 0, tile=tw), reps) for tw in (128, 256, 257, 512)}
    fused = ev.time(lambda: store.load_images(images, 0), reps)

    def per_level():
        store.load_images(images, 0, build=False)
        store.build_per_level(0, n)
    unfused = ev.time(per_level, reps)
    gbs = n * per_frame / (fused * 1e-3) / 1e9
    return {"kernel": "pyramid_fused_kernel (svo_hip_pyramid_build_from_images)", "frames": int(n),
            "ms": fused, "frames_per_s": n / (fused * 1e-3), "algorithmic_bytes_per_frame": int(per_frame),
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": None, "algorithmic_bytes_per_launch": int(n * per_frame),
            "ms_level0_copy_plus_one_launch_per_level": unfused, "ms_by_tile_width": tiles}


def dropin_sequence(n_frames: int = 600) -> dict:
    """Single-stream, image-in -> pose-out: the reference's own svo::FrameHandlerMono on a
    752x480 synthetic sequence, once with all-reference translation units on the host CPU and
    once with the drop-in HIP bodies (tests/dropin, rpg_svo_amd/host/dropin).  Reports the
    trajectory agreement (the metric's "ATE vs CPU ref") and the per-frame latency of both.

    600 frames: the map then holds what the reference's own trace shows (svo/test/benchmark.csv:
    max_n_kfs = 10 overlapping keyframes, ~950 candidate points) -- 10 keyframes from frame ~150 on and 1400-1900
    candidates; `map_size` reports the medians over the frames the latencies are taken from."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
    import pypipeline as pp
    if not (pp.available("ref") and pp.available("hip")):
        raise RuntimeError("tests/dropin/_build/*.so not built")
    cam = synth.Camera(752, 480, 315.5, 315.5, 376.0, 240.0)
    T = synth.make_trajectory(n_frames, seed=5, max_step=0.02, max_rot_deg=0.3)
    imgs = synth.render(synth.make_texture(seed=12345), T, cam, device="cuda" if torch.cuda.is_available() else "cpu").cpu().numpy()
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)  # the reference logs every frame to stderr
    try:
        ref = pp.run_sequence("ref", cam, imgs, T)
        pp.run_sequence("hip", cam, imgs[:10], T[:10])  # warm-up: context creation, first launches
        host, host_ref, host_def = {}, {}, {}
        hip = pp.run_sequence("hip", cam, imgs, T, stats_out=host)
        # the mapper off the frame's critical path, both ways: the reference's mapping thread (its default; timing
        # dependent, so no frame-by-frame parity) and the drop-in's deferred mapping (opt-in, deterministic)
        hip_def = pp.run_sequence("hip", cam, imgs, T, stats_out=host_def, defer_mapper=1)
        ref_thr = pp.run_sequence("ref", cam, imgs, T, stats_out=host_ref, mapper_thread=1)
        hip_thr = pp.run_sequence("hip", cam, imgs, T, mapper_thread=1)
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
    Tr = np.stack([r["T_f_w"] for r in ref])
    Th = np.stack([r["T_f_w"] for r in hip])
    d = se3.log_norm(Th, Tr)
    # A pipeline of thresholds is chaotic over hundreds of frames: the first discrete decision that falls the other way
    # (a seed whose variance sits on the convergence threshold converges one update later; the mock device, which runs
    # the oracle's arithmetic, does so at frame 137 of this sequence) changes the map, and the two runs are different
    # -- equally good -- SLAM runs from there on.  So: frame-by-frame agreement up to that frame, and the accuracy of
    # BOTH runs against the ground-truth trajectory over the whole sequence.
    dec_keys = ("is_keyframe", "n_obs", "repr_n_mps", "repr_n_new_references", "n_kfs", "stage", "img_align_n_tracked",
                "n_candidates", "n_seeds")
    first_diff = next((i for i, (a, b) in enumerate(zip(ref, hip)) if any(a[k] != b[k] for k in dec_keys)), None)
    # ... and of the TRACKER's decisions alone (keyframes, matches, trials, tracked patches: the keys
    # tests/test_dropin_pipeline.py compares over 120 frames), which the mapper's one-update shifts reach only later
    trk_keys = ("is_keyframe", "n_obs", "repr_n_mps", "repr_n_new_references", "n_kfs", "stage", "img_align_n_tracked")
    first_trk = next((i for i, (a, b) in enumerate(zip(ref, hip)) if any(a[k] != b[k] for k in trk_keys)), None)
    pre = slice(0, first_diff if first_diff is not None else n_frames)
    pos = lambda TT: se3.inv(TT)[:, 9:]
    med = lambda rs, k: float(np.median([r[k] for r in rs[1:]]) * 1e3)
    stages = ("tot_time", "sparse_img_align", "reproject", "pose_optimizer")
    medi = lambda rs, k: float(np.median([r[k] for r in rs[1:]]))
    map_size = {"frames_with_full_keyframe_set": int(sum(r["n_kfs"] >= 10 for r in hip)),
                "n_kfs": medi(hip, "n_kfs"), "overlap_kfs": medi(hip, "n_overlap_kfs"),
                "n_candidates": medi(hip, "n_candidates"), "n_candidates_max": int(max(r["n_candidates"] for r in hip)),
                "kf_points_in_frame": medi(hip, "n_kf_points_in_frame"), "n_seeds": medi(hip, "n_seeds"),
                "trials": medi(hip, "repr_n_mps"), "matches": medi(hip, "repr_n_new_references"),
                "same_as_cpu_reference": all(all(a[k] == b[k] for k in ("n_kfs", "n_overlap_kfs", "n_kf_points_in_frame", "repr_n_mps", "repr_n_new_references"))
                                             for a, b in zip(ref, hip))}
    # Reprojector::reprojectMap on the list-walking path (SVO_HIP_MAP_MIRROR=off) in a child process: what the
    # device-resident map mirror (row N2, the default) buys, and that it changes nothing
    list_walk = None
    try:
        import tempfile
        dump = tempfile.mktemp(suffix=".npy", dir="/tmp")
        code = f"import sys, json; sys.path.insert(0, {ROOT!r}); import bench; print(json.dumps(bench.dropin_hip_only({n_frames}, {dump!r})))"
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SVO_HIP_MAP_MIRROR="off"), capture_output=True, text=True,
                           timeout=300)
        if p.returncode == 0:
            list_walk = json.loads(p.stdout.strip().splitlines()[-1])
            list_walk["trajectory_identical_to_the_mirror_path"] = bool(np.array_equal(np.load(dump), Th))
            list_walk.pop("map_mirror", None)
            os.unlink(dump)
        else:
            list_walk = {"skipped": p.stderr[-300:]}
    except Exception as e:
        list_walk = {"skipped": repr(e)}
    # the same sequence with the host pyramid built as the reference builds it (SVO_HIP_HOST_PYRAMID=1, read once per process: a
    # child): what rpg_svo_amd/host/dropin/frame.cpp saves by keeping level 0 only
    try:
        dump = tempfile.mktemp(suffix=".npy", dir="/tmp")
        code = f"import sys, json; sys.path.insert(0, {ROOT!r}); import bench; print(json.dumps(bench.dropin_hip_only({n_frames}, {dump!r})))"
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SVO_HIP_HOST_PYRAMID="1"), capture_output=True, text=True,
                           timeout=300)
        if p.returncode == 0:
            with_pyr = json.loads(p.stdout.strip().splitlines()[-1])
            with_pyr["trajectory_identical"] = bool(np.array_equal(np.load(dump), Th))
            with_pyr.pop("map_mirror", None)
            os.unlink(dump)
        else:
            with_pyr = {"skipped": p.stderr[-300:]}
    except Exception as e:
        with_pyr = {"skipped": repr(e)}
    # every step a call of its own (SVO_HIP_CHAIN=0: the round-5 path): what the chain behind the sparse alignment buys
    try:
        dump = tempfile.mktemp(suffix=".npy", dir="/tmp")
        code = f"import sys, json; sys.path.insert(0, {ROOT!r}); import bench; print(json.dumps(bench.dropin_hip_only({n_frames}, {dump!r})))"
        p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SVO_HIP_CHAIN="0"), capture_output=True, text=True,
                           timeout=300)
        if p.returncode == 0:
            no_chain = json.loads(p.stdout.strip().splitlines()[-1])
            no_chain["trajectory_identical_to_the_chained_path"] = bool(np.array_equal(np.load(dump), Th))
            no_chain.pop("map_mirror", None)
            os.unlink(dump)
        else:
            no_chain = {"skipped": p.stderr[-300:]}
    except Exception as e:
        no_chain = {"skipped": repr(e)}
    # The drop-in in a process of its own, as a host runs it (this process has been through every other leg by now -- a
    # dozen streams, the reference's own run, counters -- and measures the same frames 5-12 % slower): synchronous and deferred
    # mapper alternating, two processes each (medians reported); same frames, same trajectory.
    own = {"hip_dropin": [], "hip_dropin_deferred_mapper": [], "hip_dropin_bound_to_the_gpus_numa_node": [],
           "hip_dropin_deferred_mapper_bound_to_the_gpus_numa_node": []}
    own_same = True
    # ... and bound (taskset) to the CPUs next to the GPU, as INTEGRATION.md recommends for a single camera: a frame is a dozen
    # host <-> device hand-overs through pinned memory (scripts/numa_placement.py, profiles/r06aa_*)
    local_cpus = None
    try:
        pr = torch.cuda.get_device_properties(0)
        f = "/sys/bus/pci/devices/%04x:%02x:%02x.0/local_cpulist" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        local_cpus = open(f).read().strip() or None
        if local_cpus is not None and not shutil.which("taskset"):
            local_cpus = None
    except Exception:
        local_cpus = None
    try:
        for rep in range(2):
            for key, dm, bound in (("hip_dropin", 0, False), ("hip_dropin_deferred_mapper", 1, False),
                                   ("hip_dropin_bound_to_the_gpus_numa_node", 0, True),
                                   ("hip_dropin_deferred_mapper_bound_to_the_gpus_numa_node", 1, True)):
                if bound and local_cpus is None:
                    continue
                dump = tempfile.mktemp(suffix=".npy", dir="/tmp")
                code = (f"import sys, json; sys.path.insert(0, {ROOT!r}); import bench; "
                        f"print(json.dumps(bench.dropin_hip_only({n_frames}, {dump!r}, defer_mapper={dm})))")
                cmd = (["taskset", "-c", local_cpus] if bound else []) + [sys.executable, "-c", code]
                p = subprocess.run(cmd, env=dict(os.environ), capture_output=True, text=True, timeout=300)
                if p.returncode != 0:
                    raise RuntimeError(p.stderr[-300:])
                r = json.loads(p.stdout.strip().splitlines()[-1])
                own[key].append(r["tot_time"])
                own_same = own_same and bool(np.array_equal(np.load(dump), Th))
                os.unlink(dump)
        own_process = {k: float(np.median(v)) for k, v in own.items() if v}
        own_process.update(runs=own, trajectory_identical_to_this_process=own_same, gpus_local_cpulist=local_cpus)
    except Exception as e:
        own_process = {"skipped": repr(e)}
    return {"frames": n_frames, "map_size": map_size, "host_pyramid_levels_built_of": host.get("host_pyramid"),
            "median_ms_per_frame_in_a_process_of_its_own": own_process,
            "median_ms_per_frame_hip_dropin_with_the_host_pyramid_built": with_pyr,
            "median_ms_per_frame_hip_dropin_without_the_frame_chain": no_chain,
            "map_mirror": host.get("map_mirror"), "seed_store": host.get("seed_store"),
            "median_ms_per_frame_hip_dropin_list_walking_reprojector": list_walk,
            "first_frame_with_a_different_decision": first_diff,
            "first_frame_with_a_different_tracking_decision": first_trk,
            "se3_lognorm_max_before_the_first_tracking_difference": float(d[:first_trk if first_trk is not None else n_frames].max()),
            "se3_lognorm_max_before_it": float(d[pre].max()), "se3_lognorm_median_before_it": float(np.median(d[pre])),
            "ate_rmse_vs_ground_truth_m": {"cpu_reference": horn_ate(pos(Tr), pos(T)), "hip_dropin": horn_ate(pos(Th), pos(T))}, "image": "752x480", "se3_lognorm_max": float(d.max()), "se3_lognorm_median": float(np.median(d)),
            "ate_rmse_vs_cpu_m": horn_ate(se3.inv(Th)[:, 9:], se3.inv(Tr)[:, 9:]),
            "keyframes": int(sum(r["is_keyframe"] for r in hip)),
            "same_keyframe_frames": [r["is_keyframe"] for r in ref] == [r["is_keyframe"] for r in hip],
            "median_ms_per_frame_cpu_reference": {k: med(ref, "t_" + k) for k in stages},
            "median_ms_per_frame_hip_dropin": {k: med(hip, "t_" + k) for k in stages},
            # opt-in (svo_hip::Device::setDeferredMapping): updateSeeds returns with its kernels running, results
            # reach the seed list / the map at the next reprojectMap -- same trajectory, bit for bit
            "median_ms_per_frame_hip_dropin_deferred_mapper": {k: med(hip_def, "t_" + k) for k in stages},
            "deferred_mapper_trajectory_identical": bool(np.array_equal(np.stack([r["T_f_w"] for r in hip_def]), Th)),
            "frame_period_ms_back_to_back": {"hip_dropin": host.get("wall_ms_per_frame"),
                                             "hip_dropin_deferred_mapper": host_def.get("wall_ms_per_frame")},
            # DepthFilter's own thread running (the reference's default mode): tot_time then excludes the mapper
            "median_ms_per_frame_mapper_thread": {"cpu_reference": med(ref_thr, "t_tot_time"), "hip_dropin": med(hip_thr, "t_tot_time")},
            "predicted_pose_refinements": {"taken": host.get("predicted_pose_hits"), "not_taken": host.get("predicted_pose_misses")},
            # round 6: reprojection + matching + selection + pose refinement enqueued behind the sparse alignment
            # (rpg_svo_amd/host/dropin/frame_chain.h), verified and taken by reprojectMap
            "frame_chain": {"taken": host.get("frame_chain_hits"), "not_taken": host.get("frame_chain_misses")},
            # ... and the depth filter's update enqueued by the pose optimizer's drop-in, before the host's bookkeeping of the
            # frame (dropin/depth_filter.cpp, EarlyUpdate): taken by the reference's updateSeeds call / dropped (keyframes)
            "early_mapper": {"taken": host.get("early_mapper_taken"), "dropped": host.get("early_mapper_dropped"),
                             "launched_in_two_phases": host.get("early_mapper_two_phase")},
            # N2 evidence: per drop-in call, the host walking the reference's pointer graph into the pinned
            # arena and back (marshal/unmarshal) against the device round trip (H2D + kernels + D2H + sync)
            "host_vs_device_us_per_call": {k: {q: round(v, 2) if isinstance(v, float) else v for q, v in st.items()}
                                           for k, st in host.get("stages", {}).items()},
            "pyramid_uploads": host.get("uploads"), "pyramid_upload_us_per_frame": host.get("pyramid_upload_us_total", 0.0) / max(1, n_frames - 1)}


def reference_cameras_leg(args, ev: Events, dev, rank: int, lib, B: int = 4096, dropin_frames: int = 200) -> dict:
    """the camera models the reference actually ships (svo_ros/param/camera_atan.yaml, camera_pinhole.yaml:
    both distorted) next to the undistorted pinhole every other number is measured on.  Per camera, on 752x480 images rendered
    through the model: K1 at configs[1]'s shape (4 levels 3 -> 0, 200 patches; B frames), the representative full-track step
    (configs[2]) and the single-stream drop-in (hip flavour, `dropin_frames` frames, a child process); K1's counter traffic is
    added by reference_cameras_traffic() after the headline's own counter passes."""
    out = {"frames_per_step": B, "image": "752x480", "workload": "ref752_4_n200_sparse_align", "cameras": {}}
    cams = reference_cameras()
    for name, cam in cams.items():
        if time_left(args.time_budget) < 25.0:
            out["cameras"][name] = {"skipped": "time budget"}
            continue
        r = {}
        try:
            W = Workload("ref752_4_n200_sparse_align", B, dev, rank, cam=cam)
            sia = SparseImgAlign(W.max_level, W.min_level, args.n_iter)
            sia.kernel = "workgroup"
            o = sia.alloc_result(B, dev)
            ms = ev.time(lambda: W.run_align(sia, out=o), 10, warmup=3)
            torch.cuda.synchronize()
            st = W.align_stats(o)
            r["sparse_align"] = {"frames_per_s": B / ms * 1e3, "ms_per_step": ms,
                                 "mean_gn_iterations_per_frame": float(st["iters"].sum(1).mean()),
                                 "mean_tracked_patches": float(st["n_tracked"].mean()),
                                 "median_pose_error_vs_gt": float(np.median(st["gt_err"])),
                                 "roofline": _pick(roofline("sia_kernel", st["alg_bytes"], ms), ("achieved", "frac", "ms", "algorithmic_bytes_per_launch"))}
            if not args.no_cpu_baseline:
                # the reference's own SparseImgAlign through the same camera model on a sample of the same frame pairs: the pose,
                # and that the iteration counts -- higher on the distorted models -- are the reference's own
                try:
                    from oracle import pyoracle
                    k = 64
                    which = "ref" if pyoracle.ref_available() else "orc"
                    pyrs = [pyoracle.create_img_pyramid(im, W.n_levels, pyoracle.HALFSAMPLE_AUTO) for im in W.images[:k + 1].cpu().numpy()]
                    rs = np.arange(k, dtype=np.int32)
                    T_cpu, res = pyoracle.sparse_img_align_batch(pyrs, rs, rs + 1, cam, W.T_ref_w[:k], W.T_prior_w[:k], np.full(k, W.n_patches, np.int32),
                                                                 W.px_all[:k].cpu().numpy(), W.f_all[:k].cpu().numpy(), np.ones((k, W.n_patches), np.uint8),
                                                                 W.pos_all[:k].cpu().numpy(), W.max_level, W.min_level, args.n_iter,
                                                                 n_threads=min(16, os.cpu_count() or 1), which=which)
                    d = se3.log_norm(st["T_est_w"][:k], T_cpu)
                    it_g = o.iters.cpu().numpy()[:k]
                    r["sparse_align"]["parity"] = {"frames_compared": k, "against": "reference" if which == "ref" else "port",
                                                   "se3_lognorm_max": float(d.max()), "se3_lognorm_median": float(np.median(d)),
                                                   "same_iteration_counts_frac": float(np.mean([np.array_equal(a["iters"], b) for a, b in zip(res, it_g)])),
                                                   "mean_gn_iterations_per_frame_reference": float(np.mean([np.sum(a["iters"]) for a in res]))}
                except Exception as e:
                    r["sparse_align"]["parity"] = {"skipped": repr(e)}
            full = FullTrack(W, dev, rank)
            marks = []
            for i in range(4):
                e0 = ev.mark()
                W.run_align(sia, out=o)
                full.step(o.T_cur_from_ref, ev if i > 0 else None)
                marks.append((e0, ev.mark()))
            torch.cuda.synchronize()
            stages = full.stage_ms(ev)
            d = full.describe()
            T_ref = d.pop("_T_refined")
            r["full_track"] = {"ms_per_step": float(np.mean([ev.ms(a, b) for a, b in marks[1:]])), "stages_ms": stages,
                               "matches_per_frame": d["matches_per_frame"], "seeds_per_frame": d["seeds_per_frame"],
                               "seed_status_per_frame": d["seed_status_per_frame"],
                               "median_pose_error_vs_gt_after_refine": float(np.median(se3.log_norm(T_ref, W.T_gt[1:B + 1])))}
            r["full_track"]["frames_per_s"] = B / r["full_track"]["ms_per_step"] * 1e3
            del full, W, o
            torch.cuda.empty_cache()
        except Exception as e:
            r["skipped"] = repr(e)
        out["cameras"][name] = r
    # the single-stream drop-in per camera (children: the device context is per process and camera)
    for name in cams:
        if time_left(args.time_budget) < 20.0 or "skipped" in out["cameras"].get(name, {}):
            continue
        try:
            code = (f"import sys, json; sys.path.insert(0, {ROOT!r}); import bench; "
                    f"print(json.dumps(bench.dropin_hip_only({dropin_frames}, '', camera={name!r})))")
            p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
            out["cameras"][name]["dropin"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else {"skipped": p.stderr[-300:]}
        except Exception as e:
            out["cameras"][name]["dropin"] = {"skipped": repr(e)}
    # The distorted models converge in MORE Gauss-Newton iterations on the same frames -- the reference's Jacobian is the
    # undistorted pinhole's whatever the camera (sparse_img_align.cpp:104-105: frame jacobian_xyz2uv * focal_length), so
    # its steps are less exact where the distortion is strong: the kernel is compared per ITERATION, the whole run beside it
    pin = out["cameras"].get("pinhole_undistorted", {}).get("sparse_align", {})
    for name, r in out["cameras"].items():
        sa = r.get("sparse_align")
        if isinstance(sa, dict):
            sa["us_per_gn_iteration_of_the_batch"] = sa["ms_per_step"] * 1e3 / sa["mean_gn_iterations_per_frame"]
    if pin:
        for name, r in out["cameras"].items():
            sa = r.get("sparse_align")
            if isinstance(sa, dict):
                sa["ms_over_undistorted_pinhole"] = sa["ms_per_step"] / pin["ms_per_step"]
                sa["per_iteration_over_undistorted_pinhole"] = sa["us_per_gn_iteration_of_the_batch"] / pin["us_per_gn_iteration_of_the_batch"]
    return out


def reference_cameras_traffic(args, out: dict, B: int = 4096) -> None:
    """K1's counter traffic on the two distorted cameras (rocprofv3 --pmc children: FETCH_SIZE, WRITE_SIZE), while the time
    budget lasts; the undistorted pinhole's is the headline kernel's own ratio."""
    for name in ("atan", "radtan"):
        sa = out.get("cameras", {}).get(name, {}).get("sparse_align")
        if not sa or time_left(args.time_budget) < 2 * PMC_PASS_RESERVE_S:
            continue
        try:
            import copy
            a2 = copy.copy(args)
            a2.camera, a2.workload, a2.batch, a2.noise = name, "ref752_4_n200_sparse_align", B, 0.0
            pm = pmc_leg(a2, sa["ms_per_step"], only=("fetch", "write"), k1_kernel="workgroup")
            if pm.get("traffic_bytes_per_launch"):
                sa["roofline"]["traffic"] = pm["traffic_bytes_per_launch"]
                sa["roofline"]["traffic_over_algorithmic"] = pm["traffic_bytes_per_launch"] / sa["roofline"]["algorithmic_bytes_per_launch"]
            else:
                sa["roofline"]["traffic"] = None
                sa["roofline"]["pmc_passes"] = pm.get("passes", pm.get("skipped"))
        except Exception as e:
            sa["roofline"]["traffic"] = None
            sa["roofline"]["pmc_passes"] = repr(e)


def dropin_hip_only(n_frames: int, dump: str, camera: str | None = None, defer_mapper: int = 0) -> dict:
    """child of dropin_sequence (one process per SVO_HIP_MAP_MIRROR mode: the mode is read once): the hip flavour alone"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
    import pypipeline as pp
    cam = synth.Camera(752, 480, 315.5, 315.5, 376.0, 240.0) if camera is None else reference_cameras()[camera]
    T = synth.make_trajectory(n_frames, seed=5, max_step=0.02, max_rot_deg=0.3)
    imgs = synth.render(synth.make_texture(seed=12345), T, cam, device="cuda" if torch.cuda.is_available() else "cpu").cpu().numpy()
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)
    pp.run_sequence("hip", cam, imgs[:10], T[:10])
    st = {}
    hip = pp.run_sequence("hip", cam, imgs, T, stats_out=st, defer_mapper=defer_mapper)
    if dump:
        np.save(dump, np.stack([r["T_f_w"] for r in hip]))
    med = lambda k: float(np.median([r[k] for r in hip[1:]]) * 1e3)
    if camera is not None:  # reference_cameras_leg: the frame's time by API call, the accuracy, the map the run built
        pos = lambda TT: se3.inv(TT)[:, 9:]
        return {"frames": n_frames, "tot_time": med("t_tot_time"), "sparse_img_align": med("t_sparse_img_align"),
                "reproject": med("t_reproject"), "pose_optimizer": med("t_pose_optimizer"),
                "ate_rmse_vs_ground_truth_m": horn_ate(pos(np.stack([r["T_f_w"] for r in hip])), pos(T)),
                "keyframes": int(sum(r["is_keyframe"] for r in hip)), "matches": float(np.median([r["repr_n_new_references"] for r in hip[1:]])),
                "stage_default_frame_frac": float(np.mean([r["stage"] == pp.STAGE_DEFAULT_FRAME for r in hip[1:]]))}
    return {"tot_time": med("t_tot_time"), "reproject": med("t_reproject"), "map_mirror": st.get("map_mirror"),
            "wall_ms_per_frame": st.get("wall_ms_per_frame")}


_CPU_REF: dict = {}  # poses and iteration counts of the CPU reference run (cpu_baseline), for the f64_partials leg

F64_VARIANT_LIB = os.path.join(ROOT, "rpg_svo_amd", "lib", "variants", "libsvo_hip_SIA_F64_PARTIALS.so")


def f64_partials_leg(args, T_default, iters_default, result) -> dict:
    """What reference-width arithmetic costs and buys.  The default K1 forms the per-pixel products and their 16-term
    patch sums in f32 and evaluates SE3::exp as an f32 series (Jacobian rows, partials and reductions are f64 since round
    5); the reference keeps all of it in f64 (sparse_img_align.cpp:228-230,253-258).  The same library built with -DSIA_F64_PARTIALS (built by
    __graft_entry__.build()) does it the reference's way: this leg re-runs the headline workload on it in a child
    process and reports its frames/s and its agreement with the CPU reference next to the default's."""
    import tempfile
    if not os.path.exists(F64_VARIANT_LIB):
        return {"skipped": F64_VARIANT_LIB + " not built (python -m rpg_svo_amd.build -DSIA_F64_PARTIALS)"}
    dump = tempfile.mktemp(suffix=".npz", dir="/tmp")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(min(args.steps, 20)), "--warmup", "3", "--batch", str(args.batch),
           "--workload", args.workload, "--noise", str(args.noise), "--n-iter", str(args.n_iter), "--no-cpu-baseline",
           "--extras", "none", "--pmc-child", "f64", "--k1-kernel", "workgroup", "--dump-result", dump]
    env = dict(os.environ, SVO_HIP_LIB=F64_VARIANT_LIB)
    env.pop("SVO_BENCH_FORCE_DIST", None)
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    if p.returncode != 0:
        return {"skipped": f"child rc={p.returncode}: {p.stderr[-300:]}"}
    child = json.loads(p.stdout.strip().splitlines()[-1])
    z = np.load(dump)
    os.unlink(dump)
    T64, it64 = z["T_est_w"], z["iters"]
    out = {"build": "-DSIA_F64_PARTIALS: f64 Jacobian rows, per-pixel products, per-lane partials, wave reductions and SE3::exp "
                    "(142 VGPRs, 3 waves/SIMD instead of 128 / 4)",
           "frames_per_s": child["value"], "kernel_ms": child["roofline"]["ms"],
           "frames_per_s_sparse_align_only": child.get("value_sparse_align_only"),
           "what_frames_per_s_is": "the headline step (K1 + K4 when --pipeline refine) on the reference-width library",
           "frames_per_s_default": result["value"], "kernel_ms_default": result["roofline"]["ms"],
           "slowdown": result["value"] / child["value"],
           "vs_default_kernel": {"se3_lognorm_max": float(se3.log_norm(T64, T_default).max()),
                                 "same_iteration_counts_frac": float(np.mean((it64 == iters_default).all(1)))}}
    # its own roofline line: the child's HIP-event average over its timed launches, its own algorithmic bytes (its own
    # iteration counts), and the counter traffic of two more children under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE)
    rl = dict(child["roofline"])
    try:
        pm = pmc_leg(args, child["roofline"]["ms"], only=("fetch", "write"), lib_path=F64_VARIANT_LIB, k1_kernel="workgroup")
        rl["traffic"] = pm.get("traffic_bytes_per_launch")
        if rl["traffic"]:
            rl["traffic_over_algorithmic"] = rl["traffic"] / rl["algorithmic_bytes_per_launch"]
        rl["pmc_passes"] = pm.get("passes")
    except Exception as e:
        rl["traffic"] = None
        rl["pmc_passes"] = repr(e)
    out["roofline"] = rl
    if _CPU_REF:
        S = len(_CPU_REF["iters"])
        for name, T, it in (("f64_partials", T64, it64), ("default", T_default, iters_default)):
            d = se3.log_norm(T[:S], _CPU_REF["T"])
            out[f"{name}_vs_cpu_{_CPU_REF['which']}"] = {
                "frames_compared": S, "se3_lognorm_max": float(d.max()), "se3_lognorm_median": float(np.median(d)),
                "same_iteration_counts_frac": float(np.mean([np.array_equal(a, b) for a, b in zip(_CPU_REF["iters"], it[:S])]))}
    return out


def cpu_pose_refine(W: Workload, refine, T_start_w: np.ndarray, which: str, k: int):
    """pose_optimizer::optimizeGaussNewton of the reference (oracle/_ref; the C port where it is absent) on the first k
    frames of the headline step, one thread, from the CPU's own K1 poses, on the matches the timed step refines on.
    Only the calls themselves are timed (arguments converted beforehand)."""
    from oracle import pytrack
    trk = pytrack.Track("ref" if which.startswith("ref") and pytrack.ref_available() else "orc")
    pc = pytrack.make_cam(W.cam)
    f = refine.f_new[:k].cpu().numpy()
    lv = refine.level[:k].cpu().numpy().astype(np.int32)
    ok = refine.okb[:k].cpu().numpy().astype(np.uint8)
    pos = refine.pt_pos[:k].cpu().numpy()
    N = f.shape[1]
    T_out = np.zeros((k, 12))
    seconds = 0.0
    for b in range(k):
        fb, lb, hb, pb = np.ascontiguousarray(f[b]), np.ascontiguousarray(lv[b]), ok[b].copy(), np.ascontiguousarray(pos[b])
        Tb = np.ascontiguousarray(T_start_w[b], dtype=np.float64)
        res = pytrack.PoseOptResult()
        call = (C.c_double(2.0), C.c_int(10), C.byref(pc), Tb.ctypes.data_as(C.c_void_p), C.c_int(N), fb.ctypes.data_as(C.c_void_p),
                lb.ctypes.data_as(C.c_void_p), hb.ctypes.data_as(C.c_void_p), pb.ctypes.data_as(C.c_void_p), C.byref(res))
        t0 = time.perf_counter()
        trk._pose_optimize(*call)
        seconds += time.perf_counter() - t0
        T_out[b] = np.array(res.T_f_w[:])
    return T_out, seconds


def cpu_baseline(args, W: Workload, T_est_w, result, iters_gpu, refine=None, po_last=None) -> dict:
    """Times the reference's own SparseImgAlign translation unit (oracle/_ref, kind "reference";
    the C port where that library is absent) on a bounded sample of the same problems, on this
    box's host cores, and fills result["parity"] from the same run."""
    from oracle import pyoracle
    S = min(args.cpu_sample, W.B)
    imgs = W.images[:S + 1].cpu().numpy()
    pyrs = [pyoracle.create_img_pyramid(im, W.n_levels, pyoracle.HALFSAMPLE_AUTO) for im in imgs]
    rs = np.arange(S, dtype=np.int32)
    cs = rs + 1
    nn = np.full(S, W.n_patches, dtype=np.int32)
    px = W.px_all[:S].cpu().numpy()
    f = W.f_all[:S].cpu().numpy()
    pos = W.pos_all[:S].cpu().numpy()
    hp = np.ones((S, W.n_patches), dtype=np.uint8)
    cores = os.cpu_count() or 1
    s1 = min(S, 2048)
    # kind "reference": the reference's own SparseImgAlign translation unit (oracle/_ref, built
    # from /root/reference/svo/src in the build container, travels prebuilt); otherwise the C port
    which = "ref" if pyoracle.ref_available() else "orc"

    def timed(k, threads):
        tm = {}
        t0 = time.perf_counter()
        T, r = pyoracle.sparse_img_align_batch(pyrs, rs[:k], cs[:k], W.cam, W.T_ref_w[:k], W.T_prior_w[:k], nn[:k], px[:k],
                                               f[:k], hp[:k], pos[:k], W.max_level, W.min_level, args.n_iter,
                                               n_threads=threads, which=which, timing=tm)
        return T, r, tm.get("run_seconds", time.perf_counter() - t0)

    _, _, t1 = timed(s1, 1)
    T_cpu, res, tn = timed(S, cores)
    best_threads, best_rate = cores, S / tn
    sweep = {str(cores): S / tn}
    for th in (cores // 2, cores // 4):  # SMT siblings / memory bandwidth: fewer threads can be faster
        if th >= 1:
            k = max(1, S // 2)
            _, _, tt = timed(k, th)
            sweep[str(th)] = k / tt
            if k / tt > best_rate:
                best_threads, best_rate = th, k / tt
    _CPU_REF.update(T=T_cpu, iters=[r["iters"] for r in res], which=which)
    d = se3.log_norm(T_est_w[:S], T_cpu)
    pos_gpu = se3.inv(T_est_w[:S])[:, 9:]
    pos_cpu = se3.inv(T_cpu)[:, 9:]
    result["parity"] = {
        "frames_compared": int(S), "se3_lognorm_max": float(d.max()), "se3_lognorm_median": float(np.median(d)),
        "ate_rmse_vs_cpu_m": horn_ate(pos_gpu, pos_cpu),
        "same_iteration_counts_frac": float(np.mean([np.array_equal(r["iters"], it) for r, it in zip(res, iters_gpu[:S])])),
        "against": "reference" if which == "ref" else "port",
    }
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    impl = ("the reference's own sparse_img_align.cpp (oracle/_ref/libsvo_ref.so, g++ -O3 against dependency shims), "
            "run() calls only") if which == "ref" else "oracle/libsvo_oracle.so (C port, gcc -O3)"
    # the same translation unit built with the reference's own release flags (svo/CMakeLists.txt:34-45: -O3 -funroll-loops
    # -fno-signed-zeros ..., -march=native as x86-64-v3; oracle/Makefile target ref_release): what the CPU path does when it
    # is built for speed, not for bit-comparability -- one thread and the best thread count of the sweep above
    release = {}
    if which == "ref" and pyoracle.ref_release_available():
        try:
            which_saved, which = which, "ref_release"
            _, _, t1r = timed(s1, 1)
            Tr_, _, tnr = timed(S, best_threads)
            which = which_saved
            release = {"value_release_flags": s1 / t1r, "value_release_flags_best_threads": S / tnr,
                       "release_flags": "g++ -O3 -march=x86-64-v3 -funroll-loops -fno-signed-zeros -fomit-frame-pointer -fsee "
                                        "-fno-math-errno (svo/CMakeLists.txt:34-45 with -march=native -> x86-64-v3)",
                       "release_vs_bit_comparable_build_se3_lognorm_max": float(se3.log_norm(Tr_, T_cpu).max())}
        except Exception as e:
            which = which_saved
            release = {"value_release_flags": None, "release_flags_skipped": repr(e)}
    # headline: ONE core, the reference's own execution model (tracking is single-threaded, frame_handler_mono.cpp);
    # the thread sweep over frame pairs is the throughput comparison for batched replay and sits beside it
    value, k4 = s1 / t1, {}
    if refine is not None:
        # the step the headline times is K1 + K4: the reference's pose_optimizer.cpp on the same sample, same thread
        try:
            T_ref4, t4 = cpu_pose_refine(W, refine, T_cpu, which, s1)
            value = s1 / (t1 + t4)
            k4 = {"value_sparse_align_only": s1 / t1, "pose_optimize_us_per_frame": t4 / s1 * 1e6,
                  "sparse_align_us_per_frame": t1 / s1 * 1e6}
            if po_last is not None:
                d4 = se3.log_norm(po_last.T_f_w[:s1].cpu().numpy(), T_ref4)
                result["parity"].update(refined_pose_se3_lognorm_max=float(d4.max()), refined_pose_se3_lognorm_median=float(np.median(d4)),
                                        refined_frames_compared=int(s1))
            if release.get("value_release_flags"):
                # (K4 stays the bit-comparable build's: the release library holds SparseImgAlign only)
                release["value_release_flags_sparse_align_only"] = release["value_release_flags"]
                release["value_release_flags"] = s1 / (s1 / release["value_release_flags"] + t4)
        except Exception as e:
            k4 = {"pose_optimize_skipped": repr(e)}
    step = "run() + optimizeGaussNewton() calls" if refine is not None and "pose_optimize_us_per_frame" in k4 else "run() calls"
    return {**release, **k4, "value": value, "unit": "frames/s", "cores": 1, "kind": "reference" if which == "ref" else "port",
            "sample": f"{s1} of the benchmark's own frame pairs on one thread, {impl}".replace("run() calls", step),
            "sample_short": f"{s1} of the same frame pairs, 1 thread, " + ("reference's own sparse_img_align.cpp" if which == "ref" else "C port")
                            + (" + pose_optimizer.cpp" if "pose_optimize_us_per_frame" in k4 else ""),
            "value_best_threads": best_rate, "best_threads": best_threads,
            "sample_threads": f"{S} frame pairs over all / half / a quarter of the logical CPUs",
            "frames_per_s_by_threads": sweep, "host_logical_cpus": cores, "cpu_model": model}


class FullTrack:
    """BASELINE configs[2]: what FrameHandlerMono::processFrame does after sparse alignment
    (frame_handler_mono.cpp:145-190) plus the depth-filter update of the mapping thread, for
    the same B replay frames: reproject map points, findMatchDirect (affine warp + align2D),
    pose_optimizer::optimizeGaussNewton, DepthFilter::updateSeeds.

    mode "representative" (the `full_track` leg) shapes the work like the reference's own trace
    (svo/test/benchmark.csv:14-306: ~190 trials -> ~120 matches per frame) and like its depth filter, where a seed
    is updated by EVERY later frame until it converges (depth_filter.cpp:197-291):
      * map points of problem b are the features of the keyframe 8 frames back, observed a second time in the
        keyframe 16 frames back (Point::getCloseViewObs picks between them).  65 % are good points with 7 % depth
        error: the trial starts f * baseline/depth * 0.07 ~ 2-3 px off and alignment needs several evaluations.
        35 % are bad points (in the reference: unconverged candidates, wrong matches of the depth filter) whose depth
        is 25-60 % off: they project many pixels away, alignment walks its 10 iterations and mostly fails -- the
        reference trace's 37 % failing trials.  Points that left the image are not tried (Reprojector::reprojectPoint);
      * the seeds of problem b are what the depth filter holds when frame b+1 arrives.  Every frame starts seeds
        (SEEDS_MATCHED + SEEDS_UNMATCHED of its features).  A "matched" seed has been updated by every frame since
        (ages 1..12: its state is produced by running the update kernel itself over those earlier frames during set-up)
        and leaves when it converges; sigma shrinks as the baseline grows, so its epipolar segment stays a few
        pixels long.  An "unmatched" seed has failed every search so far -- only b was incremented
        (depth_filter.cpp:238-245), sigma is still the initial one -- and stays until its keyframe batch is three
        batches old (:216-219; ages 1..30 here): its segment grows with the baseline, tens to hundreds of ZMSSD
        positions (matcher.cpp:248-291).  30 % of new seeds are taken to be of that kind; they are about half of
        the live population and nearly all of the scan work.
    mode "easy" (`full_track_easy`) is the workload of rounds 1-2, kept for continuity: points and seeds of the
    previous frame only, 1 % depth error (every trial matches after 1.65 evaluations, the scan hardly moves)."""

    STAGES = ("compose_pose", "reproject", "find_match_direct", "cam2world", "pose_optimize", "update_seeds")
    KF1, KF2 = 8, 16          # representative: keyframes the map points are observed in, frames back from b
    DEPTH_NOISE = 0.07        # ... relative depth error of the good points
    BAD_POINTS = 0.35         # ... share of points whose depth is grossly wrong (25-60 %)
    SEED_AGES = 12            # matched seeds: created 0..11 frames before frame b, updated by every frame since
    SEEDS_MATCHED = 32        # ... of them per frame
    SEED_AGES_UNMATCHED = 30  # unmatched seeds: every search failed so far; kept for three keyframe batches
    SEEDS_UNMATCHED = 14      # ... of them per frame (30 % of the new seeds)

    def __init__(self, W: Workload, dev, rank, mode: str = "representative", with_seeds: bool = True):
        from rpg_svo_amd import tracking
        assert mode in ("representative", "easy")
        self.tr = tracking
        self.W = W
        self.mode = mode
        cam, store = W.cam, W.store
        self.cam, self.store, self.dev = cam, store, dev
        B, N = W.B, W.n_patches
        self.B, self.N = B, N
        px_all, f_all, pos_all = W.px_all, W.f_all, W.pos_all
        g = torch.Generator().manual_seed(4242 + rank)
        T = torch.as_tensor(W.T_gt, dtype=torch.float64, device=dev)
        # frame table: rows 0..B = replay frames with their (ground truth) keyframe poses,
        # rows B+1+b = frame b+1 as the frame being tracked (pose = sparse-align output)
        slot = torch.cat([torch.arange(0, B + 1, dtype=torch.int32, device=dev),
                          torch.arange(1, B + 1, dtype=torch.int32, device=dev)])
        self.frame_T = torch.cat([T, T[1:B + 1].clone()]).contiguous()
        self.frames = tracking.FrameTable(slot, self.frame_T)
        self.T_ref = T[:B].contiguous()
        self.cur_rows = torch.arange(B + 1, 2 * B + 1, dtype=torch.int32, device=dev)
        M = B * N
        self.M = M
        self.cur_frame = self.cur_rows.repeat_interleave(N).contiguous()
        b_idx = torch.arange(B, device=dev)
        centre = lambda rows: -(T[rows, :9].reshape(-1, 3, 3).transpose(1, 2) @ T[rows, 9:, None])[..., 0]

        def project(rows, pts):  # pts [B,N,3] into frames `rows` [B]: pixels (through the camera's model), depth
            pc = (T[rows, :9].reshape(-1, 3, 3)[:, None] @ pts[..., None])[..., 0] + T[rows, None, 9:]
            return torch.stack(synth.cam_distort(cam, pc[..., 0] / pc[..., 2], pc[..., 1] / pc[..., 2]), -1), pc[..., 2]

        def inside(px, z, border):
            return (z > 0) & (px[..., 0] >= border) & (px[..., 0] < cam.width - border) & (px[..., 1] >= border) & (px[..., 1] < cam.height - border)

        # ---- map points and their observations -------------------------------------------------------
        if mode == "easy":
            src, older = b_idx, (b_idx - 2).clamp(min=0)
            depth_noise, has_older = 0.01, b_idx >= 2
        else:
            src, older = (b_idx - self.KF1).clamp(min=0), (b_idx - self.KF2).clamp(min=0)
            depth_noise, has_older = self.DEPTH_NOISE, b_idx >= self.KF2
        self.src, self.older = src, older
        pos_src, px_src, f_src = pos_all[src], px_all[src], f_all[src]
        c_src = centre(src)
        ray = pos_src - c_src[:, None, :]
        noise = 1.0 + depth_noise * torch.randn(B, N, 1, generator=g, dtype=torch.float64).to(dev)
        if mode != "easy":
            bad = (torch.rand(B, N, 1, generator=g) < self.BAD_POINTS).to(dev)
            gross = (0.25 + 0.35 * torch.rand(B, N, 1, generator=g, dtype=torch.float64)) * (2.0 * (torch.rand(B, N, 1, generator=g) < 0.5) - 1.0)
            noise = torch.where(bad, 1.0 + gross.to(dev), noise)
            self.bad_point = bad.reshape(M)
        self.pt_pos = (c_src[:, None, :] + ray * noise).reshape(M, 3).contiguous()
        # Reprojector::reprojectPoint (reprojector.cpp:206-217): only points inside the frame (8 px border) are
        # tried; decided here with the ground-truth pose of the tracked frame (a point without observations is
        # skipped by the kernels)
        if mode == "easy":
            in_cur = torch.ones(B, N, dtype=torch.bool, device=dev)
        else:
            pxc, zc = project(b_idx + 1, self.pt_pos.view(B, N, 3))
            in_cur = inside(pxc, zc, 8)
        self.in_cur = in_cur
        px2, z2 = project(older, pos_src)
        has2 = has_older[:, None] & inside(px2, z2, 12) & in_cur
        n_obs = (in_cur.to(torch.int32) + has2.to(torch.int32)).reshape(M)
        ptr = torch.zeros(M + 1, dtype=torch.int32, device=dev)
        ptr[1:] = torch.cumsum(n_obs, 0)
        n_total = int(ptr[-1].item())
        first = ptr[:-1].long()[in_cur.reshape(M)]
        o_frame = torch.zeros(n_total, dtype=torch.int32, device=dev)
        o_px = torch.zeros(n_total, 2, dtype=torch.float64, device=dev)
        o_f = torch.zeros(n_total, 3, dtype=torch.float64, device=dev)
        o_frame[first] = src.repeat_interleave(N).to(torch.int32)[in_cur.reshape(M)]
        o_px[first] = px_src.reshape(M, 2)[in_cur.reshape(M)]
        o_f[first] = f_src.reshape(M, 3)[in_cur.reshape(M)]
        sec = ptr[:-1].long()[has2.reshape(M)] + 1
        o_frame[sec] = older.repeat_interleave(N)[has2.reshape(M)].to(torch.int32)
        o_px[sec] = px2.reshape(M, 2)[has2.reshape(M)]
        d2 = torch.stack([*synth.cam_undistort(cam, o_px[sec][:, 0], o_px[sec][:, 1]),
                          torch.ones(len(sec), dtype=torch.float64, device=dev)], -1)
        o_f[sec] = d2 / d2.norm(dim=-1, keepdim=True)
        self.obs_ptr = ptr
        self.obs = tracking.FeatureSet(frame=o_frame, level=torch.zeros(n_total, dtype=torch.int32, device=dev),
                                       px=o_px, f=o_f)
        self.matcher = tracking.Matcher(align_max_iter=10, n_pyr_levels=W.n_levels)
        self.n = torch.full((B,), N, dtype=torch.int32, device=dev)
        self.df = tracking.DepthFilter(n_pyr_levels=W.n_levels)
        self.f_new = torch.empty(M, 3, dtype=torch.float64, device=dev)
        # result blocks reused by every step (no allocation / memset inside the timed stages)
        self.cell_px = (torch.zeros(M, dtype=torch.int32, device=dev), torch.zeros(M, 2, dtype=torch.float64, device=dev))
        self.match = self.matcher.alloc_result(M, dev)
        self.okb = torch.zeros(B, N, dtype=torch.uint8, device=dev)
        self.events = []
        self.last = {}
        self.S = 0
        if not with_seeds:  # the headline's set-up (RefineStep): only the matcher's stages are run
            return
        # ---- seeds ---------------------------------------------------------------------------------
        if mode == "easy":
            # one per reference feature, inverse depth known to 10 %, range from 0.6 x depth
            depth = ray.norm(dim=-1).reshape(M)
            dm = depth * (1.0 + 0.1 * torch.randn(M, generator=g, dtype=torch.float64).to(dev))
            z_range = (1.0 / (0.6 * depth)).float()
            self.seed0 = dict(a=torch.full((M,), 10.0, device=dev), b=torch.full((M,), 10.0, device=dev),
                              mu=(1.0 / dm).float(), z_range=z_range, sigma2=z_range * z_range / 36.0)
            self.seed_ftr = tracking.FeatureSet(frame=b_idx.repeat_interleave(N).to(torch.int32).contiguous(),
                                                level=torch.zeros(M, dtype=torch.int32, device=dev),
                                                px=px_all.reshape(M, 2).contiguous(), f=f_all.reshape(M, 3).contiguous())
            self.seed_cur = self.cur_frame
            self.seed_age = torch.ones(M, dtype=torch.int32, device=dev)
            self.seed_frame_of = b_idx.repeat_interleave(N)
            self.seed_unmatched = torch.zeros(M, dtype=torch.bool, device=dev)
        else:
            self._make_seed_population(T, g)
        S = self.seed0["mu"].shape[0]
        self.S = S
        self.seeds = tracking.SeedSet(**{k: v.clone() for k, v in self.seed0.items()},
                                      batch_id=torch.zeros(S, dtype=torch.int32, device=dev))
        self.po = tracking.PoseOptResult(torch.empty(B, 12, dtype=torch.float64, device=dev),
                                         torch.zeros(B, 36, dtype=torch.float64, device=dev),
                                         torch.zeros(B, 4, dtype=torch.float64, device=dev),
                                         torch.zeros(B, dtype=torch.int32, device=dev),
                                         torch.empty(B, N, dtype=torch.uint8, device=dev))
        self.seed_out = (torch.zeros(S, dtype=torch.int32, device=dev), torch.zeros(S, 3, dtype=torch.float64, device=dev),
                         torch.zeros(S, 2, dtype=torch.float64, device=dev))

    def _make_seed_population(self, T, g):
        """Seeds as the depth filter holds them while frame b+1 arrives: created at frames b, b-1, ... (ages 1, 2, ...)
        by DepthFilter::initializeSeeds (Seed ctor, depth_filter.cpp:37-46: a = b = 10, mu = 1/depth_mean,
        z_range = 1/depth_min with depth_min = half the closest scene depth, sigma2 = z_range^2/36) and since updated
        with every frame in between.  Those earlier updates are run here, with the update kernel and the ground-truth
        poses: state[j] = the seeds of every frame after j updates.  A seed leaves the population when it converges,
        turns NaN (depth_filter.cpp:261-287) or, in this replay, is no longer visible."""
        tr, W, dev, cam = self.tr, self.W, self.dev, self.cam
        B, Nm, Nu, A, Au = self.B, self.SEEDS_MATCHED, self.SEEDS_UNMATCHED, self.SEED_AGES, self.SEED_AGES_UNMATCHED
        Ns = Nm + Nu
        sel = torch.randperm(W.n_patches, generator=g)[:Ns].sort().values.to(dev)
        px = W.px_all[:, sel].contiguous()      # [B,Ns,2] seed features of frame r: the first Nm matched, the rest unmatched
        f = W.f_all[:, sel].contiguous()
        pos = W.pos_all[:, sel]
        c = -(T[:B, :9].reshape(B, 3, 3).transpose(1, 2) @ T[:B, 9:, None])[..., 0]
        depth_all = (W.pos_all - c[:, None, :]).norm(dim=-1)   # scene depth of the keyframe (frame_utils::getSceneDepth)
        depth_mean, depth_min = depth_all.mean(1, keepdim=True), 0.5 * depth_all.min(1, keepdim=True).values
        z_range = (1.0 / depth_min).expand(B, Ns).float()
        state = dict(a=torch.full((B, Ns), 10.0, device=dev), b=torch.full((B, Ns), 10.0, device=dev),
                     mu=(1.0 / depth_mean).expand(B, Ns).float().contiguous(), z_range=z_range.contiguous(),
                     sigma2=(z_range * z_range / 36.0).contiguous())
        is_matched = (torch.arange(Ns, device=dev) < Nm)[None, :].expand(B, Ns)
        alive = torch.ones(B, Ns, dtype=torch.bool, device=dev)
        r_idx = torch.arange(B, device=dev)
        ftr = tr.FeatureSet(frame=r_idx.repeat_interleave(Ns).to(torch.int32).contiguous(),
                            level=torch.zeros(B * Ns, dtype=torch.int32, device=dev),
                            px=px.reshape(-1, 2).contiguous(), f=f.reshape(-1, 3).contiguous())
        # the population of problem b at age a is the state of the seeds of frame r = b+1-a after their a-1 earlier frames
        parts = {k: [] for k in ("a", "b", "mu", "z_range", "sigma2")}
        part_ftr, part_cur, part_age, part_b, part_kind = [], [], [], [], []
        GONE = (capi.SEED_CONVERGED, capi.SEED_NAN, capi.SEED_ERASED_OLD, capi.SEED_NOT_IN_FRAME, capi.SEED_BEHIND)
        init = {k: v.clone() for k, v in state.items()}
        for j in range(max(A, Au)):  # j = frames seen so far = age - 1
            b_of_r = r_idx + j          # problem whose tracked frame b+1 = r + j + 1 gives these seeds their next update
            in_pop = (is_matched & alive & (j < A)) | (~is_matched & (j < Au))
            keep = (in_pop & (b_of_r < B)[:, None]).reshape(-1).nonzero()[:, 0]
            um = (~is_matched).reshape(-1)[keep]
            for k in parts:
                v = state[k].reshape(-1)[keep].clone()
                if k == "b":  # an unmatched seed: b++ for every failed search (depth_filter.cpp:238-245), the rest untouched
                    v = torch.where(um, init["b"].reshape(-1)[keep] + float(j), v)
                elif k != "z_range":
                    v = torch.where(um, init[k].reshape(-1)[keep], v)
                parts[k].append(v)
            part_ftr.append(keep)
            part_cur.append((B + 1 + b_of_r).repeat_interleave(Ns)[keep].to(torch.int32))
            part_age.append(torch.full((len(keep),), j + 1, dtype=torch.int32, device=dev))
            part_b.append(b_of_r.repeat_interleave(Ns)[keep])
            part_kind.append(um)
            if j + 1 < A:
                # the next update of every frame's (matched) seeds, against frame r+j+1 at its ground-truth pose (rows 0..B)
                cur = (r_idx + j + 1).clamp(max=B).repeat_interleave(Ns).to(torch.int32).contiguous()
                seeds = tr.SeedSet(**{k: v.reshape(-1).contiguous() for k, v in state.items()},
                                   batch_id=torch.zeros(B * Ns, dtype=torch.int32, device=dev))
                status, _, _ = self.df.update_seeds(self.store, cam, self.frames, cur, ftr, seeds, 0)
                state = {k: getattr(seeds, k).view(B, Ns) for k in state}
                st = status.view(B, Ns)
                gone = torch.zeros_like(alive)
                for code in GONE:
                    gone |= st == code
                alive = alive & ~gone
        order = torch.argsort(torch.cat(part_b), stable=True)  # seeds of one problem next to each other (as one frame's list)
        cat = lambda xs: torch.cat(xs)[order].contiguous()
        self.seed0 = {k: cat(v) for k, v in parts.items()}
        fi = cat(part_ftr)
        self.seed_ftr = tr.FeatureSet(frame=ftr.frame[fi].contiguous(), level=torch.zeros(len(fi), dtype=torch.int32, device=dev),
                                      px=ftr.px[fi].contiguous(), f=ftr.f[fi].contiguous())
        self.seed_cur = cat(part_cur)
        self.seed_age = cat(part_age)
        self.seed_frame_of = cat(part_b)
        self.seed_unmatched = cat(part_kind)

    def match_stage(self, T_cur_from_ref, mk=lambda: None):
        """K1's pose -> compose -> Reprojector::reprojectPoint -> Matcher::findMatchDirect (K2 + K3) -> cam2world: the
        observations pose_optimizer::optimizeGaussNewton reads (f_new, search_level, okb)"""
        tr = self.tr
        B, N = self.B, self.N
        tr.compose_poses(T_cur_from_ref, self.T_ref, out=self.frame_T, out_index=self.cur_rows)
        mk()
        cell, px = tr.reproject_points(self.cam, self.frames, self.cur_frame, self.pt_pos, 30, (self.cam.width + 29) // 30,
                                       out=self.cell_px)
        mk()
        m = self.matcher.find_match_direct(self.store, self.cam, self.frames, self.cur_frame, self.pt_pos, self.obs_ptr,
                                           self.obs, px, out=self.match)
        mk()
        tr.cam2world(self.cam, m.px_cur, out=self.f_new)
        mk()
        torch.gt(m.ok.view(B, N), 0, out=self.okb.view(torch.bool))
        return m, px

    def step(self, T_cur_from_ref, ev: Events | None):
        tr = self.tr
        B, N = self.B, self.N
        mk = (lambda: self.events.append(ev.mark())) if ev is not None else (lambda: None)
        mk()
        m, px = self.match_stage(T_cur_from_ref, mk)
        po = tr.optimize_gauss_newton(self.cam, self.n, self.f_new.view(B, N, 3), m.search_level.view(B, N),
                                      self.pt_pos.view(B, N, 3), self.okb, self.frame_T[B + 1:], 2.0, 10, out=self.po)
        self.frame_T[B + 1:].copy_(po.T_f_w)  # the mapper sees the refined pose (frame_handler_mono.cpp:190,221)
        mk()
        for k, v in self.seed0.items():
            getattr(self.seeds, k).copy_(v)
        status, _, _ = self.df.update_seeds(self.store, self.cam, self.frames, self.seed_cur, self.seed_ftr, self.seeds, 0,
                                            out=self.seed_out)
        mk()
        self.last = dict(match=m, pose=po, seed_status=status, px_proj=px)

    def step_mapper_overlapped(self, T_cur_from_ref, s_map, pose_ready, mapper_done):
        """The same step with DepthFilter::updateSeeds on a stream of its own -- the reference runs its depth filter in a
        thread of its own (depth_filter.cpp:75-93, startThread) while the tracker is on the next frame: the mapper's update
        of step i runs beside sparse alignment, matching and pose refinement of step i + 1.  The mapper reads a frame table
        of its own (frame_T_mapper), which the tracker fills with the refined poses once the mapper's previous update has
        finished with it; nothing else is shared between the two (seed state and the update's workspace are the mapper's,
        matches and poses the tracker's)."""
        tr = self.tr
        B, N = self.B, self.N
        if not hasattr(self, "frames_mapper"):
            self.frame_T_mapper = self.frame_T.clone()
            self.frames_mapper = tr.FrameTable(self.frames.slot, self.frame_T_mapper)
        m, px = self.match_stage(T_cur_from_ref)
        po = tr.optimize_gauss_newton(self.cam, self.n, self.f_new.view(B, N, 3), m.search_level.view(B, N),
                                      self.pt_pos.view(B, N, 3), self.okb, self.frame_T[B + 1:], 2.0, 10, out=self.po)
        trk = torch.cuda.current_stream(self.dev)
        trk.wait_event(mapper_done)              # (the mapper's previous update has read its table)
        self.frame_T_mapper[B + 1:].copy_(po.T_f_w)
        pose_ready.record(trk)
        with torch.cuda.stream(s_map):
            s_map.wait_event(pose_ready)
            for k, v in self.seed0.items():
                getattr(self.seeds, k).copy_(v)
            status, _, _ = self.df.update_seeds(self.store, self.cam, self.frames_mapper, self.seed_cur, self.seed_ftr, self.seeds, 0,
                                                out=self.seed_out)
            mapper_done.record(s_map)
        return status

    def stage_ms(self, ev: Events):
        k = len(self.STAGES) + 1
        acc = {s: [] for s in self.STAGES}
        for i in range(0, len(self.events) - k + 1, k):
            for j, sname in enumerate(self.STAGES):
                acc[sname].append(ev.ms(self.events[i + j], self.events[i + j + 1]))
        return {s: float(np.mean(v)) for s, v in acc.items() if v}

    def describe(self):
        m, po, st = self.last["match"], self.last["pose"], self.last["seed_status"].cpu().numpy()
        T_est = po.T_f_w.cpu().numpy()
        names = {capi.SEED_ERASED_OLD: "erased_old", capi.SEED_BEHIND: "behind", capi.SEED_NOT_IN_FRAME: "not_in_frame",
                 capi.SEED_NO_MATCH: "no_match", capi.SEED_UPDATED: "updated", capi.SEED_CONVERGED: "converged",
                 capi.SEED_NAN: "nan"}
        age = self.seed_age.cpu().numpy()
        return {"mode": self.mode,
                "match_trials_per_frame": float(self.in_cur.float().sum().item() / self.B),
                "matches_per_frame": float(m.ok.float().sum().item() / self.B),
                **({"matches_per_frame_among_bad_points": float((m.ok > 0)[self.bad_point].float().sum().item() / self.B),
                    "bad_points_tried_per_frame": float((self.bad_point & self.in_cur.reshape(-1)).float().sum().item() / self.B)}
                   if self.mode != "easy" else {}),
                "pose_refine_obs_after_pruning": float(po.stats[:, 3].mean().item()),
                "seeds_per_frame": self.S / self.B,
                "unmatched_seeds_per_frame": float(self.seed_unmatched.float().sum().item() / self.B),
                "seeds_per_frame_by_age": {str(a): float((age == a).sum() / self.B) for a in np.unique(age)},
                "seed_status_per_frame": {names.get(int(k), str(k)): float((st == k).sum() / self.B) for k in np.unique(st)},
                "pipeline": "sparse_align -> reproject -> findMatchDirect -> pose_optimize -> updateSeeds",
                "_T_refined": T_est}

    def stage_rooflines(self, lib, stages: dict) -> dict:
        """Per stage: algorithmic bytes of one step (SURVEY 8d byte formulas) / measured stage time.
          findMatchDirect (K2+K3): per trial 121 B template footprint + 100 B warped patch + 48 B geometry
                                   + 81 B per alignment evaluation (9x9 window)
          pose refine (K4):        M*52 + 416 B per frame
          depth update (K5):       36 B seed state + 64 B per scanned epipolar position + the template"""
        dev, M, B, N = self.dev, self.M, self.B, self.N
        m = self.last["match"]
        stream = torch.cuda.current_stream(dev).cuda_stream
        # alignment evaluations per trial: K3 alone (instrumented variant) on the patches / start positions of the step
        lvl = m.search_level
        scale = (1 << lvl.long()).double()[:, None]
        px0 = (self.last["px_proj"] / scale).contiguous()
        slot = self.frames.slot[self.cur_frame.long()].contiguous()
        ok = torch.zeros(M, dtype=torch.int32, device=dev)
        evals = torch.zeros(M, dtype=torch.int32, device=dev)
        capi.check(lib.svo_hip_align_batch_counted(C.byref(self.store.layout), self.store.ptr, M, slot.data_ptr(), lvl.data_ptr(),
                                                   m.patch_with_border.data_ptr(), None, None, 10, px0.data_ptr(), ok.data_ptr(),
                                                   None, evals.data_ptr(), stream), "svo_hip_align_batch_counted")
        tried = m.ref_obs >= 0
        n_eval = float(evals[tried].sum().item())
        n_tried = float(tried.sum().item())
        # a wave of K3 runs as long as its slowest trial: mean over the 64-trial groups of the group maximum
        ev64 = evals[: (M // 64) * 64].view(-1, 64)
        eval_hist = torch.bincount(evals[tried].long(), minlength=12)[:12].tolist()
        wave_max_mean = float(ev64.max(dim=1).values.float().mean().item())
        fm_bytes = n_tried * (121 + 100 + 48) + 81.0 * n_eval
        S = self.S
        steps_ptr = lib.svo_hip_update_seeds_scan_steps(self.df.last_workspace.data_ptr())
        scan = torch.empty(S, dtype=torch.int32, device=dev)
        capi.check(lib.svo_hip_memcpy_d2d(scan.data_ptr(), steps_ptr, S * 4, stream), "svo_hip_memcpy_d2d")
        torch.cuda.synchronize()
        n_scan = float(scan.sum().item())
        # residual evaluations of the depth filter's sub-pixel alignment: one more update with the instrumented alignment
        # kernel (svo_hip_update_seeds_count_evaluations; outside every timed region, the same seeds, the same results)
        seed_evals = None
        try:
            lib.svo_hip_update_seeds_count_evaluations(1)
            for k, v in self.seed0.items():
                getattr(self.seeds, k).copy_(v)
            self.df.update_seeds(self.store, self.cam, self.frames, self.seed_cur, self.seed_ftr, self.seeds, 0, out=self.seed_out)
            ev_ptr = lib.svo_hip_update_seeds_align_evaluations(self.df.last_workspace.data_ptr(), S)
            evs = torch.empty(S, dtype=torch.int32, device=dev)
            capi.check(lib.svo_hip_memcpy_d2d(evs.data_ptr(), ev_ptr, S * 4, stream), "svo_hip_memcpy_d2d")
            torch.cuda.synchronize()
            seed_evals = {"aligned_seeds": float((evs > 0).sum().item()), "evaluations": float(evs.sum().item()),
                          "evaluations_per_wave_of_64_seeds": float(evs[: (S // 64) * 64].view(-1, 64).max(dim=1).values.float().mean().item()),
                          "evaluations_histogram": torch.bincount(evs[evs > 0].long(), minlength=12)[:12].tolist()}
        except Exception as e:
            seed_evals = {"skipped": repr(e)}
        finally:
            lib.svo_hip_update_seeds_count_evaluations(0)
        seed_bytes = S * 36.0 + 64.0 * n_scan + S * (121.0 + 100.0)
        edges = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1001]
        scan_hist = {f"{lo}..{hi - 1}": float(((scan >= lo) & (scan < hi)).sum().item() / self.B) for lo, hi in zip(edges[:-1], edges[1:])}
        scan_by_age = {str(int(a)): float(scan[self.seed_age == a].float().mean().item()) for a in torch.unique(self.seed_age)}
        um = self.seed_unmatched
        scan_by_kind = {"matched": float(scan[~um].float().mean().item()) if (~um).any() else None,
                        "unmatched": float(scan[um].float().mean().item()) if um.any() else None}
        return {
            "find_match_direct": roofline("match_prepare + warp_kernel + align_kernel", fm_bytes, stages["find_match_direct"],
                                          trials=n_tried, candidates=float(M), observations=float(self.obs_ptr[-1].item()),
                                          alignment_evaluations_per_trial=n_eval / max(n_tried, 1),
                                          alignment_evaluations_histogram=eval_hist,
                                          alignment_evaluations_per_wave_of_64_trials=wave_max_mean),
            "pose_optimize": roofline("pose_opt_wave_kernel", B * (N * 52.0 + 416.0), stages["pose_optimize"]),
            "update_seeds": roofline("seed_prepare + epi_scan (with the affine warp) + align_kernel (with seed_finish)", seed_bytes,
                                     stages["update_seeds"], seeds=S, scanned_positions_per_seed=n_scan / S,
                                     alignment=seed_evals,
                                     seeds_scanning=float((scan > 0).sum().item()),
                                     scanned_positions_per_seed_by_kind=scan_by_kind, scanned_positions_per_seed_by_age=scan_by_age,
                                     seeds_per_frame_by_scanned_positions=scan_hist),
        }


def long_scan_leg(W: Workload, ev: Events, dev, lib, ages=(60, 120, 240, 480), seeds_per_frame: int = 24, reps: int = 5) -> dict:
    """The long-scan regime of the epipolar search.  The reference scans up to
    max_epi_search_steps = 1000 positions per seed (svo/src/matcher.cpp:248-291); the representative workload tops out at
    ~130.  Here every seed is a NEW one (a = b = 10, mu = 1 / mean scene depth, sigma = z_range / 6: the Seed constructor,
    depth_filter.cpp:37-46) of a keyframe `age` frames back, searched in the current frame of each problem: the baseline
    grows with the age, and with it the segment the interval mu +- sigma projects to -- hundreds of positions.  Reports
    the scanned positions per seed and the time of the whole update per scanned position (the update is the scan here)."""
    from rpg_svo_amd import tracking
    B = W.B
    T = torch.as_tensor(W.T_gt, dtype=torch.float64, device=dev)
    frames = tracking.FrameTable(torch.arange(0, B + 1, dtype=torch.int32, device=dev), T.contiguous())
    df = tracking.DepthFilter(n_pyr_levels=W.n_levels)
    g = torch.Generator().manual_seed(991)
    sel = torch.randperm(W.n_patches, generator=g)[:seeds_per_frame].sort().values.to(dev)
    c = -(T[:B, :9].reshape(B, 3, 3).transpose(1, 2) @ T[:B, 9:, None])[..., 0]
    depth_all = (W.pos_all - c[:, None, :]).norm(dim=-1)
    depth_mean, depth_min = depth_all.mean(1), 0.5 * depth_all.min(1).values
    out = {"seeds_per_frame_and_age": seeds_per_frame, "by_age": {}}
    tot_pos = tot_ms = 0.0
    for age in ages:
        r = torch.arange(0, B - age, device=dev)            # keyframe r, searched in frame r + age
        n = len(r) * seeds_per_frame
        if n == 0:
            continue
        ftr = tracking.FeatureSet(frame=r.repeat_interleave(seeds_per_frame).to(torch.int32).contiguous(),
                                  level=torch.zeros(n, dtype=torch.int32, device=dev),
                                  px=W.px_all[r][:, sel].reshape(n, 2).contiguous(), f=W.f_all[r][:, sel].reshape(n, 3).contiguous())
        cur = (r + age).repeat_interleave(seeds_per_frame).to(torch.int32).contiguous()
        zr = (1.0 / depth_min[r]).repeat_interleave(seeds_per_frame).float().contiguous()
        init = dict(a=torch.full((n,), 10.0, device=dev), b=torch.full((n,), 10.0, device=dev),
                    mu=(1.0 / depth_mean[r]).repeat_interleave(seeds_per_frame).float().contiguous(), z_range=zr,
                    sigma2=(zr * zr / 36.0).contiguous())
        seeds = tracking.SeedSet(**{k: v.clone() for k, v in init.items()}, batch_id=torch.zeros(n, dtype=torch.int32, device=dev))
        res = (torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, 3, dtype=torch.float64, device=dev),
               torch.zeros(n, 2, dtype=torch.float64, device=dev))

        def run():
            for k, v in init.items():
                getattr(seeds, k).copy_(v)
            df.update_seeds(W.store, W.cam, frames, cur, ftr, seeds, 0, out=res)

        ms = ev.time(run, reps, warmup=1)
        steps_ptr = lib.svo_hip_update_seeds_scan_steps(df.last_workspace.data_ptr())
        scan = torch.empty(n, dtype=torch.int32, device=dev)
        capi.check(lib.svo_hip_memcpy_d2d(scan.data_ptr(), steps_ptr, n * 4, torch.cuda.current_stream(dev).cuda_stream), "svo_hip_memcpy_d2d")
        torch.cuda.synchronize()
        st = res[0]
        n_pos = float(scan.sum().item())
        out["by_age"][str(age)] = {
            "seeds": n, "ms": ms, "scanned_positions_per_seed": n_pos / n, "scanned_positions_max": int(scan.max().item()),
            "seeds_scanning_128_or_more_frac": float((scan >= 128).float().mean().item()),
            "skipped_not_visible_frac": float(((st == capi.SEED_NOT_IN_FRAME) | (st == capi.SEED_BEHIND)).float().mean().item()),
            "matched_frac": float(((st == capi.SEED_UPDATED) | (st == capi.SEED_CONVERGED)).float().mean().item()),
            "update_seeds_ns_per_scanned_position": ms * 1e6 / max(n_pos, 1.0),
            # what the scan has to read at least: the union of the 8 x 8 windows along the segment (64 + 8 * 0.7 * n *
            # (|cos| + |sin|) bytes, 4 / pi on average over directions) -- not 64 bytes per position
            "scan_compulsory_GBs": (64.0 * n + 7.13 * n_pos) / (ms * 1e-3) / 1e9}
        tot_pos += n_pos
        tot_ms += ms
    out["scanned_positions_per_seed"] = tot_pos / max(1, sum(v["seeds"] for v in out["by_age"].values()))
    out["ms_per_step"] = tot_ms
    out["epi_scan_ns_per_position"] = tot_ms * 1e6 / max(tot_pos, 1.0)
    return out


def full_track_leg(W: Workload, sia, ev: Events, dev, rank, lib, with_parity: bool, steps: int = 5, mode: str = "representative") -> dict:
    """BASELINE configs[2] on the headline frames: the whole track as one step (FullTrack: representative / easy)."""
    full = FullTrack(W, dev, rank, mode=mode)
    out = sia.alloc_result(W.B, dev)
    marks = []

    def step(timed):
        e0 = ev.mark() if timed else None
        W.run_align(sia, out=out)
        e1 = ev.mark() if timed else None
        full.step(out.T_cur_from_ref, ev if timed else None)
        if timed:
            marks.append((e0, e1, ev.mark()))

    step(False)
    torch.cuda.synchronize()
    # (a library built with -DSCAN_PROFILE -- scripts/scan_phase_profile.sh -- keeps per-region clock totals of the scan kernel)
    scan_prof = getattr(lib, "svo_hip_scan_profile_read", None) if hasattr(lib, "svo_hip_scan_profile_read") else None
    if scan_prof is not None:
        import ctypes
        buf = (ctypes.c_ulonglong * 8)()
        scan_prof(buf)  # clear what the warm-up step left
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    scan_profile = None
    if scan_prof is not None:
        scan_prof(buf)
        names = ("parameters_and_affine_warp", "template_and_setup", "positions_of_a_pass", "box_fetch", "scoring_from_the_box",
                 "fallback_scores_advance", "reduction_and_results")
        tot = float(sum(buf[k] for k in range(7))) or 1.0
        scan_profile = {"wave_clock_share": {n: buf[k] / tot for k, n in enumerate(names)}, "wave_iterations_per_step": buf[7] / steps,
                        "clocks_per_wave_iteration": tot / max(1, buf[7])}
    stages = full.stage_ms(ev)
    stages["sparse_align"] = float(np.mean([ev.ms(a, b) for a, b, _ in marks]))
    step_ms = float(np.mean([ev.ms(a, c) for a, _, c in marks]))
    d = full.describe()
    # The same steps with the depth filter on a second stream, as the reference's mapper thread runs beside its tracker
    # (FullTrack.step_mapper_overlapped): the period of a step in steady state, host wall clock around 2 x `steps` steps.
    overlapped = None
    if mode == "representative":
        try:
            status_serial = full.last["seed_status"].clone()
            s_map = torch.cuda.Stream(dev)
            pose_ready, mapper_done = torch.cuda.Event(), torch.cuda.Event()
            mapper_done.record(torch.cuda.current_stream(dev))

            def ostep():
                W.run_align(sia, out=out)
                return full.step_mapper_overlapped(out.T_cur_from_ref, s_map, pose_ready, mapper_done)

            ostep()
            torch.cuda.synchronize()
            n_o = 2 * steps
            t0 = time.perf_counter()
            for _ in range(n_o):
                st_o = ostep()
            torch.cuda.synchronize()
            o_ms = (time.perf_counter() - t0) / n_o * 1e3
            overlapped = {"ms_per_step": o_ms, "frames_per_s": W.B / o_ms * 1e3, "steps": n_o,
                          "same_seed_statuses_as_the_serial_step": bool(torch.equal(st_o, status_serial)),
                          "what": "depth filter of step i on a second stream beside sparse alignment, matching and pose refinement of "
                                  "step i + 1 (the reference's mapper thread); host wall clock over the steps / their number"}
        except Exception as e:
            overlapped = {"skipped": repr(e)}
    T_ref_est = d.pop("_T_refined")
    res = {"workload": "vga4_n200_full_track" + ("_easy" if mode == "easy" else ""), "frames_per_step": W.B,
           "frames_per_s": W.B / step_ms * 1e3, "ms_per_step": step_ms,
           "ms_per_step_host_wall": wall, "stages_ms": stages,
           "median_pose_error_vs_gt_after_refine": float(np.median(se3.log_norm(T_ref_est, W.T_gt[1:W.B + 1])))}
    res.update(d)
    if overlapped is not None:
        res["mapper_on_its_own_stream"] = overlapped
    if scan_profile is not None:
        res["scan_profile"] = scan_profile
    st = W.align_stats(out)
    rl = full.stage_rooflines(lib, stages)
    rl["sparse_align"] = roofline("sia_kernel", st["alg_bytes"], stages["sparse_align"])
    res["rooflines"] = rl
    if with_parity:
        try:
            res["parity"] = full_track_parity(W, full, out, T_ref_est)
        except Exception as e:
            res["parity"] = {"skipped": repr(e)}
    del full
    torch.cuda.empty_cache()
    return res


def full_track_parity(W: Workload, full: FullTrack, out, T_refined_gpu, n_sample: int = 32) -> dict:
    """The same chain on the host for a sample of frames -- the reference's own translation units
    where oracle/_ref is present (SparseImgAlign::run -> Matcher::findMatchDirect per point ->
    pose_optimizer::optimizeGaussNewton -> DepthFilter::updateSeeds) -- compared stage by stage.  Every stage is run
    twice on the host: fed by the previous stage of the HOST chain (differences accumulate along the chain, as they
    would between two machines), and fed with the DEVICE's output of the previous stage (the stage on its own)."""
    from oracle import pytrack
    which = "ref" if pytrack.ref_available() else "orc"
    trk = pytrack.Track(which)
    B, N, cam = W.B, W.n_patches, W.cam
    lo = max(FullTrack.KF2, FullTrack.SEED_AGES, FullTrack.SEED_AGES_UNMATCHED) + 1
    idx = np.unique(np.linspace(lo, B - 1, n_sample).astype(int))
    m = full.last["match"]
    ok_g = m.ok.view(B, N).cpu().numpy()
    px_g = m.px_cur.view(B, N, 2).cpu().numpy()
    T_k1_g = se3.mul(out.T_cur_from_ref.cpu().numpy(), W.T_ref_w)
    pt_pos = full.pt_pos.view(B, N, 3).cpu().numpy()
    px_all, f_all, pos_all = W.px_all.cpu().numpy(), W.f_all.cpu().numpy(), W.pos_all.cpu().numpy()
    optr = full.obs_ptr.cpu().numpy()
    o_px, o_f, o_fr = full.obs.px.cpu().numpy(), full.obs.f.cpu().numpy(), full.obs.frame.cpu().numpy()
    # seeds of a problem are contiguous (sorted by problem)
    s_b = full.seed_frame_of.cpu().numpy()
    s_lo, s_hi = np.searchsorted(s_b, idx, "left"), np.searchsorted(s_b, idx, "right")
    s_fr, s_px, s_f = full.seed_ftr.frame.cpu().numpy(), full.seed_ftr.px.cpu().numpy(), full.seed_ftr.f.cpu().numpy()
    seed0 = {k: v.cpu().numpy() for k, v in full.seed0.items()}
    st_g = full.last["seed_status"].cpu().numpy()
    mu_g = full.seeds.mu.cpu().numpy()
    opt = pytrack.matcher_options(n_pyr_levels=W.n_levels)
    acc = {k: [] for k in ("d_k1", "d_final", "d_final_same", "ok", "px", "ok_same", "px_same", "st", "st_same", "mu", "mu_same")}
    t0 = time.time()
    for bi, b in enumerate(idx):
        rows = sorted(set([b] + [int(o_fr[k]) for k in range(optr[b * N], optr[(b + 1) * N])] + [int(x) for x in s_fr[s_lo[bi]:s_hi[bi]]]))
        local = {r: i for i, r in enumerate(rows)}
        cur = len(rows)  # the tracked frame b+1 sits last
        pyrs = [trk.create_img_pyramid(W.images[r].cpu().numpy(), W.n_levels) for r in rows + [b + 1]]
        hp = np.ones(N, dtype=np.uint8)
        T_cur, _ = trk.sparse_img_align_run(pyrs[local[b]], pyrs[cur], cam, W.T_ref_w[b], W.T_prior_w[b], px_all[b], f_all[b], hp,
                                            pos_all[b], W.max_level, W.min_level)
        acc["d_k1"].append(se3.log_norm(T_k1_g[b][None], T_cur[None])[0])
        T_rows = [W.T_gt[r] for r in rows]

        def match_all(T_pose):
            frames = pytrack.make_frames(pyrs, np.stack(T_rows + [T_pose]))
            ok_c, px_c, lvl_c = np.zeros(N, dtype=np.int32), np.zeros((N, 2)), np.zeros(N, dtype=np.int32)
            for i in range(N):
                k0, k1 = optr[b * N + i], optr[b * N + i + 1]
                if k1 == k0:
                    continue  # left the image: not tried (Reprojector::reprojectPoint)
                obs = [pytrack.make_feature(local[int(o_fr[k])], o_px[k], o_f[k]) for k in range(k0, k1)]
                _, px_init = trk.reproject_point(cam, T_pose, pt_pos[b, i], 30, (cam.width + 29) // 30)
                ok, px, r = trk.find_match_direct(frames, cam, cur, pt_pos[b, i], obs, px_init, opt)
                ok_c[i], px_c[i], lvl_c[i] = ok, px, r["search_level"]
            return ok_c, px_c, lvl_c

        def refine(T_pose, ok_c, px_c, lvl_c):
            dd = np.stack([(px_c[:, 0] - cam.cx) / cam.fx, (px_c[:, 1] - cam.cy) / cam.fy, np.ones(N)], -1)
            f_new = dd / np.linalg.norm(dd, axis=1, keepdims=True)
            return trk.pose_optimize(cam, T_pose, f_new, lvl_c, (ok_c > 0).astype(np.uint8), pt_pos[b], 2.0, 10)["T_f_w"]

        def seeds_all(T_pose):
            frames = pytrack.make_frames(pyrs, np.stack(T_rows + [T_pose]))
            seeds = []
            for k in range(s_lo[bi], s_hi[bi]):
                sd = pytrack.Seed()
                sd.ftr = pytrack.make_feature(local[int(s_fr[k])], s_px[k], s_f[k])
                sd.batch_id, sd.a, sd.b, sd.mu = 0, float(seed0["a"][k]), float(seed0["b"][k]), float(seed0["mu"][k])
                sd.z_range, sd.sigma2 = float(seed0["z_range"][k]), float(seed0["sigma2"][k])
                seeds.append(sd)
            _, so, io = trk.update_seeds(frames, cam, cur, seeds, batch_counter=0, opt=opt)
            return np.array([x.status for x in io]), np.array([x.mu for x in so])

        def cmp_match(ok_c, px_c, key_ok, key_px):
            acc[key_ok].append(np.mean((ok_c > 0) == (ok_g[b] > 0)))
            both = (ok_c > 0) & (ok_g[b] > 0)
            acc[key_px].append(np.mean(np.abs(px_c[both] - px_g[b][both]).max(1) < 1e-6) if both.any() else 1.0)

        def cmp_seeds(stc, muc, key_st, key_mu):
            sg, mg = st_g[s_lo[bi]:s_hi[bi]].copy(), mu_g[s_lo[bi]:s_hi[bi]]
            # a seed behind the camera or outside the image is left untouched by the reference (depth_filter.cpp:225-232;
            # the host checker reports 0 for both): the device's two codes for it compare as that
            sg[np.isin(sg, (capi.SEED_BEHIND, capi.SEED_NOT_IN_FRAME))] = 0
            acc[key_st].append(np.mean(stc == sg))
            upd = (stc == sg) & np.isin(stc, (pytrack.SEED_UPDATED, pytrack.SEED_CONVERGED))
            if upd.any():
                acc[key_mu].append(np.max(np.abs(muc[upd] - mg[upd]) / np.abs(muc[upd])))

        # (1) the host chain on its own outputs
        ok_c, px_c, lvl_c = match_all(T_cur)
        cmp_match(ok_c, px_c, "ok", "px")
        T_po = refine(T_cur, ok_c, px_c, lvl_c)
        acc["d_final"].append(se3.log_norm(T_refined_gpu[b][None], T_po[None])[0])
        cmp_seeds(*seeds_all(T_po), "st", "mu")
        # (2) each host stage fed with the device's previous stage
        ok_s, px_s, lvl_s = match_all(T_k1_g[b])
        cmp_match(ok_s, px_s, "ok_same", "px_same")
        lvl_g = m.search_level.view(B, N)[b].cpu().numpy()
        T_po_s = refine(T_k1_g[b], ok_g[b], px_g[b], lvl_g)
        acc["d_final_same"].append(se3.log_norm(T_refined_gpu[b][None], T_po_s[None])[0])
        cmp_seeds(*seeds_all(T_refined_gpu[b]), "st_same", "mu_same")
    mx = lambda k: float(np.max(acc[k])) if acc[k] else None
    return {"frames_compared": int(len(idx)), "against": "reference" if which == "ref" else "port",
            "sparse_align_se3_lognorm_max": mx("d_k1"), "sparse_align_se3_lognorm_median": float(np.median(acc["d_k1"])),
            "host_chain_on_its_own_outputs": {
                "find_match_direct_same_verdict_frac": float(np.mean(acc["ok"])),
                "find_match_direct_same_pixel_frac_of_common_matches": float(np.mean(acc["px"])),
                "refined_pose_se3_lognorm_max": mx("d_final"), "refined_pose_se3_lognorm_median": float(np.median(acc["d_final"])),
                "seed_status_same_frac": float(np.mean(acc["st"])), "seed_mu_max_rel_diff": mx("mu")},
            "host_stage_fed_with_the_device_previous_stage": {
                "find_match_direct_same_verdict_frac": float(np.mean(acc["ok_same"])),
                "find_match_direct_same_pixel_frac_of_common_matches": float(np.mean(acc["px_same"])),
                "refined_pose_se3_lognorm_max": mx("d_final_same"),
                "seed_status_same_frac": float(np.mean(acc["st_same"])), "seed_mu_max_rel_diff": mx("mu_same")},
            "seeds_compared_per_frame": float(np.mean(s_hi - s_lo)), "seconds": time.time() - t0}


if __name__ == "__main__":
    main()
