#!/usr/bin/env python
"""bench.py -- frames/s of sparse image alignment (+ the Gauss-Newton pose
refinement it contains) on synthetic VGA pyramid batches.

A "step" is one pass of the hot path (svo_hip_sparse_align, K1) over one batch
of B independent (reference frame, current frame) problems per GPU, with the
image pyramids and feature arrays already resident in HBM.  Workload at N=1 is
BASELINE.json configs[1]: 640x480 mono, 4 pyramid levels (3 -> 0), 200 reference
patches, SparseImgAlign only.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the keys below).  The oracle (oracle/, CPU
restatement) is used here only for the `cpu_baseline` leg and a parity read-out;
it is never the thing measured as `value`.
"""
from __future__ import annotations

import argparse
import json
import os
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

WORKLOADS = {
    # name: (width, height, f, n_levels, max_level, min_level, n_patches, margin, cell)
    "vga4_n200_sparse_align": (640, 480, 400.0, 4, 3, 0, 200, 28, 32),
    "svo_default_752_l4to2_n120": (752, 480, 315.5, 5, 4, 2, 120, 56, 40),
    "xga5_n1000_sparse_align": (1280, 960, 800.0, 5, 4, 0, 1000, 56, 32),
}


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


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def spawn_command(n_gpus: int, argv: list[str], port: int) -> list[str]:
    """The launch line of the contract (one rank per GPU of one node, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def spawn_ranks(n_gpus: int) -> int:
    """`python bench.py --gpus N` from a plain shell: re-launch under torch.distributed.run."""
    import subprocess
    cmd = spawn_command(n_gpus, sys.argv[1:], free_port())
    if os.environ.get("SVO_BENCH_DRY_SPAWN") == "1":
        print(json.dumps({"spawn": cmd}))
        return 0
    return subprocess.call(cmd)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16384, help="frames per step per GPU")
    ap.add_argument("--workload", default="vga4_n200_sparse_align", choices=sorted(WORKLOADS))
    ap.add_argument("--noise", type=float, default=0.0, help="image noise sigma (gray levels)")
    ap.add_argument("--cpu-sample", type=int, default=8192, help="frames timed on the host for cpu_baseline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--n-iter", type=int, default=30, help="Gauss-Newton iteration cap per level (30 in the pipeline)")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step (all kernel launches of the pipeline) in a HIP graph and replay it: "
                         "small batches -- e.g. one frame per camera of a rig -- are launch-bound")
    ap.add_argument("--pipeline", default="align", choices=["align", "full"],
                    help="align: SparseImgAlign only (BASELINE configs[1], the default); full: configs[2] -- "
                         "sparse align + reprojection matching (align2D) + pose refinement + depth-filter update")
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
    lib = capi.load()

    width, height, focal, n_levels, max_level, min_level, n_patches, margin, cell = WORKLOADS[args.workload]
    cam = synth.Camera(width, height, focal, focal, width / 2.0, height / 2.0)
    B = args.batch

    # ---- synthetic replay sequence: B+1 frames, problem b = (frame b -> frame b+1) ----
    t_gen = time.time()
    tex = synth.make_texture(seed=12345)
    T_gt = synth.make_trajectory(B + 1, seed=12345 + rank)
    images = synth.render(tex, T_gt, cam, device=dev, chunk=32)
    if args.noise > 0:
        g = torch.Generator(device=dev).manual_seed(99 + rank)
        images = (images.float() + args.noise * torch.randn(images.shape, generator=g, device=dev)).round().clamp(0, 255).to(torch.uint8)
    px_all = synth.select_features(images[:B], n_patches, margin=margin, cell=cell)
    g = torch.Generator().manual_seed(777 + rank)
    px_all = px_all + (torch.rand(px_all.shape, generator=g, dtype=torch.float64) - 0.5).to(dev)
    f_all, pos_all = synth.features_3d(T_gt[:B], cam, px_all)
    store = PyramidStore(width, height, n_levels, B + 1, device=dev)
    store.load_images(images)  # level 0 copy + K0 pyramid build (untimed here)
    T_ref_w = T_gt[:B]
    T_prior_w = T_ref_w.copy()  # constant-position prior, frame_handler_mono.cpp:132
    T_cr, xyz_ref = marshal_problem(T_ref_w, T_prior_w, f_all.cpu().numpy(), pos_all.cpu().numpy())
    tdev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    ref_slot = torch.arange(0, B, dtype=torch.int32, device=dev)
    cur_slot = torch.arange(1, B + 1, dtype=torch.int32, device=dev)
    n_t = torch.full((B,), n_patches, dtype=torch.int32, device=dev)
    px_t = px_all.contiguous()
    xyz_t = tdev(xyz_ref, torch.float64)
    T_in = tdev(T_cr, torch.float64)
    sia = SparseImgAlign(max_level, min_level, args.n_iter)
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
    full = FullTrack(args, cam, store, T_gt, px_all, f_all, pos_all, n_patches, n_levels, dev, rank) if args.pipeline == "full" else None
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    stream = torch.cuda.current_stream(dev).cuda_stream
    ev = []
    for _ in range(2 * args.steps):
        e = C_void()
        capi.check(lib.svo_hip_event_create(e.ref()))
        ev.append(e.value)

    graph = None

    counter = [0]

    def step_compute(i: int | None) -> None:
        st = torch.cuda.current_stream(dev).cuda_stream
        k = counter[0] % len(outs)
        o = outs[k]
        if gather is not None:
            gather.local(counter[0] if len(outs) > 1 else 0)  # waits for the gather that last read this buffer
        if i is not None:
            lib.svo_hip_event_record(ev[2 * i], st)
        sia.run(store, cam, ref_slot, cur_slot, n_t, px_t, xyz_t, T_in, out=o)
        if i is not None:
            lib.svo_hip_event_record(ev[2 * i + 1], st)
        if full is not None:
            full.step(o.T_cur_from_ref, lib, st, timed=i is not None)

    def step(i: int | None) -> None:
        if graph is not None:
            if i is not None:
                lib.svo_hip_event_record(ev[2 * i], stream)
            graph.replay()
            if i is not None:
                lib.svo_hip_event_record(ev[2 * i + 1], stream)
        else:
            step_compute(i)
        if gather is not None:  # RCCL gather of the SE(3) results (the only exchange step)
            if len(outs) > 1:
                gather.submit(counter[0])
            else:
                gather.submit(0)
                gather.result(0)
        counter[0] += 1

    for _ in range(args.warmup):
        step(None)
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
            step_compute(None)
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.cuda.graph(graph):
            step_compute(None)
        torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
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
        # the exchange step on its own (untimed region): blocking all-gathers of one pose block
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
        gather_stats = {"collective": "all_gather_into_tensor (RCCL)", "bytes_per_rank_per_step": int(B * 12 * 8),
                        "bytes_gathered_per_step": int(world * B * 12 * 8), "ms_blocking_avg": float(tg.item()) * 1e3,
                        "overlapped_in_timed_region": len(outs) > 1}

    # per-launch kernel duration from HIP events on the launch stream
    kms = []
    for i in range(args.steps):
        ms = C_float()
        capi.check(lib.svo_hip_event_elapsed_ms(ev[2 * i], ev[2 * i + 1], ms.ref()))
        kms.append(ms.value)
    kernel_ms = float(np.mean(kms)) if kms else float("nan")

    if rank != 0:
        dist.destroy_process_group()
        return

    out = outs[(counter[0] - 1) % len(outs)]  # the result block of the last step
    n_tracked = out.n_tracked.cpu().numpy().astype(np.float64)
    iters = out.iters.cpu().numpy().astype(np.float64)
    alg_bytes = algorithmic_bytes(np.full(B, n_patches, dtype=np.float64), n_tracked, iters, max_level, min_level)
    achieved_gbs = alg_bytes / (kernel_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"{args.workload}:B{B}"
            if key in tj:
                traffic = tj[key]
        except Exception:
            traffic = None

    T_est_w = se3.mul(out.T_cur_from_ref.cpu().numpy(), T_ref_w)
    gt_err = se3.log_norm(T_est_w, T_gt[1:B + 1])

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
        "dtype": "f32 pixels / f64 pose+normal equations",
        "data": "synthetic",
        "config": {
            "workload": args.workload if full is None else args.workload.replace("sparse_align", "full_track"), "image": f"{width}x{height}", "pyr_levels": n_levels,
            "schedule": f"levels {max_level}->{min_level}", "patches_per_frame": n_patches,
            "frames_per_step_per_gpu": B, "n_iter_cap": args.n_iter, "image_noise_sigma": args.noise,
            "parallelism": f"frames sharded 1 rank/GPU x{world}" + (", RCCL all_gather of poses, double-buffered and overlapped with the next step" if world > 1 else ""),
            "hip_graph": bool(args.graph),
            "mean_gn_iterations_per_frame": float(iters.sum(1).mean()),
            "mean_tracked_patches": float(n_tracked.mean()),
            "median_pose_error_vs_gt": float(np.median(gt_err)),
        },
        "roofline": {
            "bound": "hbm", "kernel": "sia_kernel (svo_hip_sparse_align)",
            "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
            "kernel_ms_avg": kernel_ms, "algorithmic_bytes_per_launch": alg_bytes,
            "algorithmic_bytes_per_frame": alg_bytes / B,
            # iterations/s and per-iteration time of the batch
            "gn_iterations_per_s": float(iters.sum()) / (kernel_ms * 1e-3),
            "us_per_gn_iteration_of_the_batch": kernel_ms * 1e3 / max(float(iters.sum(1).mean()), 1e-9),
        },
        "setup_s": t_gen,
    }
    if full is not None:
        d = full.describe()
        T_ref_est = d.pop("_T_refined")
        d["median_pose_error_vs_gt_after_refine"] = float(np.median(se3.log_norm(T_ref_est, T_gt[1:B + 1])))
        result["config"].update(d)
        result["stages_ms"] = full.stage_ms(lib) if not args.graph else {}
        result["stages_ms"]["sparse_align" if not args.graph else "whole_graph"] = kernel_ms

    if not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline(args, cam, images, T_gt, px_all, f_all, pos_all, T_ref_w, T_prior_w,
                                              n_levels, max_level, min_level, n_patches, T_est_w, result,
                                              out.iters.cpu().numpy())
    if world == 1:
        result["k0_pyramid"] = pyramid_roofline(lib, store, images, stream)
    if not args.no_cpu_baseline and world == 1:
        try:
            result["dropin_sequence"] = dropin_sequence()
        except Exception as e:  # the demonstration libraries are optional (built from the reference checkout)
            result["dropin_sequence"] = {"skipped": str(e)}
    if gather_stats is not None:
        result["gather"] = gather_stats
    print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


def pyramid_roofline(lib, store, images, stream, reps: int = 5) -> dict:
    """K0, the one HBM-streaming kernel of the path: image pyramids of the whole
    replay batch rebuilt from the packed images in a single fused pass.  Algorithmic bytes per
    frame = w*h read + every level written once."""
    n = images.shape[0]
    per_frame = images.shape[1] * images.shape[2] + store.bytes_per_pyramid()

    def timed(fn):
        ms = []
        for _ in range(reps + 1):
            e0, e1 = C_void(), C_void()
            capi.check(lib.svo_hip_event_create(e0.ref()))
            capi.check(lib.svo_hip_event_create(e1.ref()))
            lib.svo_hip_event_record(e0.value, stream)
            fn()
            lib.svo_hip_event_record(e1.value, stream)
            m = C_float()
            capi.check(lib.svo_hip_event_elapsed_ms(e0.value, e1.value, m.ref()))
            ms.append(m.value)
            lib.svo_hip_event_destroy(e0.value)
            lib.svo_hip_event_destroy(e1.value)
        return float(np.mean(ms[1:]))

    tiles = {}
    for tw in (128, 256, 257, 512):
        lib.svo_hip_pyramid_set_tile(tw)
        tiles[str(tw)] = timed(lambda: store.load_images(images, 0))
    lib.svo_hip_pyramid_set_tile(0)
    fused = timed(lambda: store.load_images(images, 0))

    def per_level():
        store.load_images(images, 0, build=False)
        store.build_per_level(0, n)
    unfused = timed(per_level)
    gbs = n * per_frame / (fused * 1e-3) / 1e9
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        traffic = tj.get(f"k0_pyramid:vga4:B{n}") if (images.shape[1], images.shape[2]) == (480, 640) else None
    except Exception:
        pass
    return {"kernel": "pyramid_fused_kernel (svo_hip_pyramid_build_from_images)", "frames": int(n),
            "ms": fused, "frames_per_s": n / (fused * 1e-3), "algorithmic_bytes_per_frame": int(per_frame),
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": traffic, "algorithmic_bytes_per_launch": int(n * per_frame),
            "ms_level0_copy_plus_one_launch_per_level": unfused, "ms_by_tile_width": tiles}


def dropin_sequence(n_frames: int = 120) -> dict:
    """Single-stream, image-in -> pose-out: the reference's own svo::FrameHandlerMono on a
    752x480 synthetic sequence, once with all-reference translation units on the host CPU and
    once with the drop-in HIP bodies (tests/dropin, rpg_svo_amd/host/dropin).  Reports the
    trajectory agreement (the metric's "ATE vs CPU ref") and the per-frame latency of both."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
    import pypipeline as pp
    if not (pp.available("ref") and pp.available("hip")):
        raise RuntimeError("tests/dropin/_build/*.so not built")
    cam = synth.Camera(752, 480, 315.5, 315.5, 376.0, 240.0)
    T = synth.make_trajectory(n_frames, seed=5, max_step=0.02, max_rot_deg=0.3)
    imgs = synth.render(synth.make_texture(seed=12345), T, cam).numpy()
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(2)
    os.dup2(devnull, 2)  # the reference logs every frame to stderr
    try:
        ref = pp.run_sequence("ref", cam, imgs, T)
        pp.run_sequence("hip", cam, imgs[:10], T[:10])  # warm-up: context creation, first launches
        hip = pp.run_sequence("hip", cam, imgs, T)
    finally:
        os.dup2(saved, 2)
        os.close(devnull)
    Tr = np.stack([r["T_f_w"] for r in ref])
    Th = np.stack([r["T_f_w"] for r in hip])
    d = se3.log_norm(Th, Tr)
    med = lambda rs, k: float(np.median([r[k] for r in rs[1:]]) * 1e3)
    stages = ("tot_time", "sparse_img_align", "reproject", "pose_optimizer")
    return {"frames": n_frames, "image": "752x480", "se3_lognorm_max": float(d.max()), "se3_lognorm_median": float(np.median(d)),
            "ate_rmse_vs_cpu_m": horn_ate(se3.inv(Th)[:, 9:], se3.inv(Tr)[:, 9:]),
            "keyframes": int(sum(r["is_keyframe"] for r in hip)),
            "same_keyframe_frames": [r["is_keyframe"] for r in ref] == [r["is_keyframe"] for r in hip],
            "median_ms_per_frame_cpu_reference": {k: med(ref, "t_" + k) for k in stages},
            "median_ms_per_frame_hip_dropin": {k: med(hip, "t_" + k) for k in stages}}


def cpu_baseline(args, cam, images, T_gt, px_all, f_all, pos_all, T_ref_w, T_prior_w, n_levels, max_level,
                 min_level, n_patches, T_est_w, result, iters_gpu) -> dict:
    """Times the oracle (CPU restatement of the reference path; the reference itself
    cannot be built: Eigen/OpenCV/Sophus/vikit are absent) on a bounded sample of
    the same problems, on this box's host cores."""
    from oracle import pyoracle
    S = min(args.cpu_sample, px_all.shape[0])
    imgs = images[:S + 1].cpu().numpy()
    pyrs = [pyoracle.create_img_pyramid(im, n_levels, pyoracle.HALFSAMPLE_AUTO) for im in imgs]
    rs = np.arange(S, dtype=np.int32)
    cs = rs + 1
    nn = np.full(S, n_patches, dtype=np.int32)
    px = px_all[:S].cpu().numpy()
    f = f_all[:S].cpu().numpy()
    pos = pos_all[:S].cpu().numpy()
    hp = np.ones((S, n_patches), dtype=np.uint8)
    cores = os.cpu_count() or 1
    s1 = min(S, 2048)
    # kind "reference": the reference's own SparseImgAlign translation unit (oracle/_ref, built
    # from /root/reference/svo/src in the build container, travels prebuilt); otherwise the C port
    which = "ref" if pyoracle.ref_available() else "orc"

    def timed(k, threads):
        tm = {}
        t0 = time.perf_counter()
        T, r = pyoracle.sparse_img_align_batch(pyrs, rs[:k], cs[:k], cam, T_ref_w[:k], T_prior_w[:k], nn[:k], px[:k],
                                               f[:k], hp[:k], pos[:k], max_level, min_level, args.n_iter,
                                               n_threads=threads, which=which, timing=tm)
        return T, r, tm.get("run_seconds", time.perf_counter() - t0)

    _, _, t1 = timed(s1, 1)
    T_cpu, res, tn = timed(S, cores)
    best_threads, best_rate = cores, S / tn
    sweep = {str(cores): S / tn}
    for th in (cores // 2, cores // 4):  # SMT siblings / allocator contention: fewer threads can be faster
        if th >= 1:
            k = max(1, S // 2)
            _, _, tt = timed(k, th)
            sweep[str(th)] = k / tt
            if k / tt > best_rate:
                best_threads, best_rate = th, k / tt
    d = se3.log_norm(T_est_w[:S], T_cpu)
    pos_gpu = se3.inv(T_est_w[:S])[:, 9:]
    pos_cpu = se3.inv(T_cpu)[:, 9:]
    result["parity"] = {
        "frames_compared": int(S), "se3_lognorm_max": float(d.max()), "se3_lognorm_median": float(np.median(d)),
        "ate_rmse_vs_cpu_m": horn_ate(pos_gpu, pos_cpu),
        "same_iteration_counts_frac": float(np.mean([np.array_equal(r["iters"], it) for r, it in zip(res, iters_gpu[:S])])),
    }
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    impl = ("the reference's own sparse_img_align.cpp (oracle/_ref/libsvo_ref.so, g++ -O3 against dependency shims), "
            "run() calls only") if which == "ref" else "oracle/libsvo_oracle.so (C port, gcc -O3)"
    return {"value": best_rate, "unit": "frames/s", "cores": best_threads, "kind": "reference" if which == "ref" else "port",
            "sample": f"{S} of the benchmark's own frame pairs, {impl}; best of the thread counts tried",
            "frames_per_s_by_threads": sweep, "host_logical_cpus": cores,
            "value_1core": s1 / t1, "sample_1core": f"{s1} frame pairs, 1 thread", "cpu_model": model}


class FullTrack:
    """BASELINE configs[2]: what FrameHandlerMono::processFrame does after sparse alignment
    (frame_handler_mono.cpp:145-190) plus the depth-filter update of the mapping thread, for
    the same B replay frames: reproject the reference frame's map points, findMatchDirect
    (affine warp + align2D), pose_optimizer::optimizeGaussNewton, DepthFilter::updateSeeds."""

    STAGES = ("compose_pose", "reproject", "find_match_direct", "cam2world", "pose_optimize", "update_seeds")

    def __init__(self, args, cam, store, T_gt, px_all, f_all, pos_all, n_patches, n_levels, dev, rank):
        from rpg_svo_amd import tracking
        self.tr = tracking
        self.cam, self.store, self.dev = cam, store, dev
        B, N = px_all.shape[0], n_patches
        self.B, self.N = B, N
        g = torch.Generator().manual_seed(4242 + rank)
        T = torch.as_tensor(T_gt, dtype=torch.float64, device=dev)
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
        # map points: the reference frame's features, 1% depth error along the viewing ray
        c_ref = -(T[:B, :9].reshape(B, 3, 3).transpose(1, 2) @ T[:B, 9:, None])[..., 0]
        ray = pos_all - c_ref[:, None, :]
        noise = 1.0 + 0.01 * torch.randn(B, N, 1, generator=g, dtype=torch.float64).to(dev)
        self.pt_pos = (c_ref[:, None, :] + ray * noise).reshape(M, 3).contiguous()
        # observations: the feature in frame b, and (where it projects inside) in frame b-2
        b_idx = torch.arange(B, device=dev)
        older = (b_idx - 2).clamp(min=0)
        R2 = T[older, :9].reshape(B, 3, 3)
        p2 = (R2[:, None] @ pos_all[..., None])[..., 0] + T[older, None, 9:]
        px2 = torch.stack([cam.fx * p2[..., 0] / p2[..., 2] + cam.cx, cam.fy * p2[..., 1] / p2[..., 2] + cam.cy], -1)
        has2 = ((b_idx >= 2)[:, None] & (px2[..., 0] > 12) & (px2[..., 0] < cam.width - 12) & (px2[..., 1] > 12)
                & (px2[..., 1] < cam.height - 12) & (p2[..., 2] > 0))
        n_obs = 1 + has2.reshape(M).to(torch.int32)
        ptr = torch.zeros(M + 1, dtype=torch.int32, device=dev)
        ptr[1:] = torch.cumsum(n_obs, 0)
        n_total = int(ptr[-1].item())
        first = ptr[:-1].long()
        o_frame = torch.zeros(n_total, dtype=torch.int32, device=dev)
        o_px = torch.zeros(n_total, 2, dtype=torch.float64, device=dev)
        o_f = torch.zeros(n_total, 3, dtype=torch.float64, device=dev)
        o_frame[first] = b_idx.repeat_interleave(N).to(torch.int32)
        o_px[first] = px_all.reshape(M, 2)
        o_f[first] = f_all.reshape(M, 3)
        sec = first[has2.reshape(M)] + 1
        o_frame[sec] = older.repeat_interleave(N)[has2.reshape(M)].to(torch.int32)
        o_px[sec] = px2.reshape(M, 2)[has2.reshape(M)]
        d2 = torch.stack([(o_px[sec][:, 0] - cam.cx) / cam.fx, (o_px[sec][:, 1] - cam.cy) / cam.fy,
                          torch.ones(len(sec), dtype=torch.float64, device=dev)], -1)
        o_f[sec] = d2 / d2.norm(dim=-1, keepdim=True)
        self.obs_ptr = ptr
        self.obs = tracking.FeatureSet(frame=o_frame, level=torch.zeros(n_total, dtype=torch.int32, device=dev),
                                       px=o_px, f=o_f)
        self.matcher = tracking.Matcher(align_max_iter=10, n_pyr_levels=n_levels)
        self.n = torch.full((B,), N, dtype=torch.int32, device=dev)
        # seeds: one per reference feature, inverse depth known to 10 %, range from 0.6 x depth
        depth = ray.norm(dim=-1).reshape(M)
        dm = depth * (1.0 + 0.1 * torch.randn(M, generator=g, dtype=torch.float64).to(dev))
        z_range = (1.0 / (0.6 * depth)).float()
        self.seed0 = dict(a=torch.full((M,), 10.0, device=dev), b=torch.full((M,), 10.0, device=dev),
                          mu=(1.0 / dm).float(), z_range=z_range, sigma2=z_range * z_range / 36.0)
        self.seeds = tracking.SeedSet(**{k: v.clone() for k, v in self.seed0.items()},
                                      batch_id=torch.zeros(M, dtype=torch.int32, device=dev))
        self.seed_ftr = tracking.FeatureSet(frame=b_idx.repeat_interleave(N).to(torch.int32).contiguous(),
                                            level=torch.zeros(M, dtype=torch.int32, device=dev),
                                            px=px_all.reshape(M, 2).contiguous(), f=f_all.reshape(M, 3).contiguous())
        self.df = tracking.DepthFilter(n_pyr_levels=n_levels)
        self.f_new = torch.empty(M, 3, dtype=torch.float64, device=dev)
        self.events = []
        self.last = {}

    def _mark(self, lib, stream, timed):
        if timed:
            e = C_void()
            capi.check(lib.svo_hip_event_create(e.ref()))
            lib.svo_hip_event_record(e.value, stream)
            self.events.append(e.value)

    def step(self, T_cur_from_ref, lib, stream, timed):
        tr = self.tr
        self._mark(lib, stream, timed)
        tr.compose_poses(T_cur_from_ref, self.T_ref, out=self.frame_T, out_index=self.cur_rows)
        self._mark(lib, stream, timed)
        cell, px = tr.reproject_points(self.cam, self.frames, self.cur_frame, self.pt_pos, 30, (self.cam.width + 29) // 30)
        self._mark(lib, stream, timed)
        m = self.matcher.find_match_direct(self.store, self.cam, self.frames, self.cur_frame, self.pt_pos, self.obs_ptr,
                                           self.obs, px)
        self._mark(lib, stream, timed)
        tr.cam2world(self.cam, m.px_cur, out=self.f_new)
        self._mark(lib, stream, timed)
        B, N = self.B, self.N
        po = tr.optimize_gauss_newton(self.cam, self.n, self.f_new.view(B, N, 3), m.search_level.view(B, N),
                                      self.pt_pos.view(B, N, 3), (m.ok > 0).to(torch.uint8).view(B, N),
                                      self.frame_T[B + 1:], 2.0, 10)
        self._mark(lib, stream, timed)
        for k, v in self.seed0.items():
            getattr(self.seeds, k).copy_(v)
        status, _, _ = self.df.update_seeds(self.store, self.cam, self.frames, self.cur_frame, self.seed_ftr, self.seeds, 0)
        self._mark(lib, stream, timed)
        self.last = dict(match=m, pose=po, seed_status=status)

    def stage_ms(self, lib):
        k = len(self.STAGES) + 1
        acc = {s: [] for s in self.STAGES}
        for i in range(0, len(self.events) - k + 1, k):
            for j, sname in enumerate(self.STAGES):
                ms = C_float()
                capi.check(lib.svo_hip_event_elapsed_ms(self.events[i + j], self.events[i + j + 1], ms.ref()))
                acc[sname].append(ms.value)
        return {s: float(np.mean(v)) for s, v in acc.items() if v}

    def describe(self):
        m, po, st = self.last["match"], self.last["pose"], self.last["seed_status"].cpu().numpy()
        T_est = po.T_f_w.cpu().numpy()
        return {"match_trials_per_frame": self.N, "matches_per_frame": float(m.ok.float().sum().item() / self.B),
                "pose_refine_obs_after_pruning": float(po.stats[:, 3].mean().item()),
                "seeds_per_frame": self.N,
                "seed_status_hist": {str(k): int((st == k).sum()) for k in np.unique(st)},
                "pipeline": "sparse_align -> reproject -> findMatchDirect -> pose_optimize -> updateSeeds",
                "_T_refined": T_est}


class C_void:
    def __init__(self):
        import ctypes
        self._v = ctypes.c_void_p()

    def ref(self):
        import ctypes
        return ctypes.byref(self._v)

    @property
    def value(self):
        return self._v.value


class C_float:
    def __init__(self):
        import ctypes
        self._v = ctypes.c_float()

    def ref(self):
        import ctypes
        return ctypes.byref(self._v)

    @property
    def value(self):
        return self._v.value


if __name__ == "__main__":
    main()
