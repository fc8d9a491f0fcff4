#!/usr/bin/env python3
"""The drop-in sequence of bench.py's dropin leg alone (hip flavour, synchronous mapper), for a
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace run: where a frame's host time goes, API call by API call.

    rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace -- \
        python scripts/dropin_trace.py [defer]
    python scripts/dropin_trace.py --report gpurun_out/trace     # afterwards: the timeline of a median frame
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))


def _frames():
    for a in sys.argv[1:]:
        if a.startswith("frames="):
            return int(a.split("=")[1])
    return 120


def run(defer):
    import numpy as np
    import pypipeline as pp
    from rpg_svo_amd import synth
    cam = synth.Camera(752, 480, 315.5, 315.5, 376.0, 240.0)
    T = synth.make_trajectory(_frames(), seed=5, max_step=0.02, max_rot_deg=0.3)
    import torch
    imgs = synth.render(synth.make_texture(seed=12345), T, cam, device="cuda" if torch.cuda.is_available() else "cpu").cpu().numpy()
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)
    pp.run_sequence("hip", cam, imgs[:20], T[:20], defer_mapper=defer)
    st = {}
    res = pp.run_sequence("hip", cam, imgs, T, stats_out=st, defer_mapper=defer)
    tail = res[-min(len(res) - 1, 400):]  # with frames=600: the frames that see the full map (10 keyframes)
    print("tot_time median %.1f us (last %d frames: %.1f us, %g keyframes, %g candidates), frame period %.1f us" % (
        np.median([r["t_tot_time"] for r in res[1:]]) * 1e6, len(tail), np.median([r["t_tot_time"] for r in tail]) * 1e6,
        np.median([r["n_kfs"] for r in tail]), np.median([r["n_candidates"] for r in tail]), st["wall_ms_per_frame"] * 1e3))
    for k, v in st["stages"].items():
        print(k, {a: round(b, 1) for a, b in v.items()})


def _rows(d, pattern):
    out = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


def report(d, frame=-1):
    api = _rows(d, "*hip_api_trace.csv")
    ker = _rows(d, "*kernel_trace.csv")
    mem = _rows(d, "*memory_copy_trace.csv")
    ev = []
    for r in api:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api", r["Function"]))
    for r in ker:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "gpu", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-48:]))
    for r in mem:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "dma", r.get("Direction", "copy")))
    ev.sort()
    # frames are delimited by the sparse-alignment kernel (one launch per frame; the last 120-frame run counts)
    sia = [e for e in ev if e[2] == "gpu" and "sia_kernel" in e[3]]
    sia = sia[-(_frames() - 1):]
    if frame < 0:  # the frame of median length (tracing itself produces outliers)
        periods = sorted((sia[i][0] - sia[i - 1][0], i) for i in range(20, len(sia)))
        frame = periods[len(periods) // 2][1]
    t0, t1 = sia[frame - 1][0], sia[frame][0]
    print("frame %d: %.1f us between two sparse-alignment kernel starts (under the trace; every API call costs 2-3 us more "
          "than untraced)" % (frame, (t1 - t0) / 1e3))
    for s, e, kind, name in ev:
        if t0 - 60000 <= s < t1 - 60000:
            print("%9.1f  %-4s %7.1f us  %s" % ((s - t0) / 1e3, kind, (e - s) / 1e3, name))
    # host time inside HIP API calls per frame, by function
    tot = {}
    for s, e, kind, name in ev:
        if kind == "api" and sia[0][0] <= s < sia[-1][0]:
            c = tot.setdefault(name, [0, 0.0])
            c[0] += 1
            c[1] += (e - s) / 1e3
    n = len(sia) - 1
    print("\nHIP API time per frame (us), %d frames:" % n)
    for name, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("  %-36s %6.2f calls %8.2f us" % (name, c / n, t / n))
    print("  total %.1f us" % (sum(t for _, t in tot.values()) / n))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        rest = [a for a in sys.argv[3:] if not a.startswith("frames=")]
        report(sys.argv[2], int(rest[0]) if rest else -1)
    else:
        run(1 if "defer" in sys.argv[1:] else 0)
