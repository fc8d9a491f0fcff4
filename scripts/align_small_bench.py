#!/usr/bin/env python3
"""K3 on a camera frame's worth of trials: svo_hip_align_batch on M trials (default 330, the depth filter's seeds of the
single-stream drop-in; 130 = the reprojector's), timed with events over many launches, for each library named on the
command line ("main" = rpg_svo_amd/lib/libsvo_hip.so, else build/variants/lib<name>.so), one child process per library.
Every library's refined pixels are compared bit for bit with the first one's.

    python scripts/align_small_bench.py [M=330] [reps=300] main [variant ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _arg(name, default):
    for a in sys.argv[1:]:
        if a.startswith(name + "="):
            return a.split("=", 1)[1]
    return default


def child(M, reps):
    import numpy as np
    import torch
    from rpg_svo_amd import synth, tracking
    from rpg_svo_amd.pyramid import PyramidStore
    seq = synth.make_sequence(4, 200)
    imgs = seq.images
    n, h, w = imgs.shape
    store = PyramidStore(w, h, 5, n, device="cuda:0")
    store.load_images(imgs.to("cuda:0"))
    rng = np.random.default_rng(11)
    slot = rng.integers(0, n, size=M).astype(np.int32)
    level = rng.integers(0, 3, size=M).astype(np.int32)
    pwb, px0 = np.zeros((M, 100), np.uint8), np.zeros((M, 2))
    dirs = rng.normal(size=(M, 2)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    use_1d = (rng.uniform(size=M) < 0.2).astype(np.uint8)
    pyr = [[store.level(s, l) for l in range(3)] for s in range(n)]
    for t in range(M):
        img = pyr[slot[t]][level[t]]
        hh, ww = img.shape
        u, v = rng.integers(12, ww - 12), rng.integers(12, hh - 12)
        src = pyr[(slot[t] + (t % 2)) % n][level[t]]
        pwb[t] = src[v - 5:v + 5, u - 5:u + 5].ravel()
        px0[t] = [u + rng.uniform(-2.0, 2.0), v + rng.uniform(-2.0, 2.0)]
    d = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
    slot_d, level_d, pwb_d, dirs_d, u1_d = d(slot, torch.int32), d(level, torch.int32), d(pwb, torch.uint8), d(dirs, torch.float32), d(use_1d, torch.uint8)
    ev = torch.zeros(M, dtype=torch.int32, device="cuda:0")
    px = d(px0, torch.float64)
    ok, h_inv = tracking.align_batch(store, slot_d, level_d, pwb_d, px, 10, dir=dirs_d, use_1d=u1_d)
    torch.cuda.synchronize()
    ref = (px.cpu().numpy().copy(), ok.cpu().numpy().copy(), h_inv.cpu().numpy().copy())
    pxs = [d(px0, torch.float64) for _ in range(reps)]
    for _ in range(20):
        tracking.align_batch(store, slot_d, level_d, pwb_d, d(px0, torch.float64), 10, dir=dirs_d, use_1d=u1_d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        tracking.align_batch(store, slot_d, level_d, pwb_d, pxs[r], 10, dir=dirs_d, use_1d=u1_d)
    e1.record()
    torch.cuda.synchronize()
    import hashlib
    print("ALIGNBENCH " + json.dumps({"us_per_launch": 1e3 * e0.elapsed_time(e1) / reps, "M": M, "converged": int(ref[1].sum()),
                                      "digest": hashlib.sha256(ref[0].tobytes() + ref[1].tobytes() + ref[2].tobytes()).hexdigest()[:16]}))


def main():
    M, reps = int(_arg("M", "330")), int(_arg("reps", "300"))
    names = [a for a in sys.argv[1:] if "=" not in a] or ["main"]
    first = None
    for rep in range(2):
        for name in names:
            lib = os.path.join(ROOT, "rpg_svo_amd", "lib", "libsvo_hip.so") if name == "main" else os.path.join(ROOT, "build", "variants", f"lib{name}.so")
            # under rocprofv3 --kernel-trace --stats: the kernel's own average duration (back-to-back launches from Python are
            # bound by the host's 18 us per call, not by an 11 us kernel)
            import glob, csv, shutil, tempfile
            out = tempfile.mkdtemp(prefix="alignbench_", dir="/tmp")
            p = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", out, "-o", "t", "--",
                                sys.executable, os.path.abspath(__file__), "--child", str(M), str(reps)], env=dict(os.environ, SVO_HIP_LIB=lib, TMPDIR="/tmp"),
                               capture_output=True, text=True, timeout=300, cwd="/tmp")
            r = next((json.loads(l[11:]) for l in p.stdout.splitlines() if l.startswith("ALIGNBENCH ")), {"error": p.stderr[-300:]})
            kern = {}
            for f in glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if "align" in row["Name"]:
                        kern[row["Name"].split("(")[0][-28:]] = (int(row["Calls"]), round(float(row["AverageNs"]) / 1e3, 2))
            shutil.rmtree(out, ignore_errors=True)
            first = first or r.get("digest")
            print(f"[{rep}] {name:28s} kernel avg us {kern}  ({r.get('us_per_launch', float('nan')):.1f} us per call from Python), M {M}, "
                  f"converged {r.get('converged')}, same bits as the first library: {r.get('digest') == first} {r.get('error', '')}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        main()
