#!/usr/bin/env python3
"""Single-stream drop-in frame time under runtime settings of the HIP / HSA stack (environment variables read at
runtime start-up) and of the host layer (SVO_HIP_*): the 600-frame sequence of bench.py's dropin leg, rendered once,
then one child process per setting (a setting is read once per process), every setting run twice in alternation.

    python scripts/env_knobs.py [frames=600] [flavour=hip] [out=gpurun_out/r05h]

What it answers: of a ~300 us frame ~90 us are launches, copies and wake-ups, not kernels -- does a runtime switch
(polling instead of interrupts, kernel arguments in device memory, number of hardware queues, blit instead of SDMA
copies, active-wait time) move that, and is it a deployment setting worth naming in INTEGRATION.md?"""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))

SETTINGS = [
    ("default", {}),
    ("HSA_ENABLE_INTERRUPT=0", {"HSA_ENABLE_INTERRUPT": "0"}),
    ("ROC_ACTIVE_WAIT_TIMEOUT=1000", {"ROC_ACTIVE_WAIT_TIMEOUT": "1000"}),
    ("HSA_ENABLE_INTERRUPT=0 ROC_ACTIVE_WAIT_TIMEOUT=1000", {"HSA_ENABLE_INTERRUPT": "0", "ROC_ACTIVE_WAIT_TIMEOUT": "1000"}),
    ("HIP_FORCE_DEV_KERNARG=0", {"HIP_FORCE_DEV_KERNARG": "0"}),
    ("HIP_FORCE_DEV_KERNARG=1", {"HIP_FORCE_DEV_KERNARG": "1"}),
    ("GPU_MAX_HW_QUEUES=1", {"GPU_MAX_HW_QUEUES": "1"}),
    ("GPU_MAX_HW_QUEUES=2", {"GPU_MAX_HW_QUEUES": "2"}),
    ("GPU_MAX_HW_QUEUES=8", {"GPU_MAX_HW_QUEUES": "8"}),
    ("HSA_ENABLE_SDMA=0", {"HSA_ENABLE_SDMA": "0"}),
    ("ROC_CPU_WAIT_FOR_SIGNAL=0", {"ROC_CPU_WAIT_FOR_SIGNAL": "0"}),
    ("AMD_DIRECT_DISPATCH=0", {"AMD_DIRECT_DISPATCH": "0"}),
    ("SVO_HIP_ARENA=mapped", {"SVO_HIP_ARENA": "mapped"}),
    ("SVO_HIP_ARENA=mirrored", {"SVO_HIP_ARENA": "mirrored"}),
]


def _arg(name, default):
    for a in sys.argv[1:]:
        if a.startswith(name + "="):
            return a.split("=", 1)[1]
    return default


def child(path, flavour, defer):
    import numpy as np
    import pypipeline as pp
    z = np.load(path)
    imgs, T, range0 = z["imgs"], z["T"], z["range0"]
    cam = types.SimpleNamespace(width=int(z["cam"][0]), height=int(z["cam"][1]), fx=float(z["cam"][2]), fy=float(z["cam"][3]),
                                cx=float(z["cam"][4]), cy=float(z["cam"][5]))
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 2)
    pp.run_sequence(flavour, cam, imgs[:20], T[:20], range0=range0, defer_mapper=defer)
    st = {}
    res = pp.run_sequence(flavour, cam, imgs, T, stats_out=st, range0=range0, defer_mapper=defer)
    tail = res[-min(len(res) - 1, 400):]
    out = {"tot_time_median_us": float(np.median([r["t_tot_time"] for r in res[1:]]) * 1e6),
           "tot_time_median_us_last_400": float(np.median([r["t_tot_time"] for r in tail]) * 1e6),
           "frame_period_us": st["wall_ms_per_frame"] * 1e3,
           "keyframes": int(sum(r["is_keyframe"] for r in res)),
           "pose_checksum": float(np.sum(np.abs(np.stack([r["T_f_w"] for r in res])))),
           "device_us_per_call": {k: round(v["device_us"], 2) for k, v in st["stages"].items()}}
    print("ENVKNOBS " + json.dumps(out))


def run_child(path, flavour, env_extra, defer):
    env = dict(os.environ)
    env.update(env_extra)
    cmd = [sys.executable, os.path.abspath(__file__), "--child", path, flavour, "1" if defer else "0"]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    except subprocess.TimeoutExpired:
        return {"error": "timeout"}
    for line in p.stdout.splitlines():
        if line.startswith("ENVKNOBS "):
            return json.loads(line[9:])
    return {"error": (p.stderr or p.stdout)[-300:]}


def main():
    import numpy as np
    import torch
    import pypipeline as pp
    from rpg_svo_amd import synth
    n = int(_arg("frames", "600"))
    flavour = _arg("flavour", "hip")
    out_dir = os.path.join(ROOT, _arg("out", "gpurun_out/r05h"))
    os.makedirs(out_dir, exist_ok=True)
    cam = synth.Camera(752, 480, 315.5, 315.5, 376.0, 240.0)
    T = synth.make_trajectory(n, seed=5, max_step=0.02, max_rot_deg=0.3)
    imgs = synth.render(synth.make_texture(seed=12345), T, cam, device="cuda" if torch.cuda.is_available() else "cpu").cpu().numpy()
    import tempfile
    fd, path = tempfile.mkstemp(prefix="svo_env_knobs_seq_", suffix=".npz")
    os.close(fd)
    np.savez(path, imgs=imgs, T=np.asarray(T), range0=pp.range_map(cam, T[0]),
             cam=np.array([cam.width, cam.height, cam.fx, cam.fy, cam.cx, cam.cy], dtype=np.float64))
    only = _arg("only", "")
    settings = [x for x in SETTINGS if not only or x[0] == "default" or any(o in x[0] for o in only.split(","))]
    results = {name: [] for name, _ in settings}
    for rep in range(int(_arg("reps", "2"))):
        for name, env in settings:
            r = run_child(path, flavour, env, False)
            results[name].append(r)
            print(f"[{rep}] {name:55s} {json.dumps(r)[:230]}", flush=True)

    def best(name):
        v = [r["tot_time_median_us"] for r in results[name] if "tot_time_median_us" in r]
        return min(v) if v else None

    base = best("default")
    # the settings that gained more than 1.5 % on their own, together; then with the deferred mapper, against the default
    winners = {}
    for name, env in settings[1:]:
        b = best(name)
        if base and b and b < 0.985 * base and not name.startswith("SVO_HIP_"):
            winners.update(env)
    extra = {}
    if winners:
        label = " ".join(f"{k}={v}" for k, v in sorted(winners.items()))
        extra["combined: " + label] = [run_child(path, flavour, winners, False) for _ in range(2)]
        extra["combined, deferred mapper: " + label] = [run_child(path, flavour, winners, True) for _ in range(2)]
    extra["default, deferred mapper"] = [run_child(path, flavour, {}, True) for _ in range(2)]
    for k, v in extra.items():
        print(f"{k:70s} {json.dumps(v)[:300]}", flush=True)
    ref = results["default"][0].get("pose_checksum")
    same = {name: all(r.get("pose_checksum") == ref for r in rs) for name, rs in {**results, **extra}.items()}
    summary = {"frames": n, "flavour": flavour, "settings": results, "follow_up": extra, "default_best_us": base,
               "combined_env": winners, "same_trajectory_as_default": same}
    os.unlink(path)
    with open(os.path.join(out_dir, "env_knobs.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("\nsetting, best of two tot_time medians (us), change against the default, same trajectory")
    for name, _ in settings:
        b = best(name)
        print(f"  {name:55s} {b if b is None else round(b, 1)!s:>8s} {'' if not (b and base) else '%+.1f %%' % (100 * (b / base - 1)):>8s}   {same[name]}")
    for k, v in extra.items():
        vals = [r.get("tot_time_median_us") for r in v if "tot_time_median_us" in r]
        print(f"  {k:90s} {min(vals) if vals else None}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3], sys.argv[4] == "1")
    else:
        main()
