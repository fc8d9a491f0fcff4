#!/usr/bin/env python
"""Is the two-mode spread of the single-stream frame between processes a matter of WHERE the process runs?  The drop-in
(600 frames, hip flavour, a child process per run) pinned to the NUMA node of the visible GPU, to the other node, unpinned.
GPU box:  python scripts/numa_placement.py [reps=4]"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("reps=")), 4))
import torch  # noqa: E402

p = torch.cuda.get_device_properties(0)
bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
node, local = None, None
for d in glob.glob("/sys/bus/pci/devices/*"):
    if os.path.basename(d).lower() == bdf:
        node = int(open(d + "/numa_node").read())
        local = open(d + "/local_cpulist").read().strip()
nodes = {int(os.path.basename(n)[4:]): open(n + "/cpulist").read().strip() for n in glob.glob("/sys/devices/system/node/node[0-9]*")}
other = next((c for k, c in sorted(nodes.items()) if k != node), None)
print(json.dumps({"gpu": bdf, "numa_node": node, "local_cpulist": local, "nodes": nodes, "allowed_cpus": len(os.sched_getaffinity(0))}))
code = (f"import sys, json; sys.path.insert(0, {ROOT!r}); import bench; r = bench.dropin_hip_only(600, ''); "
        "print(json.dumps({k: r[k] for k in ('tot_time', 'reproject')}))")
for rep in range(reps):
    for name, cpus in (("unpinned", None), ("local", local), ("other", other)):
        if name != "unpinned" and not cpus:
            continue
        cmd = ([] if cpus is None else ["taskset", "-c", cpus]) + [sys.executable, "-c", code]
        q = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        out = q.stdout.strip().splitlines()[-1] if q.returncode == 0 and q.stdout.strip() else ("failed: " + q.stderr[-200:])
        print(f"{name:9s} {out}", flush=True)
