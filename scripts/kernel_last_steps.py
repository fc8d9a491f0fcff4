#!/usr/bin/env python
"""Per-dispatch durations of our kernels from a rocprofv3 --kernel-trace CSV, last N dispatches per kernel (the timed
steps of a bench run come last; set-up launches of the same kernels come first).
usage: scripts/kernel_last_steps.py <dir with *kernel_trace.csv> [N]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"]
    if "at::native" in name or "Cijk" in name or "rocprim" in name or "rocclr" in name:
        continue
    short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    per.setdefault(short, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in per.items():
    print(f"{k:50s} calls {len(v):4d}  last {n} (us): {[round(x) for x in v[-n:]]}")
