#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace --stats run: python scripts/kernel_table.py <dir> [n_rows]"""
import csv, glob, os, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
f = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:n]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-44:]
    print(f"{name:46s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.2f} min {float(r['MinNs'])/1e3:7.2f} max {float(r['MaxNs'])/1e3:8.2f}")
