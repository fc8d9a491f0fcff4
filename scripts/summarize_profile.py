#!/usr/bin/env python
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into small files for profiles/."""
import csv
import glob
import json
import os
import sys

out, summ, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def find(sub, pat):
    r = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return r[0] if r else None


res = {}
ks = find("trace", "*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(summ, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write(open(ks).read())
    res["kernel_stats"] = rows[:8]
kt = find("trace", "*kernel_trace.csv")
if kt:
    rows = list(csv.DictReader(open(kt)))
    agg = {}
    for r in rows:
        name = r.get("Kernel_Name", "?")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(name, dict(n=0, us=0.0, vgpr=r.get("VGPR_Count"), sgpr=r.get("SGPR_Count"),
                                      lds=r.get("LDS_Block_Size"), wg=r.get("Workgroup_Size"), grid=r.get("Grid_Size")))
        a["n"] += 1
        a["us"] += d
    for a in agg.values():
        a["avg_us"] = a["us"] / a["n"]
    res["kernel_trace"] = agg
for sub, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", None)):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        continue
    rows = list(csv.DictReader(open(cc)))
    agg = {}
    for r in rows:
        name = r.get("Kernel_Name", "?")
        cn = r.get("Counter_Name")
        v = float(r.get("Counter_Value", 0))
        a = agg.setdefault(name, {}).setdefault(cn, [0, 0.0])
        a[0] += 1
        a[1] += v
    res["pmc_" + sub] = {k: {c: dict(dispatches=v[0], total=v[1], per_dispatch=v[1] / v[0]) for c, v in d.items()} for k, d in agg.items()}
json.dump(res, open(os.path.join(summ, f"{tag}_profile_summary.json"), "w"), indent=1)
print(json.dumps({k: (v if k != "kernel_stats" else "...") for k, v in res.items()}, indent=1)[:6000])
