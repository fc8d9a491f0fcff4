#!/usr/bin/env python3
"""Does any kernel pay for LDS accesses off their natural alignment?  (An 8- / 12- / 16-byte DS access off its alignment, or
a 2-byte one at an odd address, is replayed by the DS unit: SQ_LDS_UNALIGNED_STALL counts the cycles; the 2-byte reads of
profiles/r06ac_* made update_seeds 32 % slower at fewer instructions.)  One rocprofv3 --pmc pass per pipeline of bench.py
(full-track step, headline step), counters summed per kernel: stall cycles against the LDS array's active cycles.
usage (GPU box, repository root): python scripts/lds_unaligned.py [out.json]"""
import csv, glob, json, os, re, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CTRS = ["SQ_LDS_UNALIGNED_STALL", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS"]
OURS = "sia_|epi_scan|align_|warp_kernel|match_|seed_|pose_opt|reproject|pyramid|compose"  # (the bench's set-up runs torch kernels: not instrumented)


def one(pipeline):
    d = tempfile.mkdtemp(prefix="svo_lds_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", *CTRS, "--kernel-include-regex", OURS, "--output-format", "csv", "-d", d, "-o", "lds", "--", sys.executable, os.path.join(ROOT, "bench.py"),
           "--steps", "2", "--warmup", "0", "--no-cpu-baseline", "--extras", "none", "--pmc-child", "1", "--pipeline", pipeline]
    p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=600)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if p.returncode != 0 or not files:
        return {"error": f"rc={p.returncode}: {p.stderr[-3000:]}"}
    out = {}
    with open(files[0]) as fh:
        for row in csv.DictReader(fh):
            m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", row["Kernel_Name"])
            k = m.group(1) if m else row["Kernel_Name"][:60]
            e = out.setdefault(k, {c: 0.0 for c in CTRS} | {"dispatch_ids": set()})
            if row["Counter_Name"] in CTRS:
                e[row["Counter_Name"]] += float(row["Counter_Value"])
            e["dispatch_ids"].add(row["Dispatch_Id"])
    shutil.rmtree(d, ignore_errors=True)
    for k, e in out.items():
        e["dispatches"] = len(e.pop("dispatch_ids"))
        idx = e["SQ_LDS_IDX_ACTIVE"]
        e["unaligned_stall_over_idx_active"] = e["SQ_LDS_UNALIGNED_STALL"] / idx if idx > 0 else None
        e["bank_conflict_over_idx_active"] = e["SQ_LDS_BANK_CONFLICT"] / idx if idx > 0 else None
    return out


if __name__ == "__main__":
    res = {p: one(p) for p in ("full", "align")}
    for p, ks in res.items():
        print(f"== pipeline {p}")
        if "error" in ks:
            print(ks["error"]); continue
        for k, e in sorted(ks.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"]):
            if e["SQ_INSTS_LDS"] > 0:
                print(f"{k[:60]:60s} lds_instr {e['SQ_INSTS_LDS']:.3g} idx_active {e['SQ_LDS_IDX_ACTIVE']:.3g} unaligned_stall {e['SQ_LDS_UNALIGNED_STALL']:.3g} "
                      f"({e['unaligned_stall_over_idx_active']}) bank_conflict {e['bank_conflict_over_idx_active']}")
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)
