#!/usr/bin/env python3
"""The per-kernel table of the full-track step from a bench details file:
python scripts/full_track_kernels.py [bench_details.json]
ms (rocprofv3 kernel trace), fraction of the HBM roofline on compulsory bytes, counter traffic over compulsory bytes,
VALU busy (3 cycles per instruction), wave cycles waiting, LDS port busy (SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES) and the
part of the port cycles lost to bank conflicts."""
import json, sys
d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "bench_details.json"))
ft = d["full_track"]
f = lambda v, fmt="{:.2f}": "-" if v is None else fmt.format(v)
print(f"full track {ft['ms_per_step']:.2f} ms per step of {ft['frames_per_step']} frames; stages (ms): " +
      ", ".join(f"{k} {v:.3f}" for k, v in ft["stages_ms"].items()))
print(f"{'kernel':44s} {'ms':>6s} {'frac':>5s} {'traf/comp':>9s} {'valu':>5s} {'wait':>5s} {'lds':>5s} {'confl':>5s}")
for k, v in ft["kernels"].items():
    valu, lds = v.get("valu") or {}, v.get("lds") or {}
    busy = lds.get("port_busy_frac_vs_busy_cu_cycles", lds.get("port_busy_frac"))  # (the first key: files of r04u)
    print(f"{k:44s} {v['ms']:6.3f} {f(v.get('frac')):>5s} {f(v.get('traffic_over_compulsory')):>9s} "
          f"{f(valu.get('busy_frac_at_3_cycles_per_instruction')):>5s} {f(valu.get('wave_cycles_waiting_frac')):>5s} "
          f"{f(busy):>5s} {f(lds.get('bank_conflict_frac_of_port_cycles')):>5s}")
