#!/bin/bash
# usage: scripts/r03_variants.sh <outdir under gpurun_out> <extras> <lib or "main"> ...   -- headline + extras per library variant
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/$1; mkdir -p "$O"; EX=$2; shift 2
cd $R
for v in "$@"; do
  if [ "$v" = main ]; then unset SVO_HIP_LIB; else export SVO_HIP_LIB=$R/build/variants/lib$v.so; fi
  python bench.py --extras $EX --no-cpu-baseline --steps 20 > $O/bench_$v.json 2> $O/bench_$v.err
  python - "$O/bench_$v.json" "$v" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], "FAILED", e); sys.exit(0)
ft = d.get("full_track", {})
print(sys.argv[2], "K1 ms %.4f" % d["roofline"]["ms"], "full %.3f" % ft.get("ms_per_step", float("nan")),
      {k: round(v, 3) for k, v in ft.get("stages_ms", {}).items() if v > 0.1}, "k0", d.get("k0_pyramid", {}).get("ms"))
PY
done
