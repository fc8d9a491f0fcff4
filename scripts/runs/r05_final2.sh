#!/bin/bash
# Round 5, the bench lines of the final tree: the driver's own command, then the default run (its f64_partials leg needs
# the -DSIA_F64_PARTIALS library __graft_entry__.build() makes: the first final call ran with a stale one).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r05y; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== the driver's command"
(time python3 bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_driver_style_line.json 2> $O/bench_driver_style.err; cp bench_details.json $O/bench_driver_style_details.json
tail -c 300 $O/bench_driver_style_line.json; echo; tail -4 $O/bench_driver_style.err
echo "== default bench"
(time python bench.py) > $O/bench_default_line.json 2> $O/bench_default.err; cp bench_details.json $O/bench_default_details.json
tail -c 300 $O/bench_default_line.json; echo; tail -4 $O/bench_default.err
python - <<'PY'
import json
for n in ("driver_style", "default"):
    d = json.loads(open(f"gpurun_out/r05y/bench_{n}_line.json").read().strip().splitlines()[-1])
    print(n, "value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], "f64", d.get("roofline_f64_build"))
PY
} 2>&1 | tee $O/log.txt
