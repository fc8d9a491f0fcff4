#!/bin/bash
# Round 6, first GPU call: the new headline (K1 + K4 on pipeline matches), the reference-cameras leg, the sigma2 deviation
# the tracking test measures, the seed-store tests.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06a; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== tracking + dropin store tests"
timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_dropin_pipeline.py -q -m gpu -s -k "update_seeds or seed_store" 2>&1 | grep -E "update_seeds\[|passed|failed|Error" | tail -30
echo "== the driver's command"
(time python3 bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_driver_style_line.json 2> $O/bench_driver_style.err; cp bench_details.json $O/bench_driver_style_details.json
tail -c 4000 $O/bench_driver_style_line.json; echo; tail -4 $O/bench_driver_style.err
tail -30 bench_stderr.log
} 2>&1 | tee $O/log.txt
