#!/bin/bash
# Round 6, fifth GPU call: the exchange microbenchmark (three load policies), then the default bench of the tree.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06e; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== cross-workgroup exchange microbenchmark"
timeout 200 build/xwg_exchange_ubench 200 | tee $O/xwg_exchange_ubench.json
echo "== the driver's command"
(time python3 bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_driver_style_line.json 2> $O/bench_driver_style.err; cp bench_details.json $O/bench_driver_style_details.json
tail -c 6000 $O/bench_driver_style_line.json; echo; tail -4 $O/bench_driver_style.err
} 2>&1 | tee $O/log.txt
