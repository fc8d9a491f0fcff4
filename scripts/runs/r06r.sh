#!/bin/bash
# Round 6, eighteenth GPU call: parity of the tree (powers of two by exponent bits, the scan's DPP minimum), then the full-track
# step under a kernel trace with the durations of every dispatch (the alignment's three launches one by one).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06r; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== parity: tracking + full size + reference style + golden + replay"
timeout 1500 python -m pytest tests/test_tracking_gpu.py tests/test_full_size_gpu.py tests/test_reference_style_gpu.py tests/test_golden_track.py tests/test_replay_gpu.py -q -m gpu 2>&1 | tail -4
echo "== full track untraced"
bash scripts/full_variants.sh main main 2>&1 | cut -c1-260
echo "== full track under a kernel trace: every dispatch"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o trace -- python $R/bench.py --pipeline full --steps 4 --warmup 2 --no-cpu-baseline --extras none > $O/full_bench_under_trace.json 2> $O/trace_full.err)
python scripts/kernel_last_steps.py $O/trace_full 12 | tee $O/full_kernel_last_steps.txt
rm -rf $O/trace_full
} 2>&1 | tee $O/log.txt
