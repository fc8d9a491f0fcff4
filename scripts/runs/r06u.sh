#!/bin/bash
# Round 6, twenty-first GPU call: the poses of a (reference, current) pair formed once per run of seeds (seed_prepare_kernel's
# workgroups of 256, seed_finish reading them) against the tree before it (build/variants/libab128.so): parity, then the
# full-track step alternating, then every dispatch under a kernel trace.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06u; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== parity: tracking + full size + reference style + golden + replay + the drop-in's seed store"
timeout 1500 python -m pytest tests/test_tracking_gpu.py tests/test_full_size_gpu.py tests/test_reference_style_gpu.py tests/test_golden_track.py tests/test_replay_gpu.py -q -m gpu 2>&1 | tail -4
timeout 900 python -m pytest tests/test_dropin_pipeline.py -q -m gpu -k "seed_store or trajectory_matches or deferred" 2>&1 | tail -3
echo "== full track untraced (three rounds)"
bash scripts/full_variants.sh ab128 main ab128 main ab128 main 2>&1 | cut -c1-230
echo "== full track under a kernel trace: every dispatch"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o trace -- python $R/bench.py --pipeline full --steps 4 --warmup 2 --no-cpu-baseline --extras none > $O/full_bench_under_trace.json 2> $O/trace_full.err)
python scripts/kernel_last_steps.py $O/trace_full 6 | tee $O/full_kernel_last_steps.txt
rm -rf $O/trace_full
} 2>&1 | tee $O/log.txt
