#!/bin/bash
# Round 6, twenty-third GPU call: seed_prepare with the pair's poses per seed again (64-lane workgroups, no LDS, no barrier;
# the first seed of a run inside a wave files them for seed_finish) against the workgroup-of-256 / one-lane-per-run form of
# the call before (build/variants/libprep256.so): parity, the full-track step alternating, every dispatch, the drop-in.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06v; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== parity: tracking + full size + reference style + golden + replay"
timeout 1500 python -m pytest tests/test_tracking_gpu.py tests/test_full_size_gpu.py tests/test_reference_style_gpu.py tests/test_golden_track.py tests/test_replay_gpu.py -q -m gpu 2>&1 | tail -4
echo "== full track untraced (three rounds)"
bash scripts/full_variants.sh prep256 main ab128 prep256 main ab128 prep256 main ab128 2>&1 | cut -c1-230
echo "== full track under a kernel trace: every dispatch"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o trace -- python $R/bench.py --pipeline full --steps 4 --warmup 2 --no-cpu-baseline --extras none > $O/full_bench_under_trace.json 2> $O/trace_full.err)
python scripts/kernel_last_steps.py $O/trace_full 6 | head -4 | tee $O/full_kernel_last_steps.txt
rm -rf $O/trace_full
echo "== single-stream drop-in, processes of their own"
for rep in 1 2 3; do for v in prep256 main; do
  lib=$PWD/build/variants/lib$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo -n "$v: "; SVO_HIP_LIB=$lib timeout 300 python -c "
import sys, json; sys.path.insert(0, '$R'); import bench; print(json.dumps(bench.dropin_hip_only(600, '')))" 2>/dev/null | tail -1 | cut -c1-60
done; done
} 2>&1 | tee $O/log.txt
