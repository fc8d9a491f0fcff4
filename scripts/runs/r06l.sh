#!/bin/bash
# Round 6, twelfth GPU call: the split K1 with persistent epochs (no per-launch memset) and side-by-side H polls; the frame
# chain with the keyframes ranked on the device: tests, the drop-in leg (chained vs SVO_HIP_CHAIN=0), the frame's timeline.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06l; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== parity: K1 suites + the new entry point"
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_sparse_align_gpu.py tests/test_tracking_gpu.py -q -m gpu 2>&1 | tail -5
for sp in 1 0 1 0; do
  echo "== configs[3] leg, SVO_HIP_K1_SPLIT=$sp"
  SVO_HIP_K1_SPLIT=$sp timeout 300 python bench.py --no-cpu-baseline --extras config3 --full-line --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['config3_xga5_b64']
print({k: c[k] for k in ('ms_per_step','frames_per_s','mean_gn_iterations_per_frame','median_pose_error_vs_gt')})"
done
echo "== drop-in GPU tests"
timeout 1500 python -m pytest tests/test_dropin_pipeline.py tests/test_replay_gpu.py -q -m gpu -x -s 2>&1 | grep -v "INFO\|^$" | tail -12
echo "== drop-in leg of the bench"
timeout 1200 python bench.py --extras dropin --full-line --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['dropin_sequence']
for k in ('median_ms_per_frame_cpu_reference','median_ms_per_frame_hip_dropin','median_ms_per_frame_hip_dropin_deferred_mapper','median_ms_per_frame_hip_dropin_without_the_frame_chain','frame_period_ms_back_to_back','frame_chain','predicted_pose_refinements','deferred_mapper_trajectory_identical','host_vs_device_us_per_call'): print(k, c.get(k))" | tee $O/dropin_leg.txt
echo "== the frame's timeline"
(cd /tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace_timeline -- python $R/scripts/dropin_trace.py frames=600 > $R/$O/dropin_traced_run.txt 2> $R/$O/trace_timeline.err)
python scripts/dropin_trace.py --report $O/trace_timeline frames=600 > $O/dropin_frame_timeline_600.txt
head -90 $O/dropin_frame_timeline_600.txt | cut -c1-140
rm -rf $O/trace_timeline
} 2>&1 | tee $O/log.txt
