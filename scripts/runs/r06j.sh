#!/bin/bash
# Round 6, tenth GPU call: the split K1's exchange polled by every wave (the tree) against one wave + a barrier (variant); the
# frame chain behind the sparse alignment on the GPU (tests, then the drop-in leg of the bench: chained vs SVO_HIP_CHAIN=0).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06j; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
for v in main onewave main onewave; do
  lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so; [ "$v" == "onewave" ] && lib=$PWD/build/variants/libsvo_hip_SIA_X_ONEWAVE.so
  echo "== configs[3] leg: $v"
  SVO_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --extras config3 --full-line --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['config3_xga5_b64']
print({k: c[k] for k in ('ms_per_step','frames_per_s','mean_gn_iterations_per_frame','median_pose_error_vs_gt')})"
done
echo "== parity of the variant"
SVO_HIP_LIB=$PWD/build/variants/libsvo_hip_SIA_X_ONEWAVE.so timeout 600 python -m pytest tests/test_full_size_gpu.py -q -m gpu -k config3 2>&1 | tail -3
echo "== drop-in GPU tests"
timeout 1500 python -m pytest tests/test_dropin_pipeline.py tests/test_replay_gpu.py -q -m gpu -x -s 2>&1 | grep -v "DepthFilter\|^$" | tail -25
echo "== drop-in leg of the bench"
timeout 900 python bench.py --no-cpu-baseline --extras dropin --full-line --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['dropin_sequence']
for k in ('median_ms_per_frame_cpu_reference','median_ms_per_frame_hip_dropin','median_ms_per_frame_hip_dropin_deferred_mapper','median_ms_per_frame_hip_dropin_without_the_frame_chain','frame_period_ms_back_to_back','frame_chain','predicted_pose_refinements','deferred_mapper_trajectory_identical','host_vs_device_us_per_call','first_frame_with_a_different_decision','map_mirror'): print(k, c.get(k))" | tee $O/dropin_leg.txt
} 2>&1 | tee $O/log.txt
