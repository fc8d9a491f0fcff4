#!/bin/bash
# Round 6, eleventh GPU call: the drop-in leg of the bench (chained vs SVO_HIP_CHAIN=0), then the full-track step with the
# alignment's phase boundaries moved and with the packed warp sample arithmetic.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06k; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== drop-in leg of the bench"
timeout 1200 python bench.py --extras dropin --full-line --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['dropin_sequence']
for k in ('median_ms_per_frame_cpu_reference','median_ms_per_frame_hip_dropin','median_ms_per_frame_hip_dropin_deferred_mapper','median_ms_per_frame_hip_dropin_without_the_frame_chain','frame_period_ms_back_to_back','frame_chain','predicted_pose_refinements','deferred_mapper_trajectory_identical','host_vs_device_us_per_call','first_frame_with_a_different_decision','map_mirror'): print(k, c.get(k))" | tee $O/dropin_leg.txt
echo "== full track untraced: alignment phases, packed warp"
bash scripts/full_variants.sh main svo_hip_ph2_4 svo_hip_ph2_5 svo_hip_ph1_3 svo_hip_ph2_4_7 svo_hip_ph1_2_4 svo_hip_WARP_PK main svo_hip_ph2_4 svo_hip_ph2_5 svo_hip_ph1_3 svo_hip_ph2_4_7 svo_hip_ph1_2_4 svo_hip_WARP_PK 2>&1 | cut -c1-230
echo "== parity of the variants (tracking suite)"
for v in ph2_4_7 WARP_PK; do SVO_HIP_LIB=$PWD/build/variants/libsvo_hip_$v.so timeout 900 python -m pytest tests/test_tracking_gpu.py -q -m gpu -x 2>&1 | tail -2; done
} 2>&1 | tee $O/log.txt
