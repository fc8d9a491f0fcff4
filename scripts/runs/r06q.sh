#!/bin/bash
# Round 6, seventeenth GPU call: where a wave of the scan kernel (warp + epipolar scan) spends its clocks on the representative
# full-track workload (-DSCAN_PROFILE build of depth_filter.hip), the seeds' alignment histogram, and the drop-in leg with the
# runs in processes of their own.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06q; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
for v in scanprof main scanprof; do
  lib=$PWD/build/variants/lib$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo "== full track, library $v"
  SVO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --extras full --full-line --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['full_track']
print('step_ms', round(f['ms_per_step'],3), {k: round(v,3) for k,v in f['stages_ms'].items()})
print('scan_profile', json.dumps(f.get('scan_profile')))
print('seed alignment', json.dumps(f['rooflines']['update_seeds'].get('alignment')))
print('scan positions histogram per frame', json.dumps(f['rooflines']['update_seeds'].get('seeds_per_frame_by_scanned_positions')))"
done
echo "== drop-in leg of the bench"
timeout 1200 python bench.py --extras dropin --full-line --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['dropin_sequence']
for k in ('median_ms_per_frame_cpu_reference','median_ms_per_frame_hip_dropin','median_ms_per_frame_hip_dropin_deferred_mapper','median_ms_per_frame_in_a_process_of_its_own','median_ms_per_frame_hip_dropin_without_the_frame_chain','median_ms_per_frame_mapper_thread','frame_period_ms_back_to_back','frame_chain','early_mapper','predicted_pose_refinements','deferred_mapper_trajectory_identical','leg_seconds'): print(k, c.get(k))" | tee $O/dropin_leg.txt
} 2>&1 | tee $O/log.txt
