#!/bin/bash
# Round 6, sixth GPU call: K1 split over four workgroups (configs[3] at B = 64): parity, then the configs[3] leg with the split
# on and off; the distorted scan at three waves per SIMD (the tree) against four (build/variants/libsvo_hip_scan4.so).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06f; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== parity: K1 suites"
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_sparse_align_gpu.py -q -m gpu 2>&1 | tail -5
for sp in 1 0 1 0; do
  echo "== configs[3] leg, SVO_HIP_K1_SPLIT=$sp"
  SVO_HIP_K1_SPLIT=$sp timeout 300 python bench.py --no-cpu-baseline --extras config3 --full-line --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['config3_xga5_b64']
print({k: c[k] for k in ('ms_per_step','frames_per_s','mean_gn_iterations_per_frame','median_pose_error_vs_gt','frames_per_s_at_batch_1024')}, c.get('split4_latency_floor'))"
done
for v in main scan4 main scan4; do
  lib=$PWD/build/variants/libsvo_hip_$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo "== reference cameras, full track: $v"
  SVO_HIP_LIB=$lib timeout 400 python bench.py --no-cpu-baseline --extras cameras --full-line --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['reference_cameras']['cameras']
print({k: (round(v['full_track']['stages_ms']['update_seeds'],3), round(v['full_track']['ms_per_step'],3)) for k,v in c.items() if 'full_track' in v})"
done
} 2>&1 | tee $O/log.txt
