#!/bin/bash
# Round 6, GPU call r06ag: K4 -- one selection (k4a = the tree of r06af), + no selects in the factor / maximum instead of a
# select in the square root / reciprocal instead of divisions in the two passes around the iterations (k4b), + the last
# observation slot skipped when the frame does not reach it (main) -- headline step in alternating processes; the tracking and
# sparse-alignment suites as committed (the wave leg of test_pose_optimize now RUNS the wave kernel: rows of 250) and on
# eight other scenes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06ag; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== suites as committed"
timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_sparse_align_gpu.py tests/test_golden_track.py tests/test_reference_style_gpu.py -q -m gpu -rf 2>&1 | tail -5
echo "== headline, alternating processes"
for rep in 1 2 3; do for v in k4a k4b main; do
  lib=$PWD/build/variants/lib$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo -n "$v: "; SVO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --extras none 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('stages_ms'), d['parity'].get('refined_pose_se3_lognorm_max') if 'parity' in d else None)"
done; done
echo "== other scenes"
bash scripts/fuzz_tracking.sh gpu 1 8
} 2>&1 | tee $O/log.txt
