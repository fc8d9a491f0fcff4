#!/bin/bash
# Round 6, twenty-sixth GPU call: the depth filter's early update in two phases (tables marshalled and uploaded before the pose
# optimizer's result has arrived, kernels launched with the pose by value) against one phase and off, alternating processes;
# the drop-in GPU tests.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06x; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
for rep in 1 2 3 4; do
  for em in 1 1phase 0; do
    echo -n "early_mapper=$em: "
    SVO_HIP_EARLY_MAPPER=$em timeout 300 python -c "
import sys, json; sys.path.insert(0, '$R'); import bench; print(json.dumps(bench.dropin_hip_only(600, '')))" 2>/dev/null | tail -1 | cut -c1-60
  done
done
echo "== drop-in GPU tests"
timeout 1800 python -m pytest tests/test_dropin_pipeline.py tests/test_replay_gpu.py tests/test_tracking_gpu.py -q -m gpu -x -s 2>&1 | grep -v "INFO\|WARN\|^$\|update_seeds\[" | tail -12
} 2>&1 | tee $O/log.txt
