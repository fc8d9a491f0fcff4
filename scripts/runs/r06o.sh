#!/bin/bash
# Round 6, fifteenth GPU call: the chain's inputs on the second stream (beside K1): chain on / off in processes of their own,
# then the chain's GPU test.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06o; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
for rep in 1 2 3; do
  for ch in 1 0; do
      echo -n "chain=$ch: "
      SVO_HIP_CHAIN=$ch timeout 300 python -c "
import sys, json; sys.path.insert(0, '$R'); import bench; print(json.dumps(bench.dropin_hip_only(600, '')))" 2>/dev/null | tail -1 | cut -c1-120
  done
done
timeout 900 python -m pytest tests/test_dropin_pipeline.py -q -m gpu -k "frame_chain or arena_modes" -s 2>&1 | grep -v "INFO\|^$" | tail -5
} 2>&1 | tee $O/log.txt
