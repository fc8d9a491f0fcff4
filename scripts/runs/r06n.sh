#!/bin/bash
# Round 6, fourteenth GPU call: the frame chain on / off, each in a process of its own (bench.dropin_hip_only), alternating.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06n; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
for rep in 1 2 3; do
  for ch in 1 0; do
    for mp in sync deferred; do
      echo -n "chain=$ch mapper=$mp: "
      SVO_HIP_CHAIN=$ch SVO_HIP_MAPPER=$mp timeout 300 python -c "
import sys, json; sys.path.insert(0, '$R'); import bench; print(json.dumps(bench.dropin_hip_only(600, '')))" 2>/dev/null | tail -1 | cut -c1-200
    done
  done
done
} 2>&1 | tee $O/log.txt
