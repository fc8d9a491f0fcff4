#!/bin/bash
# Round 6, GPU call r06aj: test_pose_optimize with the wave kernel's three instantiations (rows of 250 / 128 / 64 observations)
# against both checkers and three cameras, as committed and on eight other scenes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06aj; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
for k in 0 1 2 3 4 5 6 7 8; do
  out=$(SVO_TEST_FUZZ=$k timeout 900 python -m pytest tests/test_tracking_gpu.py -q -m gpu -rf -k "pose_optimize" 2>&1)
  echo "fuzz $k: $(echo "$out" | tail -1)"
  echo "$out" | grep -E "^(E  |FAILED|ERROR)" | cut -c1-220 | head -24
done
} 2>&1 | tee $O/log.txt
