#!/bin/bash
# Round 6, GPU call r06ai: the map-mirror, FAST and pyramid GPU suites (bit-exact against both checkers) as committed and
# on eight other seeds (SVO_TEST_FUZZ).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06ai; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
for k in 0 1 2 3 4 5 6 7 8; do
  out=$(SVO_TEST_FUZZ=$k timeout 900 python -m pytest tests/test_map_mirror_gpu.py tests/test_fast_gpu.py tests/test_pyramid_gpu.py -q -m gpu -rf 2>&1)
  echo "fuzz $k: $(echo "$out" | tail -1)"
  echo "$out" | grep -E "^(E  |FAILED|ERROR)" | cut -c1-220 | head -24
done
} 2>&1 | tee $O/log.txt
