#!/bin/bash
# Round 6, GPU call r06af: r06ae again on a whole tree (the snapshot of r06ae caught tests/helpers.py mid-edit):
# SQ_LDS_UNALIGNED_STALL per kernel, the tracking + sparse-alignment suites as committed (K4 with one selection) and on
# twelve other scenes (ill-posed pose frames and the sigma2 bound as the first fuzz run found them).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06af; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== LDS alignment stalls"
timeout 900 python scripts/lds_unaligned.py $O/lds_unaligned.json 2>&1 | tail -60
echo "== tracking suite as committed"
timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_sparse_align_gpu.py -q -m gpu 2>&1 | tail -3
echo "== tracking suite, other scenes"
bash scripts/fuzz_tracking.sh gpu 1 12
} 2>&1 | tee $O/log.txt
