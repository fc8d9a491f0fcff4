#!/bin/bash
# Round 6, GPU call r06ad: (1) SQ_LDS_UNALIGNED_STALL per kernel of the full-track and headline steps (is any LDS access
# replayed for its alignment?), (2) the bit-exact tracking suite on eight other scenes / draws (SVO_TEST_FUZZ=1..8).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06ad; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== LDS alignment stalls"
timeout 900 python scripts/lds_unaligned.py $O/lds_unaligned.json 2>&1 | tail -40
echo "== tracking suite, other scenes"
bash scripts/fuzz_tracking.sh gpu 1 8
} 2>&1 | tee $O/log.txt
