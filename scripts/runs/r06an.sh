#!/bin/bash
# Round 6, GPU call r06an: the tracking and sparse-alignment suites on twelve more scenes (SVO_TEST_FUZZ=13..24), final tree.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06an; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{ bash scripts/fuzz_tracking.sh gpu 13 24; } 2>&1 | tee $O/log.txt
