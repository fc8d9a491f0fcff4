#!/bin/bash
# Round 6, twentieth GPU call: the scan's chunk of 256 seeds, and the combinations with the alignment's 128-trial workgroups.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06t; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== full track untraced (three rounds)"
bash scripts/full_variants.sh main sc512 sc256 ab128sc512 ab128sc256 main sc512 sc256 ab128sc512 ab128sc256 main sc512 sc256 ab128sc512 ab128sc256 2>&1 | cut -c1-230
} 2>&1 | tee $O/log.txt
