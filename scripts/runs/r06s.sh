#!/bin/bash
# Round 6, nineteenth GPU call: the alignment's workgroups of 128 / 256 trials (the later phases are launched for full queues:
# 205 k mostly empty 64-lane workgroups cost the dispatcher ~90 + ~45 us per update of 13 M seeds) and the scan's chunk of
# 512 / 2048 seeds, against the tree; parity of the tracking suite on the 256 variant.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06s; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== parity on ab256"
SVO_HIP_LIB=$PWD/build/variants/libab256.so timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_full_size_gpu.py -q -m gpu 2>&1 | tail -3
echo "== full track untraced (three rounds)"
bash scripts/full_variants.sh main ab128 ab256 sc512 sc2048 main ab128 ab256 sc512 sc2048 main ab128 ab256 sc512 sc2048 2>&1 | cut -c1-230
} 2>&1 | tee $O/log.txt
