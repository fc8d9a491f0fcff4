#!/bin/bash
# Round 6, GPU call r06ac: the affine warp's samples read as two 2-byte LDS reads per sample instead of four 1-byte reads
# (-DWARP_U16_READS in depth_filter + matcher: build/variants/libu16.so) against the tree: parity suites on the variant,
# the full-track step in alternating processes, the single-stream drop-in in alternating processes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06ac; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== parity on u16"
SVO_HIP_LIB=$PWD/build/variants/libu16.so timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_full_size_gpu.py tests/test_golden_track.py -q -m gpu 2>&1 | tail -3
echo "== full track untraced (four rounds)"
bash scripts/full_variants.sh main u16 main u16 main u16 main u16 2>&1 | cut -c1-230
echo "== drop-in, alternating processes"
for rep in 1 2 3 4; do for v in main u16; do
  lib=$PWD/build/variants/lib$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo -n "$v: "; SVO_HIP_LIB=$lib timeout 300 python -c "
import sys, json; sys.path.insert(0, '$R'); import bench; print(json.dumps(bench.dropin_hip_only(600, '')))" 2>/dev/null | tail -1 | cut -c1-60
done; done
} 2>&1 | tee $O/log.txt
