#!/bin/bash
# Round 6, second GPU call: the fused warp + scan kernel and the 8-lane warp_kernel against the round-5 library
# (build/variants/libsvo_hip_r05.so): parity first, then per-kernel times of the full-track step, then the step itself.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06b; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
V=$PWD/build/variants
{
echo "== parity: tracking suite + full-size + reference-style"
timeout 1200 python -m pytest tests/test_tracking_gpu.py tests/test_full_size_gpu.py tests/test_reference_style_gpu.py tests/test_golden_track.py -q -m gpu -s 2>&1 | grep -E "update_seeds\[|passed|failed|Error|error" | tail -40
for v in r05 main r05 main; do
  lib="$V/libsvo_hip_$v.so"; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo "== per-kernel (rocprofv3 kernel trace, full-track step): $v"
  SVO_HIP_LIB="$lib" bash scripts/profile_full.sh "$O/prof_${v}_$RANDOM" 2>&1 | grep -v rocprim | head -12 | cut -c1-150
done
echo "== full track untraced"
bash scripts/full_variants.sh svo_hip_r05 main svo_hip_r05 main 2>&1 | cut -c1-330
} 2>&1 | tee $O/log.txt
