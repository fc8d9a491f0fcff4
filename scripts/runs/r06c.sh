#!/bin/bash
# Round 6, third GPU call: parity of the tree (fused warp + scan, seed_finish as the alignment's epilogue, alignment without
# the gradient cache at three waves per SIMD), then the full-track step against four whole-library variants:
#   r05      the round-5 library          oldalign  the tree with round 5's alignment (gradient cache, two waves)
#   nofin    the tree with seed_finish_kernel as a launch of its own      box296  the tree with the scan's LDS stride 296 dwords
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06c; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
V=$PWD/build/variants
{
echo "== parity: tracking suite + full-size + reference-style + golden + dropin store"
timeout 1500 python -m pytest tests/test_tracking_gpu.py tests/test_full_size_gpu.py tests/test_reference_style_gpu.py tests/test_golden_track.py tests/test_replay_gpu.py -q -m gpu 2>&1 | tail -5
echo "== full track untraced (two rounds)"
bash scripts/full_variants.sh svo_hip_r05 main svo_hip_oldalign svo_hip_nofin svo_hip_box296 svo_hip_r05 main svo_hip_oldalign svo_hip_nofin svo_hip_box296 2>&1 | cut -c1-260
for v in main oldalign; do
  lib="$V/libsvo_hip_$v.so"; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo "== per-kernel (rocprofv3 kernel trace, full-track step): $v"
  SVO_HIP_LIB="$lib" bash scripts/profile_full.sh "$O/prof_${v}_$RANDOM" 2>&1 | grep -v rocprim | head -12 | cut -c1-150
done
} 2>&1 | tee $O/log.txt
