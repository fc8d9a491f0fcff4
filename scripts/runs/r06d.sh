#!/bin/bash
# Round 6, fourth GPU call: the alignment's window cache (wc2: two waves per SIMD, no spill; wc3 = the tree: three waves,
# 20 spilled dwords) against the committed tree (nowc); the XCD-local exchange microbenchmark.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06d; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
V=$PWD/build/variants
{
echo "== cross-workgroup exchange microbenchmark"
timeout 120 build/xwg_exchange_ubench 200 | tee $O/xwg_exchange_ubench.json
echo "== parity: tracking suite (the tree)"
timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_full_size_gpu.py -q -m gpu 2>&1 | tail -3
echo "== full track untraced (three rounds)"
bash scripts/full_variants.sh svo_hip_nowc svo_hip_wc2 svo_hip_wc3 svo_hip_nowc svo_hip_wc2 svo_hip_wc3 svo_hip_nowc svo_hip_wc2 svo_hip_wc3 2>&1 | cut -c1-260
} 2>&1 | tee $O/log.txt
