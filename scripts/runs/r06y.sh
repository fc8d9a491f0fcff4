#!/bin/bash
# Round 6, twenty-seventh GPU call: seed_finish as the epilogue of the wave-per-trial alignment kernel (a camera frame's
# seeds: one launch less) against the tree before (build/variants/libnofinwave.so), alternating processes; parity suites.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06y; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== parity: tracking + full size + reference style + golden + replay"
timeout 1500 python -m pytest tests/test_tracking_gpu.py tests/test_full_size_gpu.py tests/test_reference_style_gpu.py tests/test_golden_track.py tests/test_replay_gpu.py -q -m gpu 2>&1 | tail -3
for rep in 1 2 3 4 5; do for v in nofinwave main; do
  lib=$PWD/build/variants/lib$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo -n "$v: "; SVO_HIP_LIB=$lib timeout 300 python -c "
import sys, json; sys.path.insert(0, '$R'); import bench; print(json.dumps(bench.dropin_hip_only(600, '')))" 2>/dev/null | tail -1 | cut -c1-60
done; done
echo "== drop-in GPU tests (seed store, chain)"
timeout 1200 python -m pytest tests/test_dropin_pipeline.py -q -m gpu -x -k "seed_store or chain or trajectory_matches" 2>&1 | tail -3
} 2>&1 | tee $O/log.txt
