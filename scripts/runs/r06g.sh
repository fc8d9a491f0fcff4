#!/bin/bash
# Round 6, seventh GPU call: the split K1 with the placement-independent exchange (sc1 write-through chunks, epoch in both
# halves, blocks zeroed per launch): microbenchmark with the XCC census, parity, configs[3] leg on/off; then the whole GPU suite.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06g; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== cross-workgroup exchange microbenchmark"
timeout 300 build/xwg_exchange_ubench 200 | tee $O/xwg_exchange_ubench.json
echo "== parity: K1 suites"
timeout 900 python -m pytest tests/test_full_size_gpu.py tests/test_sparse_align_gpu.py -q -m gpu 2>&1 | tail -5
for sp in 1 0 1 0; do
  echo "== configs[3] leg, SVO_HIP_K1_SPLIT=$sp"
  SVO_HIP_K1_SPLIT=$sp timeout 300 python bench.py --no-cpu-baseline --extras config3 --full-line --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); c=d['config3_xga5_b64']
print({k: c[k] for k in ('ms_per_step','frames_per_s','mean_gn_iterations_per_frame','median_pose_error_vs_gt','frames_per_s_at_batch_1024')}, c.get('split4_latency_floor'))"
done
echo "== the whole GPU suite"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
} 2>&1 | tee $O/log.txt
