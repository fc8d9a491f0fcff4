#!/bin/bash
# Round 6, GPU call r06ae: SQ_LDS_UNALIGNED_STALL per kernel (own kernels only under the counters), the tracking suite as
# committed (K4 with one selection) and on twelve other scenes (ill-posed pose frames and the sigma2 bound as the first
# fuzz run found them), the headline step twice.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06ae; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== LDS alignment stalls"
timeout 900 python scripts/lds_unaligned.py $O/lds_unaligned.json 2>&1 | tail -60
echo "== tracking suite as committed"
timeout 900 python -m pytest tests/test_tracking_gpu.py -q -m gpu 2>&1 | tail -3
echo "== tracking suite, other scenes"
bash scripts/fuzz_tracking.sh gpu 1 12
echo "== headline"
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --extras none 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('stages_ms'))"; done
} 2>&1 | tee $O/log.txt
