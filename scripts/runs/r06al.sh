#!/bin/bash
# Round 6, GPU call r06al: K4's radix selection comparing its 64-bit keys as doubles (v_cmp_lt_f64 for v_cmp_lt_u64;
# -DRADIX_F64_COMPARE: build/variants/libk4f64.so) against the tree: pose tests on the variant, headline in alternating processes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06al; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== pose tests on k4f64"
SVO_HIP_LIB=$PWD/build/variants/libk4f64.so timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_golden_track.py tests/test_reference_style_gpu.py -q -m gpu -rf -k "pose or golden or reference" 2>&1 | tail -3
echo "== headline, alternating processes"
for rep in 1 2 3 4; do for v in main k4f64; do
  lib=$PWD/build/variants/lib$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo -n "$v: "; SVO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --extras none 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('stages_ms'))"
done; done
} 2>&1 | tee $O/log.txt
