#!/bin/bash
# Round 6, GPU call r06ah: K1 with v_rcp_f64 + two Newton steps for the IEEE divisions of the Gauss-Jordan inverse (six in a
# row per H rebuild) and of a patch's normalised coordinates (-DSIA_FAST_RCP: build/variants/libk1rcp.so) against the tree:
# the sparse-alignment suites on the variant, the headline step in alternating processes (with its parity leg against the
# reference's translation unit).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06ah; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
echo "== parity on k1rcp"
SVO_HIP_LIB=$PWD/build/variants/libk1rcp.so timeout 900 python -m pytest tests/test_sparse_align_gpu.py tests/test_full_size_gpu.py tests/test_reference_style_gpu.py -q -m gpu -rf 2>&1 | tail -4
echo "== headline, alternating processes"
for rep in 1 2 3 4; do for v in main k1rcp; do
  lib=$PWD/build/variants/lib$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo -n "$v: "; SVO_HIP_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --extras none 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('stages_ms'), d['roofline'].get('ms_last_10_launches'))"
done; done
echo "== parity leg of the bench on k1rcp (8192 frames against the reference's translation unit)"
SVO_HIP_LIB=$PWD/build/variants/libk1rcp.so timeout 900 python bench.py --extras none 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print(d.get('parity'))"
} 2>&1 | tee $O/log.txt
