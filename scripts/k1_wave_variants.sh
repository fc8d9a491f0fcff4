#!/bin/bash
# wave-per-frame kernel variants (build/variants/lib<name>.so, SIAW_MAX_PATCHES=256) at several patch counts
cd "$(dirname "$0")/.."
for n in 120 192 200; do for v in "$@"; do
  lib=$PWD/build/variants/lib$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo -n "patches $n variant $v: "
  SVO_HIP_LIB=$lib SVO_BENCH_PATCHES=$n python bench.py --no-cpu-baseline --extras none --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fps', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'iters/frame', round(d['config']['mean_gn_iterations_per_frame'],3), 'err', d['config']['median_pose_error_vs_gt'])"
done; done
