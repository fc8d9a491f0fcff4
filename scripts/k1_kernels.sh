#!/bin/bash
# K1 on the headline workload with both kernels (one wave per frame / one workgroup per frame)
cd "$(dirname "$0")/.."
for k in auto workgroup auto; do
  echo -n "k1-kernel $k: "
  python bench.py --no-cpu-baseline --extras none --steps 10 --warmup 2 --k1-kernel $k "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fps', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'iters/frame', round(d['config']['mean_gn_iterations_per_frame'],3), 'err_vs_gt', d['config']['median_pose_error_vs_gt'])"
done
