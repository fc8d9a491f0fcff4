#!/bin/bash
# K1 kernel time against the patch count (kernel experiments: is the time proportional to the instruction count?)
cd "$(dirname "$0")/.."
for k in workgroup auto; do for n in ${SVO_PATCH_COUNTS:-128 192 200}; do
  echo -n "k1-kernel $k patches $n: "
  SVO_BENCH_PATCHES=$n python bench.py --no-cpu-baseline --extras none --steps 10 --warmup 2 --k1-kernel $k "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fps', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'iters/frame', round(d['config']['mean_gn_iterations_per_frame'],3))"
done; done
