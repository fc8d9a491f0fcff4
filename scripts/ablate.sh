#!/bin/bash
# timing ablation of the sparse-align kernel (debug variants under build/variants: scripts/build_variants.sh)
for v in A B C D; do
  echo -n "variant $v: "
  SVO_HIP_LIB=$PWD/build/variants/lib$v.so python bench.py --no-cpu-baseline --extras none --n-iter 4 --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('kernel_ms', round(d['roofline']['kernel_ms_avg'],4), 'iters/frame', d['config']['mean_gn_iterations_per_frame'])"
done
for B in 1024 2048 4096 8192 16384; do
  echo -n "batch $B: "
  python bench.py --no-cpu-baseline --extras none --batch $B --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fps', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms_avg'],4))"
done
