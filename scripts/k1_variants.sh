#!/bin/bash
# times K1 (headline workload, no extra legs) with each library variant named on the command line
# ("main" = rpg_svo_amd/lib/libsvo_hip.so); extra bench arguments after "--"
names=(); extra=()
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; extra=("$@"); break; fi; names+=("$1"); shift; done
for v in "${names[@]}"; do
  lib=$PWD/build/variants/lib$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo -n "variant $v: "
  SVO_HIP_LIB=$lib python bench.py --no-cpu-baseline --extras none --full-line --steps 10 --warmup 2 "${extra[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); r=d.get('roofline',{}); print('fps', round(d['value']), 'kernel_ms', r.get('kernel_ms_avg'), 'last10_ms', r.get('ms_last_10_launches'), 'iters/frame', d.get('config',{}).get('mean_gn_iterations_per_frame'))"
done
