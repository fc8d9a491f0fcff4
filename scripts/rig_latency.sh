# small-batch (camera-rig) latency: launch-bound chain with and without a HIP graph
R=${GRAFT_REPO_ROOT:-.}
for wl in "svo_default_752_l4to2_n120"; do
for B in 1 8 64; do
  for g in "" "--graph"; do
    for pl in align full; do
      echo -n "$wl B=$B $pl ${g:-nograph}: "
      timeout 200 python $R/bench.py --workload $wl --pipeline $pl --batch $B --steps 300 --warmup 20 --no-cpu-baseline $g 2>&1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), 'frames/s', round(d['ms_per_step']*1e3,1), 'us/step')
except Exception as e: print('FAILED', e)"
    done
  done
done
done
