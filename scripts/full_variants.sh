#!/bin/bash
# times the full-track leg (BASELINE configs[2]) with each library variant named on the command line
# ("main" = rpg_svo_amd/lib/libsvo_hip.so)
for v in "$@"; do
  lib=$PWD/build/variants/lib$v.so; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo -n "variant $v: "
  SVO_HIP_LIB=$lib python bench.py --no-cpu-baseline --extras full --full-line --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); f=d['full_track']
print('step_ms', round(f['ms_per_step'],3), {k: round(v,3) for k,v in f['stages_ms'].items()}, 'matches', f.get('matches_per_frame'), 'seeds', f.get('seed_status_per_frame'))"
done
