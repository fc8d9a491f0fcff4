#!/bin/bash
# Round 6, GPU call r06am: the full-track step with the depth filter on a stream of its own beside the next step's tracking
# (FullTrack.step_mapper_overlapped) next to the serial step, three processes.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06am; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
for rep in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --extras full --full-line --steps 5 --warmup 2 2> $O/err_$rep.txt | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); f=d['full_track']; print('serial', round(f['ms_per_step'],3), {k: round(v,3) for k,v in f['stages_ms'].items() if v>0.05}, 'overlapped', f.get('mapper_on_its_own_stream'))"
  tail -2 $O/err_$rep.txt | cut -c1-300
done
} 2>&1 | tee $O/log.txt
