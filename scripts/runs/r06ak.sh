#!/bin/bash
# Round 6, GPU call r06ak: the drop-in against the all-CPU reference over six trajectories on the final tree (K4 and K1
# with this session's arithmetic), as profiles/r06w_* had it for the tree before.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06ak; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
timeout 900 python scripts/dropin_many.py frames=400 > $O/dropin_many.json 2> $O/dropin_many.err
python -c "
import json; d=json.load(open('$O/dropin_many.json')); r=d.pop('runs'); print(json.dumps(d)[:2500]); print([ (x['trajectory_seed'], x['first_frame_with_a_different_decision'], x['first_frame_with_a_different_tracking_decision']) for x in r])"
tail -3 $O/dropin_many.err
} 2>&1 | tee $O/log.txt
