#!/bin/bash
# Round 5: the single-stream drop-in frame, API call by API call (rocprofv3 --hip-trace --kernel-trace --memory-copy-trace of
# scripts/dropin_trace.py; no counters in this run), and the same untraced with the deferred mapper.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r05x; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
(cd /tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_timeline -- python $R/scripts/dropin_trace.py frames=600 > $O/dropin_traced_run.txt 2> $O/trace_timeline.err)
python scripts/dropin_trace.py --report $O/trace_timeline frames=600 > $O/dropin_frame_timeline_600.txt
head -100 $O/dropin_frame_timeline_600.txt | cut -c1-160
rm -rf $O/trace_timeline
echo "== untraced: synchronous, then deferred mapper"
python scripts/dropin_trace.py frames=600 2>/dev/null | grep "tot_time median"
SVO_HIP_MAPPER=deferred python scripts/dropin_trace.py frames=600 2>/dev/null | grep "tot_time median"
} 2>&1 | tee $O/log.txt
