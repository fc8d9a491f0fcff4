#!/bin/bash
# Round 6, sixteenth GPU call: the depth filter's update enqueued by the pose optimizer's drop-in (SVO_HIP_EARLY_MAPPER) on /
# off, each in a process of its own, alternating; the drop-in GPU tests that cover it; the frame's timeline; then the
# driver's command on the tree as it stands.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06p; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
for rep in 1 2 3; do
  for em in 1 0; do
    echo -n "early_mapper=$em: "
    SVO_HIP_EARLY_MAPPER=$em timeout 300 python -c "
import sys, json; sys.path.insert(0, '$R'); import bench; print(json.dumps(bench.dropin_hip_only(600, '')))" 2>/dev/null | tail -1 | cut -c1-200
  done
done
echo "== drop-in GPU tests (seed store, frame chain, deferred mapper, mapper thread)"
timeout 1500 python -m pytest tests/test_dropin_pipeline.py tests/test_replay_gpu.py -q -m gpu -x -s 2>&1 | grep -v "INFO\|^$" | tail -14
echo "== the frame's timeline"
(cd /tmp && timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $R/$O/trace_timeline -- python $R/scripts/dropin_trace.py frames=600 > $R/$O/dropin_traced_run.txt 2> $R/$O/trace_timeline.err)
python scripts/dropin_trace.py --report $O/trace_timeline frames=600 > $O/dropin_frame_timeline_600.txt
head -90 $O/dropin_frame_timeline_600.txt | cut -c1-140
rm -rf $O/trace_timeline
echo "== the driver's command"
(time python3 bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_driver_style_line.json 2> $O/bench_driver_style.err; cp bench_details.json $O/bench_driver_style_details.json
tail -c 6000 $O/bench_driver_style_line.json; echo; tail -4 $O/bench_driver_style.err
} 2>&1 | tee $O/log.txt
