#!/bin/bash
# Round evidence, run on the GPU box: default bench line, rocprofv3 kernel-trace stats of the headline and of the
# full-track step, single-stream drop-in kernel stats, the VALU issue microbenchmark.  Summaries land in
# gpurun_out/<tag>/ and are copied from there into profiles/ (prefix <tag>_).
# usage: scripts/final_evidence.sh <tag>
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
stats() {  # <dir> <out.csv>: our kernels' rows of the kernel_stats table
  python - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'at::native' not in r['Name'] and 'Cijk' not in r['Name'] and 'rocprim' not in r['Name'] and 'rocclr' not in r['Name']]
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
w = csv.writer(open(sys.argv[2], 'w'))
w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs'])
for r in rows: w.writerow([r['Name'], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r.get('MinNs', ''), r.get('MaxNs', '')])
PY
}
echo "== the driver's command (last stdout line = the compact summary; the full object is bench_details.json)"
(cd $R && (time python3 bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_driver_style_line.json 2> $O/bench_driver_style.err; cp bench_details.json $O/bench_driver_style_details.json)
echo "== default bench"; (cd $R && (time python bench.py) > $O/bench_default_line.json 2> $O/bench_default.err; cp bench_details.json $O/bench_default_details.json)
echo "== headline under kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_align -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --extras none > $O/align_bench_under_trace.json 2> $O/trace_align.err
stats $O/trace_align $O/align_kernel_stats.csv
python $R/scripts/kernel_last_steps.py $O/trace_align 50 > $O/align_kernel_last_steps.txt  # every launch: warm-ups first
echo "== full track under kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o trace -- python $R/bench.py --pipeline full --steps 5 --warmup 2 --no-cpu-baseline --extras none > $O/full_bench_under_trace.json 2> $O/trace_full.err
stats $O/trace_full $O/full_kernel_stats.csv
# the set-up of the representative workload launches the same kernels at other sizes: the timed steps are the last dispatches
python $R/scripts/kernel_last_steps.py $O/trace_full 10 > $O/full_kernel_last_steps.txt
echo "== drop-in sequence under kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dropin -o trace -- python -c "import sys; sys.path.insert(0, '$R'); import bench, json; print(json.dumps(bench.dropin_sequence(600)))" > $O/dropin_under_trace.json 2> $O/trace_dropin.err
stats $O/trace_dropin $O/dropin_kernel_stats.csv
echo "== drop-in frame timeline (HIP API + kernels + copies of a median frame)"
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_timeline -- python $R/scripts/dropin_trace.py frames=600 > $O/dropin_traced_run.txt 2> $O/trace_timeline.err
python $R/scripts/dropin_trace.py --report $O/trace_timeline frames=600 > $O/dropin_frame_timeline_600.txt
python $R/scripts/dropin_trace.py > $O/dropin_untraced_sync.txt 2>/dev/null
python $R/scripts/dropin_trace.py defer > $O/dropin_untraced_deferred.txt 2>/dev/null
python $R/scripts/dropin_trace.py frames=600 > $O/dropin_untraced_sync_600.txt 2>/dev/null
python $R/scripts/dropin_trace.py frames=600 defer > $O/dropin_untraced_deferred_600.txt 2>/dev/null
rm -rf $O/trace_timeline
echo "== valu microbenchmark"; [ -x $R/build/valu_ubench ] && $R/build/valu_ubench > $O/valu_ubench.json
find $O -name "*.csv" -size +300k -delete; find $O -name "*.db" -delete
ls -la $O
