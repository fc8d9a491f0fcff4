#!/bin/bash
# The short form of final_evidence.sh (when GPU minutes are short): default bench line + details (with the counter
# passes), rocprofv3 kernel-trace stats of the headline and of the full-track step.  Summaries land in
# gpurun_out/<tag>/ and are copied from there into profiles/ (prefix <tag>_).
# usage: scripts/evidence_lite.sh <tag>
set -u
TAG=${1:-r04u}
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
echo "== default bench"; (cd $R && (time python bench.py) > $O/bench_default_line.json 2> $O/bench_default.err; cp bench_details.json $O/bench_default_details.json)
tail -c 600 $O/bench_default_line.json; echo
echo "== full track under kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o trace -- python $R/bench.py --pipeline full --steps 5 --warmup 2 --no-cpu-baseline --extras none > $O/full_bench_under_trace.json 2> $O/trace_full.err
stats $O/trace_full $O/full_kernel_stats.csv
python $R/scripts/kernel_last_steps.py $O/trace_full 10 > $O/full_kernel_last_steps.txt
echo "== headline under kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_align -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --extras none > $O/align_bench_under_trace.json 2> $O/trace_align.err
stats $O/trace_align $O/align_kernel_stats.csv
python $R/scripts/kernel_last_steps.py $O/trace_align 50 > $O/align_kernel_last_steps.txt
find $O -name "*.csv" -size +300k -delete; find $O -name "*.db" -delete; rm -rf $O/trace_full $O/trace_align
ls -la $O
