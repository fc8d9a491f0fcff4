#!/bin/bash
# Round 5, the round's evidence in one call: the whole GPU suite, the default bench run (line + details), rocprofv3
# --kernel-trace --stats of the headline and of the full-track step, the single-stream drop-in untraced.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r05z; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
stats() {
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
{
echo "== GPU suite"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -6
echo "== default bench"
(time python bench.py) > $O/bench_default_line.json 2> $O/bench_default.err; cp bench_details.json $O/bench_default_details.json
tail -c 800 $O/bench_default_line.json; echo; tail -4 $O/bench_default.err
echo "== headline under kernel trace"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_align -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --extras none > $O/align_bench_under_trace.json 2> $O/trace_align.err)
stats $O/trace_align $O/align_kernel_stats.csv
python scripts/kernel_last_steps.py $O/trace_align 50 > $O/align_kernel_last_steps.txt
head -3 $O/align_kernel_stats.csv
echo "== full track under kernel trace"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o trace -- python $R/bench.py --pipeline full --steps 5 --warmup 2 --no-cpu-baseline --extras none > $O/full_bench_under_trace.json 2> $O/trace_full.err)
stats $O/trace_full $O/full_kernel_stats.csv
python scripts/kernel_last_steps.py $O/trace_full 10 > $O/full_kernel_last_steps.txt
head -12 $O/full_kernel_stats.csv | cut -c1-160
rm -rf $O/trace_align $O/trace_full
echo "== single-stream drop-in, untraced (600 frames): synchronous mapper"
for k in 1 2; do python scripts/dropin_trace.py frames=600 2>/dev/null | grep "tot_time median"; done
} 2>&1 | tee $O/log.txt
