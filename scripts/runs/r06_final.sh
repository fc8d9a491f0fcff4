#!/bin/bash
# Round 6, the round's evidence in one call: the whole GPU suite, the driver's command (line + details), rocprofv3
# --kernel-trace --stats of the headline step and of the full-track step (same commands, counters off), smoke().
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r06z; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
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
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -8
echo "== smoke"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== the driver's command"
(time python3 bench.py --gpus 1 --steps 20 --warmup 5) > $O/bench_driver_style_line.json 2> $O/bench_driver_style.err; cp bench_details.json $O/bench_driver_style_details.json
tail -c 6000 $O/bench_driver_style_line.json; echo; tail -4 $O/bench_driver_style.err
echo "== headline step under kernel trace (the driver's steps / warm-up)"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_align -o trace -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --extras none > $O/headline_bench_under_trace.json 2> $O/trace_align.err)
stats $O/trace_align $O/headline_kernel_stats.csv
python scripts/kernel_last_steps.py $O/trace_align 20 > $O/headline_kernel_last_steps.txt
head -4 $O/headline_kernel_stats.csv | cut -c1-200
echo "== full track under kernel trace"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full -o trace -- python $R/bench.py --pipeline full --steps 5 --warmup 2 --no-cpu-baseline --extras none > $O/full_bench_under_trace.json 2> $O/trace_full.err)
stats $O/trace_full $O/full_kernel_stats.csv
python scripts/kernel_last_steps.py $O/trace_full 10 > $O/full_kernel_last_steps.txt
head -14 $O/full_kernel_stats.csv | cut -c1-160
rm -rf $O/trace_align $O/trace_full
} 2>&1 | tee $O/log.txt
