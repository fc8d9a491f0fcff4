#!/bin/bash
# Round 3, on the GPU box: tiled store against the row-major A/B build (python -m rpg_svo_amd.build -DSVO_PYR_ROWMAJOR)
# on the same box -- headline + full track + K0, and the per-kernel rocprofv3 stats of the full-track step for both.
# usage: scripts/r03_ab.sh <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r03b}; mkdir -p "$O"
export TMPDIR=/tmp
RM=$R/build/variants/libsvo_hip_SVO_PYR_ROWMAJOR.so
cd $R
python bench.py --extras f64,full,k0 > $O/bench_tiled.json 2> $O/bench_tiled.err
SVO_HIP_LIB=$RM python bench.py --extras full,k0 --no-cpu-baseline > $O/bench_rowmajor.json 2> $O/bench_rowmajor.err
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
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full_tiled -o trace -- python $R/bench.py --pipeline full --steps 5 --warmup 2 --no-cpu-baseline --extras none > $O/full_tiled_under_trace.json 2> $O/trace_full_tiled.err
stats $O/trace_full_tiled $O/full_tiled_kernel_stats.csv
SVO_HIP_LIB=$RM timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_full_rowmajor -o trace -- python $R/bench.py --pipeline full --steps 5 --warmup 2 --no-cpu-baseline --extras none > $O/full_rowmajor_under_trace.json 2> $O/trace_full_rowmajor.err
stats $O/trace_full_rowmajor $O/full_rowmajor_kernel_stats.csv
find $O -name "*.csv" -size +300k -delete; find $O -name "*.db" -delete; rm -rf $O/trace_full_tiled $O/trace_full_rowmajor
ls $O
