#!/bin/bash
# rocprofv3 kernel trace of the full-track step (bench.py --pipeline full): per-kernel totals of our kernels
cd "$(dirname "$0")/.."; R=$PWD; export TMPDIR=/tmp; out=${1:-gpurun_out/prof_full}
rm -rf $out; mkdir -p $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o trace -- \
   python $R/bench.py --pipeline full --extras none --no-cpu-baseline --steps 5 --warmup 2 > $R/$out/bench.log 2>&1)
# keep the summaries only (the raw traces are ~18 MB per run; gpurun copies back at most 64 MiB)
find $out -type f ! -name '*kernel_stats.csv' ! -name 'bench.log' -delete
python - "$out" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'at::native' not in r['Name'] and 'Cijk' not in r['Name'] and 'rocclr' not in r['Name']]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:24]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):4d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
