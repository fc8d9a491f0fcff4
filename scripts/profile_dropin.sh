#!/bin/bash
# rocprofv3 kernel trace of the single-stream drop-in sequence (bench.dropin_sequence): per-kernel averages
cd "$(dirname "$0")/.."; R=$PWD; export TMPDIR=/tmp; out=${1:-gpurun_out/prof_dropin}
rm -rf $out; mkdir -p $out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o trace -- \
   python -c "import sys; sys.path.insert(0, '$R'); import bench, json; print(json.dumps(bench.dropin_sequence(120)))" > $R/$out/run.log 2>&1)
python - "$out" <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'at::native' not in r['Name'] and 'Cijk' not in r['Name']]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:18]:
    print(f"{r['Name'][:64]:64s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
