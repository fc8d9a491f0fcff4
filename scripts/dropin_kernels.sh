#!/bin/bash
# rocprofv3 kernel trace of the single-stream drop-in sequence alone (scripts/dropin_trace.py, hip flavour, 600 frames):
# per-kernel call counts and average durations -> <out>/kernels.txt
cd "$(dirname "$0")/.."; R=$PWD; export TMPDIR=/tmp; out=${1:-gpurun_out/dropin_kernels}
rm -rf $out; mkdir -p $out
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out -o trace -- \
   python $R/scripts/dropin_trace.py frames=600 > $R/$out/run.log 2>&1)
python - "$out" <<'PY' | tee $out/kernels.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'at::native' not in r['Name'] and 'Cijk' not in r['Name']]
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:24]:
    print(f"{r['Name'].replace('(anonymous namespace)::','')[:70]:70s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
grep "tot_time" $out/run.log
find $out -type f ! -name 'kernels.txt' ! -name 'run.log' -delete
