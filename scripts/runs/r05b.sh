#!/bin/bash
# Round 5, second GPU call: the tree after the queue drain (winners flipped, K1 rows / partials in f64).
#   1. the whole GPU suite (new: search-step-cap tests, ref checker on the full-size configs)
#   2. K1 width A/B, alternating: default (f32 products, f64 from the patch upwards) against -DSIA_F32_ROWS (rounds 2-4)
#   3. the default bench run (line + details; its f64_partials leg carries the reference-width build's own roofline)
#   4. rocprofv3 --kernel-trace --stats of the headline: default build and -DSIA_F64_PARTIALS build
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=$R/gpurun_out/r05b; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
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
echo "== 1. GPU suite"
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15
echo "== 2. K1 width A/B (40 steps after 15 warm-ups, alternating)"
bash scripts/k1_variants.sh main svo_hip_SIA_F32_ROWS main svo_hip_SIA_F32_ROWS main svo_hip_SIA_F32_ROWS -- --steps 40 --warmup 15
echo "== 3. default bench"
(time python bench.py) > $O/bench_default_line.json 2> $O/bench_default.err; cp bench_details.json $O/bench_default_details.json
tail -c 1500 $O/bench_default_line.json; echo; tail -4 $O/bench_default.err
echo "== 4. headline under kernel trace: default, then the reference-width build"
for v in default f64; do
  lib=$R/rpg_svo_amd/lib/libsvo_hip.so; [ $v == f64 ] && lib=$R/rpg_svo_amd/lib/variants/libsvo_hip_SIA_F64_PARTIALS.so
  (cd /tmp && SVO_HIP_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$v -o trace -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --extras none --k1-kernel workgroup > $O/align_${v}_bench_under_trace.json 2> $O/trace_$v.err)
  stats $O/trace_$v $O/align_${v}_kernel_stats.csv
  python scripts/kernel_last_steps.py $O/trace_$v 50 > $O/align_${v}_kernel_last_steps.txt
  head -3 $O/align_${v}_kernel_stats.csv; tail -2 $O/align_${v}_kernel_last_steps.txt
  rm -rf $O/trace_$v
done
} 2>&1 | tee $O/log.txt
