#!/bin/bash
# Round 5, seventh GPU call: the scan (runtime one / two positions per lane, no spills); K4 at two waves per SIMD against
# three (build/variants/libsvo_hip_pw2.so); the cross-workgroup exchange microbenchmark (configs[3] split).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r05g; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
V=$PWD/build/variants
{
echo "== parity: tracking suite"
timeout 600 python -m pytest tests/test_tracking_gpu.py -q -m gpu 2>&1 | tail -3
echo "== cross-workgroup exchange microbenchmark"
timeout 120 build/xwg_exchange_ubench 200 | tee $O/xwg_exchange_ubench.json
for v in main pw2 main pw2; do
  lib="$V/libsvo_hip_$v.so"; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo "== per-kernel (rocprofv3 kernel trace, full-track step): $v"
  SVO_HIP_LIB="$lib" bash scripts/profile_full.sh "$O/prof_${v}_$RANDOM" 2>&1 | grep -v rocprim | head -9 | cut -c1-150
done
echo "== full track untraced"
bash scripts/full_variants.sh main svo_hip_pw2 main svo_hip_pw2 2>&1 | cut -c1-230
echo "== counters of epi_scan_kernel"
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CU_CYCLES"; do
  d=/tmp/pmc_$RANDOM
  (cd /tmp && timeout 300 rocprofv3 --pmc $grp --kernel-include-regex epi_scan --output-format csv -d $d -o p -- python $R/bench.py --pipeline full --extras none --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1)
  python - "$d" main <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'epi_scan' in r.get('Kernel_Name', ''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
print(sys.argv[2], {k: round(sum(v[-6:]) / max(1, len(v[-6:])), 1) for k, v in acc.items()})
PY
  rm -rf $d
done
} 2>&1 | tee $O/log.txt
