#!/bin/bash
# Round 5, third GPU call: the 16-positions-per-pass epipolar scan against the scan of rounds 3-4 (build/variants/
# libsvo_hip_oldscan.so = the same tree at 88286ce), on one box: parity suites, per-kernel rocprofv3 tables of the
# full-track step alternating, the untraced step, and the scan's LDS / wait / traffic counters for both.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r05c; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
V=$PWD/build/variants
{
echo "== parity: tracking / full-size / replay suites on the new scan"
timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_full_size_gpu.py tests/test_replay_gpu.py tests/test_dropin_pipeline.py -q -m gpu 2>&1 | tail -6
for v in oldscan main oldscan main; do
  lib="$V/libsvo_hip_$v.so"; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo "== per-kernel (rocprofv3 kernel trace, full-track step): $v"
  SVO_HIP_LIB="$lib" bash scripts/profile_full.sh "$O/prof_${v}_$RANDOM" 2>&1 | grep -v rocprim | head -9 | cut -c1-150
done
echo "== full track untraced, alternating"
bash scripts/full_variants.sh svo_hip_oldscan main svo_hip_oldscan main 2>&1 | cut -c1-230
echo "== counters of epi_scan_kernel (rocprofv3 --pmc, one group per pass), per launch"
for v in oldscan main; do
  lib="$V/libsvo_hip_$v.so"; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM"; do
    d=/tmp/pmc_$RANDOM
    (cd /tmp && SVO_HIP_LIB="$lib" timeout 300 rocprofv3 --pmc $grp --kernel-include-regex epi_scan --output-format csv -d $d -o p -- python $R/bench.py --pipeline full --extras none --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2>&1)
    python - "$d" "$v" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(list)
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'epi_scan' in r.get('Kernel_Name', ''): acc[r['Counter_Name']].append(float(r['Counter_Value']))
# the timed steps are the last dispatches (set-up launches of the representative workload come first): last 6
print(sys.argv[2], {k: round(sum(v[-6:]) / max(1, len(v[-6:])), 1) for k, v in acc.items()}, 'launches', {k: len(v) for k, v in acc.items()})
PY
    rm -rf $d
  done
done
} 2>&1 | tee $O/log.txt
