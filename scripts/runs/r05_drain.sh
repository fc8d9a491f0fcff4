#!/bin/bash
# Round 5, first GPU call: the ten queued compile-time variants, measured per KERNEL.  Each flag of the queue touches
# one kernel, so one rocprofv3 kernel trace of the full-track step with every flag on (libsvo_hip_queue.so) against one
# of the default library times them all; the flags that cannot share a library (ALIGN_G_F16, TAU_ALGEBRAIC,
# SCAN_PREFETCH at four waves) get a trace each.  Parity first: the GPU suites on the combined library.
#   bash scripts/round5_queue.sh build     (CPU side)
#   gpurun --timeout 1200 -- 'bash scripts/r05_drain.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
V=$PWD/build/variants
O=gpurun_out/r05a
mkdir -p $O
export TMPDIR=/tmp
{
echo "== parity: GPU suites on the combined queue library"
SVO_HIP_LIB=$V/libsvo_hip_queue.so timeout 600 python -m pytest tests/test_sparse_align_gpu.py tests/test_tracking_gpu.py tests/test_map_mirror_gpu.py tests/test_full_size_gpu.py -q -m gpu 2>&1 | tail -5
for v in ALIGN_G_F16 TAU_ALGEBRAIC "SCAN_PREFETCH_SCAN_MINW=4"; do
  echo "== parity: tracking suite on $v"
  SVO_HIP_LIB="$V/libsvo_hip_$v.so" timeout 400 python -m pytest tests/test_tracking_gpu.py -q -m gpu 2>&1 | tail -3
done
for v in main queue ALIGN_G_F16 TAU_ALGEBRAIC "SCAN_PREFETCH_SCAN_MINW=4" main queue; do
  lib="$V/libsvo_hip_$v.so"; [ "$v" == "main" ] && lib=$PWD/rpg_svo_amd/lib/libsvo_hip.so
  echo "== per-kernel (rocprofv3 kernel trace, full-track step): $v"
  SVO_HIP_LIB="$lib" bash scripts/profile_full.sh "$O/prof_${v}_$RANDOM" 2>&1 | cut -c1-150
done
echo "== K1 headline, alternating"
bash scripts/k1_variants.sh main svo_hip_queue main svo_hip_queue -- --steps 40 --warmup 15
echo "== full track untraced, alternating"
bash scripts/full_variants.sh main svo_hip_queue main svo_hip_queue 2>&1 | cut -c1-260
} 2>&1 | tee $O/drain.txt
