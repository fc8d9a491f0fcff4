#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + two PMC passes of bench.py.
# usage: scripts/profile_gpu.sh <tag> [bench args...]
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and summaries that are
# meant to be copied into profiles/ under gpurun_out/profiles_<tag>/.
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
SUM=$REPO/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$SUM"
BENCH_ARGS="--steps 10 --warmup 2 --no-cpu-baseline $*"
# counters only for this repository's kernels (the data-generation GEMMs of torch are not of interest
# and thousands of instrumented dispatches have crashed the tool)
KRE="sia_kernel|pyramid_fused|half_sample|load_level0|warp_kernel|align_kernel|pose_opt|epi_scan|seed_|match_prepare|fast_"
cd /tmp && export TMPDIR=/tmp
echo "== kernel trace" 
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python "$REPO/bench.py" $BENCH_ARGS > "$OUT/trace.log" 2>&1
grep "^{\"metric\"" "$OUT/trace.log" | tail -1 > "$SUM/${TAG}_bench_under_trace.json"
echo "== pmc FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$KRE" --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python "$REPO/bench.py" $BENCH_ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "== pmc WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "$KRE" --output-format csv -d "$OUT/pmc_write" -o write -- python "$REPO/bench.py" $BENCH_ARGS > "$OUT/pmc_write.log" 2>&1
echo "== pmc SQ"
rocprofv3 --kernel-include-regex "$KRE" --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -o sq -- python "$REPO/bench.py" $BENCH_ARGS > "$OUT/pmc_sq.log" 2>&1
python "$REPO/scripts/summarize_profile.py" "$OUT" "$SUM" "$TAG"
ls -la "$SUM"
# raw traces are large (gpurun_out is capped at 64 MiB): keep only logs + summaries
find "$OUT" -name "*.csv" -size +200k -delete
