#!/bin/bash
# The host-emulated kernels (tests/emu_build.py) under a sanitizer: every load and store of the kernels checked against
# the bounds of the buffers they were handed and of their LDS arrays (address), or against the barriers and hand-overs
# that order them between work-items (thread).
#   scripts/emu_sanitize.sh address|thread|undefined [pytest arguments; default: the emulated test files]
set -e
kind=${1:-address}; shift || true
case $kind in address) name=asan; opts="ASAN_OPTIONS=detect_leaks=0";; thread) name=tsan; opts="TSAN_OPTIONS=report_signal_unsafe=0:suppressions=$(cd "$(dirname "$0")/.." && pwd)/tests/host/tsan.supp${SVO_TSAN_LOG:+:log_path=$SVO_TSAN_LOG} OMP_NUM_THREADS=1";;
  undefined) name=ubsan_standalone; opts="UBSAN_OPTIONS=print_stacktrace=1${SVO_UBSAN_LOG:+:log_path=$SVO_UBSAN_LOG}";;
  *) echo "usage: $0 address|thread|undefined [pytest args]"; exit 2;; esac
rt=$(${ROCM_PATH:-/opt/rocm}/lib/llvm/bin/clang++ -print-file-name=libclang_rt.$name-x86_64.so)
[ -f "$rt" ] || { echo "no $name runtime next to ROCm's clang++"; exit 3; }
cd "$(dirname "$0")/.."
[ $# -gt 0 ] || set -- tests/test_sparse_align_emulated.py tests/test_track_emulated.py tests/test_optimizers_emulated.py tests/test_entries_emulated.py tests/test_map_mirror_emulated.py tests/test_fast_emulated.py tests/test_pyramid_emulated.py tests/test_emulated_shapes.py -q
env LD_PRELOAD=$rt $opts SVO_EMU_SANITIZE=$kind python -m pytest "$@"
