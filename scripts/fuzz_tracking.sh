#!/bin/bash
# The bit-exact parity suites of the matcher / feature alignment / depth filter on scenes other than the committed one:
# SVO_TEST_FUZZ=k moves the scene's seed and every random draw of the test files.
#   scripts/fuzz_tracking.sh emu 1 30   -- tests/test_track_emulated.py, test_optimizers_emulated.py (the kernels through the CPU emulation, no GPU)
#   scripts/fuzz_tracking.sh gpu 1 12   -- tests/test_tracking_gpu.py, test_sparse_align_gpu.py on the GPU box (both checkers, three cameras)
# One line per k: what pytest's summary says; the failing assert, if any, underneath.
set -u
cd "$(dirname "$0")/.."
mode=$1; lo=$2; hi=$3
for k in $(seq $lo $hi); do
  if [ "$mode" == "emu" ]; then
    out=$(SVO_TEST_FUZZ=$k timeout 900 python -m pytest tests/test_track_emulated.py tests/test_optimizers_emulated.py -q -rf 2>&1)
  else
    out=$(SVO_TEST_FUZZ=$k timeout 900 python -m pytest tests/test_tracking_gpu.py tests/test_sparse_align_gpu.py -q -m gpu -rf 2>&1)
  fi
  echo "fuzz $k: $(echo "$out" | tail -1)"
  echo "$out" | grep -E "^(E  |FAILED|ERROR)" | cut -c1-220 | head -24
done
