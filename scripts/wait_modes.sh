#!/bin/bash
# Single-stream latency of the drop-in pipeline with the host waiting by hipStreamSynchronize (SVO_HIP_WAIT=sync, the
# default) and by polling a word a stream write-value command stores (SVO_HIP_WAIT=signal), synchronous and deferred
# mapper, three runs each (boxes and runs differ by a few us).  Run on the GPU box: scripts/wait_modes.sh > gpurun_out/wait_modes.txt
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for mode in sync signal; do
    for defer in "" defer; do
      echo -n "SVO_HIP_WAIT=$mode ${defer:-sync-mapper} run $rep: "
      SVO_HIP_WAIT=$mode timeout 300 python scripts/dropin_trace.py $defer "$@" 2>/dev/null | head -1
    done
  done
done
