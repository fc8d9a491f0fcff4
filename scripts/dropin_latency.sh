#!/bin/bash
# Single-stream latency of the drop-in pipeline under both arena modes (svo_hip::Arena): prints the
# dropin_sequence leg of bench.py once per mode.  Run on the GPU box.
cd "$(dirname "$0")/.."
for mode in mirrored mapped; do
  echo "== SVO_HIP_ARENA=$mode"
  SVO_HIP_ARENA=$mode timeout 300 python -c "
import json, bench
print(json.dumps(bench.dropin_sequence(120)))"
done
