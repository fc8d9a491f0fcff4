#!/bin/bash
# Round 6, thirtieth GPU call: every thread that gets a lane bound to the CPUs next to the GPU (svo_hip_pin_calling_thread,
# SVO_HIP_PIN_HOST): on / off in alternating processes, the new test, the drop-in GPU tests (mapper thread included).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06ab; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
for rep in 1 2 3 4 5; do for pin in 1 0; do
  echo -n "pin_host=$pin: "; SVO_HIP_PIN_HOST=$pin timeout 300 python -c "
import sys, json; sys.path.insert(0, '$R'); import bench; print(json.dumps(bench.dropin_hip_only(600, '')))" 2>/dev/null | tail -1 | cut -c1-60
done; done
for rep in 1 2 3; do for pin in 1 0; do
  echo -n "deferred mapper, pin_host=$pin: "; SVO_HIP_PIN_HOST=$pin timeout 300 python -c "
import sys, json; sys.path.insert(0, '$R'); import bench; print(json.dumps(bench.dropin_hip_only(600, '', defer_mapper=1)))" 2>/dev/null | tail -1 | cut -c1-60
done; done
echo "== tests"
timeout 1800 python -m pytest tests/test_capi_errors_gpu.py tests/test_dropin_pipeline.py tests/test_host_device_gpu.py -q -m gpu -x 2>&1 | tail -4
} 2>&1 | tee $O/log.txt
