#!/bin/bash
# Single-stream latency of the drop-in pipeline with Reprojector::reprojectMap on the device-resident map mirror
# (SVO_HIP_MAP_MIRROR=on, the default) and on the list-walking path (=off); synchronous and deferred mapper, two runs
# each.  Run on the GPU box: scripts/mirror_modes.sh [frames=600]
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for mode in on off; do
    for defer in "" defer; do
      echo -n "SVO_HIP_MAP_MIRROR=$mode ${defer:-sync-mapper} run $rep: "
      SVO_HIP_MAP_MIRROR=$mode timeout 300 python scripts/dropin_trace.py $defer "$@" 2>/dev/null | head -1
      [ $rep = 1 ] && [ -z "$defer" ] && SVO_HIP_MAP_MIRROR=$mode timeout 300 python scripts/dropin_trace.py "$@" 2>/dev/null | tail -4
    done
  done
done
