#!/bin/bash
# Round 6, twenty-eighth GPU call: is the two-mode spread of the single-stream frame between processes (0.255 / 0.270 ms, the
# reprojection 83 / 90 us) a matter of WHERE the process runs?  The drop-in pinned to the GPU's own NUMA node, to another
# node, and unpinned.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"; O=gpurun_out/r06aa; rm -rf "$O"; mkdir -p "$O"; export TMPDIR=/tmp
{
lscpu | grep -i "numa\|socket\|model name" | head -12
for d in /sys/class/drm/card*/device; do
  if [ -f $d/vendor ] && grep -q 0x1002 $d/vendor; then echo "$d: $(readlink -f $d | xargs basename) numa_node $(cat $d/numa_node) local_cpulist $(cat $d/local_cpulist)"; fi
done
LOCAL=$(for d in /sys/class/drm/card*/device; do if grep -q 0x1002 $d/vendor 2>/dev/null; then cat $d/local_cpulist; break; fi; done)
NODE=$(for d in /sys/class/drm/card*/device; do if grep -q 0x1002 $d/vendor 2>/dev/null; then cat $d/numa_node; break; fi; done)
echo "local cpus: $LOCAL (node $NODE)"
OTHER=$(for n in /sys/devices/system/node/node*; do k=$(basename $n | sed s/node//); if [ "$k" != "$NODE" ]; then cat $n/cpulist; break; fi; done)
echo "other cpus: $OTHER"
run() { timeout 300 "$@" python -c "
import sys, json, os; sys.path.insert(0, '$R'); import bench; r = bench.dropin_hip_only(600, ''); print(json.dumps({k: r[k] for k in ('tot_time', 'reproject')}), 'cpu', os.sched_getcpu())" 2>/dev/null | tail -1; }
for rep in 1 2 3 4; do
  echo -n "unpinned: "; run env
  if [ -n "$LOCAL" ]; then echo -n "local:    "; run taskset -c "$LOCAL"; fi
  if [ -n "$OTHER" ]; then echo -n "other:    "; run taskset -c "$OTHER"; fi
done
} 2>&1 | tee $O/log.txt
