#!/bin/bash
# SQ counter passes over the headline K1 launch (rocprofv3 --pmc only, no trace domain): prints the per-launch
# average of every counter for the kernel matching sia_.  usage: scripts/k1_pmc.sh <k1-kernel> "<CTR CTR ...>" ["<CTR ...>" ...]
cd "$(dirname "$0")/.."; R=$PWD; export TMPDIR=/tmp
k=$1; shift
i=0
for set in "$@"; do
  i=$((i+1)); d=/tmp/k1pmc_$i; rm -rf $d
  (cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-include-regex "sia_" --output-format csv -d $d -o p -- \
     python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --extras none --pmc-child 1 --k1-kernel $k >/dev/null 2>&1)
  python - "$d" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])): acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v))
PY
done
