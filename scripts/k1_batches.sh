#!/bin/bash
# K1 throughput against the batch size (headline workload) and both kernels on the 752x480 / 120-patch workload
cd "$(dirname "$0")/.."
run() { python bench.py --no-cpu-baseline --extras none --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('fps', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms_avg'],4), d['roofline']['kernel'].split()[0])"; }
for b in 1024 4096 8192 16384 32768; do echo -n "vga4_n200 batch $b: "; run --batch $b; done
for k in auto workgroup; do echo -n "svo_default_752 n120 batch 16384 $k: "; run --workload svo_default_752_l4to2_n120 --k1-kernel $k; done
