cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { SVO_HIP_LIB=$R/build/variants/lib$1.so timeout 120 python $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 ${@:2} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['roofline']['kernel_ms_avg'],4), round(d['value']))"; }
for v in B3 W3; do
  echo "== $v vga B8192: $(run $v) | $(run $v)"
  echo "== $v vga B16384: $(run $v --batch 16384)"
  echo "== $v 752 B8192: $(run $v --workload svo_default_752_l4to2_n120)"
  echo "== $v vga noise2: $(run $v --noise 2)"
  SVO_HIP_LIB=$R/build/variants/lib$v.so timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_$v -o f -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
  python - <<EOF
import csv,glob
fs=glob.glob("$R/gpurun_out/pmc_$v/**/*counter_collection.csv", recursive=True)
tot=[];
for f in fs:
    for r in csv.DictReader(open(f)):
        if 'sia_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': tot.append(float(r['Counter_Value']))
print("$v FETCH_SIZE per dispatch (KiB):", sum(tot)/max(1,len(tot)), len(tot))
EOF
  rm -rf $R/gpurun_out/pmc_$v
done
