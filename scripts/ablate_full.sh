R=$GRAFT_REPO_ROOT
for v in "$@"; do
  L=$R/build/variants/lib$v.so; [ "$v" = "main" ] && L=$R/rpg_svo_amd/lib/libsvo_hip.so
  echo -n "== $v: "; SVO_HIP_LIB=$L timeout 200 python $R/bench.py --pipeline full --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), {k: round(v,3) for k,v in d['stages_ms'].items()})"
done
