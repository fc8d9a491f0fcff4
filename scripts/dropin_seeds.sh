# trajectory agreement of the drop-in on several synthetic sequences
R=${GRAFT_REPO_ROOT:-.}
for s in "$@"; do
  timeout 300 python $R/scripts/dropin_compare.py --frames 150 --seed $s --json /tmp/dc_$s.json > /dev/null 2>&1
  python - <<EOF
import json
d=json.load(open("/tmp/dc_$s.json"))
print("seed $s max %.2e median %.2e ATE %.2e m  kf ref/hip %d/%d  n_obs same %.3f" % (d["se3_lognorm_hip_vs_ref_max"], d["se3_lognorm_hip_vs_ref_median"], d["ate_rmse_hip_vs_ref_m"], d["keyframes"]["ref"], d["keyframes"]["hip"], d["identical_counter_fraction"]["n_obs"]))
EOF
done
