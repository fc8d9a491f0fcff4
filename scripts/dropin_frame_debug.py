"""Per-frame deviation of the drop-in pipeline from the CPU reference on the 120-frame test sequence
(tests/test_dropin_pipeline.py): where does the trajectory start to differ, and by how much."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
import numpy as np
import pypipeline as pp
from rpg_svo_amd import synth, se3
import test_dropin_pipeline as t
cam, imgs, T = t._sequence(120)
dn = os.open(os.devnull, os.O_WRONLY); sv = os.dup(2); os.dup2(dn, 2)
ref = pp.run_sequence("ref", cam, imgs, T)
hip = pp.run_sequence("hip", cam, imgs, T)
os.dup2(sv, 2)
Tr = np.stack([r["T_f_w"] for r in ref]); Th = np.stack([r["T_f_w"] for r in hip])
d = se3.log_norm(Th, Tr)
print("max %.3e median %.3e" % (d.max(), np.median(d)))
prev = 1e-12
for i in range(len(ref)):
    a, b = ref[i], hip[i]
    flag = d[i] > 3 * prev
    if flag or i % 20 == 0:
        print(i, f"{d[i]:.2e}", "JUMP" if flag else "", "kf", a["is_keyframe"], b["is_keyframe"], "n_obs", a["n_obs"], b["n_obs"],
              "tracked", a["img_align_n_tracked"], b["img_align_n_tracked"], "seeds", a["n_seeds"], b["n_seeds"])
    prev = max(d[i], 1e-12)
