"""Per-frame comparison of the reference pipeline and the drop-in under a distorted camera."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
import numpy as np
import pypipeline as pp
from rpg_svo_amd import synth, se3
from helpers import camera_models
kind = sys.argv[1] if len(sys.argv) > 1 else "atan"
cam = camera_models()[kind]
T = synth.make_trajectory(80, seed=7, max_step=0.02, max_rot_deg=0.3)
imgs = synth.render(synth.make_texture(seed=12345), T, cam).numpy()
dn = os.open(os.devnull, os.O_WRONLY); sv = os.dup(2); os.dup2(dn, 2)
ref = pp.run_sequence("ref", cam, imgs, T)
hip = pp.run_sequence("hip", cam, imgs, T)
os.dup2(sv, 2)
Tr = np.stack([r["T_f_w"] for r in ref]); Th = np.stack([r["T_f_w"] for r in hip])
d = se3.log_norm(Th, Tr)
for i in range(len(ref)):
    a, b = ref[i], hip[i]
    print(i, f"{d[i]:.2e}", "kf", a["is_keyframe"], b["is_keyframe"], "n_obs", a["n_obs"], b["n_obs"], "tracked", a["img_align_n_tracked"], b["img_align_n_tracked"],
          "mps", a["repr_n_mps"], b["repr_n_mps"], "newref", a["repr_n_new_references"], b["repr_n_new_references"], "edges", a["sfba_n_edges_final"], b["sfba_n_edges_final"],
          "seeds", a["n_seeds"], b["n_seeds"])
