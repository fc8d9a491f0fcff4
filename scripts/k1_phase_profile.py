"""Where a workgroup of K1 spends its time under full load: the SIA_PROFILE build
(build/variants/libP.so, scripts/build_variant.sh P sparse_align -DSIA_PROFILE) writes per-phase
shader-clock totals of wave 0 of every workgroup over H_out.  Prints the mean per Gauss-Newton
iteration and per frame.  usage: SVO_HIP_LIB=build/variants/libP.so python scripts/k1_phase_profile.py [B]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from rpg_svo_amd import capi
from rpg_svo_amd.sparse_img_align import SparseImgAlign

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda:0")
lib = capi.load()
W = bench.Workload("vga4_n200_sparse_align", B, dev)
sia = SparseImgAlign(W.max_level, W.min_level, 30)
out = sia.alloc_result(B, dev)
for _ in range(3):
    W.run_align(sia, out=out)
torch.cuda.synchronize()
H = out.H.cpu().numpy()[:, :8]
names = ["pixel_work", "reduce+lds_write", "barrier1", "h_rebuild(changed)", "solve+barrier2+pose_read", "per_level_precompute", "total", "iterations"]
it = H[:, 7].mean()
res = {"B": B, "mean_iterations": float(it)}
for k, n in enumerate(names[:6]):
    res[n + "_cycles_per_frame"] = float(H[:, k].mean())
    if k < 5:
        res[n + "_cycles_per_iteration"] = float(H[:, k].sum() / H[:, 7].sum())
res["total_cycles_per_frame"] = float(H[:, 6].mean())
res["accounted_frac"] = float(H[:, :6].sum() / H[:, 6].sum())
print(json.dumps(res, indent=1))
