#!/usr/bin/env python
"""Where do the seed verdicts of svo_hip_update_seeds and of the reference's DepthFilter::updateSeeds part ways on the
representative full-track workload of bench.py?  Confusion matrix of the status codes and the first disagreeing seeds
with what Matcher::findEpipolarMatchDirect says about them on the host.  GPU box:  python scripts/seed_parity_debug.py [B]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import pytrack  # noqa: E402
from rpg_svo_amd import capi, se3  # noqa: E402
from rpg_svo_amd.sparse_img_align import SparseImgAlign  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda", 0)
    lib = capi.load()
    W = bench.Workload("vga4_n200_sparse_align", B, dev, 0)
    sia = SparseImgAlign(W.max_level, W.min_level, 30)
    out = sia.alloc_result(B, dev)
    full = bench.FullTrack(W, dev, 0)
    W.run_align(sia, out=out)
    full.step(out.T_cur_from_ref, None)
    torch.cuda.synchronize()
    T_ref_est = full.last["pose"].T_f_w.cpu().numpy()
    trk = pytrack.Track("ref" if pytrack.ref_available() else "orc")
    opt = pytrack.matcher_options(n_pyr_levels=W.n_levels)
    s_b = full.seed_frame_of.cpu().numpy()
    s_fr, s_px, s_f = full.seed_ftr.frame.cpu().numpy(), full.seed_ftr.px.cpu().numpy(), full.seed_ftr.f.cpu().numpy()
    seed0 = {k: v.cpu().numpy() for k, v in full.seed0.items()}
    st_g = full.last["seed_status"].cpu().numpy()
    um = full.seed_unmatched.cpu().numpy()
    age = full.seed_age.cpu().numpy()
    steps_ptr = lib.svo_hip_update_seeds_scan_steps(full.df.last_workspace.data_ptr())
    scan = torch.empty(full.S, dtype=torch.int32, device=dev)
    capi.check(lib.svo_hip_memcpy_d2d(scan.data_ptr(), steps_ptr, full.S * 4, torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
    scan = scan.cpu().numpy()
    conf = {}
    shown = 0
    for b in range(40, B, max(1, (B - 40) // 24)):
        lo, hi = np.searchsorted(s_b, b, "left"), np.searchsorted(s_b, b, "right")
        rows = sorted(set(int(x) for x in s_fr[lo:hi]))
        local = {r: i for i, r in enumerate(rows)}
        cur = len(rows)
        pyrs = [trk.create_img_pyramid(W.images[r].cpu().numpy(), W.n_levels) for r in rows + [b + 1]]
        frames = pytrack.make_frames(pyrs, np.stack([W.T_gt[r] for r in rows] + [T_ref_est[b]]))
        seeds = []
        for k in range(lo, hi):
            sd = pytrack.Seed()
            sd.ftr = pytrack.make_feature(local[int(s_fr[k])], s_px[k], s_f[k])
            sd.batch_id, sd.a, sd.b, sd.mu = 0, float(seed0["a"][k]), float(seed0["b"][k]), float(seed0["mu"][k])
            sd.z_range, sd.sigma2 = float(seed0["z_range"][k]), float(seed0["sigma2"][k])
            seeds.append(sd)
        _, so, io = trk.update_seeds(frames, W.cam, cur, seeds, batch_counter=0, opt=opt)
        stc = np.array([x.status for x in io])
        for k in range(lo, hi):
            key = (int(stc[k - lo]), int(st_g[k]), bool(um[k]))
            conf[key] = conf.get(key, 0) + 1
            if stc[k - lo] != st_g[k] and shown < 12:
                shown += 1
                sd = seeds[k - lo]
                z_inv_min = sd.mu + np.sqrt(sd.sigma2)
                z_inv_max = max(sd.mu - np.sqrt(sd.sigma2), 1e-8)
                ok, r = trk.find_epipolar_match_direct(frames, W.cam, local[int(s_fr[k])], cur, sd.ftr, 1.0 / sd.mu, 1.0 / z_inv_min, 1.0 / z_inv_max, opt)
                print(f"b={b} seed {k} age {age[k]} unmatched {um[k]}: host status {stc[k - lo]} device {st_g[k]}; device scan steps {scan[k]}; "
                      f"host epipolar: ok={ok} epi_length={r['epi_length']:.3f} level={r['search_level']} reject={r['reject']} depth={r['depth']:.4f} px={r['px_cur']}")
    print("confusion (host status, device status, unmatched) -> count:")
    for k in sorted(conf):
        print("  ", k, conf[k])


if __name__ == "__main__":
    main()
