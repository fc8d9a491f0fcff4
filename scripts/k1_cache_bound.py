#!/usr/bin/env python
"""How much of K1's time is exposed memory latency?  The same arithmetic twice: B copies of ONE frame pair, once with
every problem reading the same two pyramid slots (every window fetch after the first hits L2) and once with every
problem reading its own copy of the two pyramids (the fetches of a batch go to HBM as in the benchmark).  The gap
bounds what any prefetching of the window fetches could buy.  GPU box:  python scripts/k1_cache_bound.py [B]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rpg_svo_amd import capi  # noqa: E402
from rpg_svo_amd.pyramid import PyramidStore  # noqa: E402
from rpg_svo_amd.sparse_img_align import SparseImgAlign  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    dev = torch.device("cuda", 0)
    lib = capi.load()
    ev = bench.Events(lib, dev)
    W = bench.Workload("vga4_n200_sparse_align", 64, dev, 0)
    sia = SparseImgAlign(W.max_level, W.min_level, 30)
    out = sia.alloc_result(B, dev)
    res = {}
    for k in (3, 17, 40):  # three different frame pairs
        pair = W.images[k:k + 2]
        store = PyramidStore(W.width, W.height, W.n_levels, 2 * B, device=dev)
        store.load_images(pair.repeat(B, 1, 1))  # slots 2b, 2b+1 = the pair
        rep = lambda t: t[k:k + 1].expand(B, *t.shape[1:]).contiguous()
        px, xyz, T_in, n = rep(W.px_all), rep(W.xyz_t), rep(W.T_in), W.n_t[:1].expand(B).contiguous()
        b = torch.arange(B, dtype=torch.int32, device=dev)
        same = (torch.zeros_like(b), torch.ones_like(b))
        own = ((2 * b).contiguous(), (2 * b + 1).contiguous())
        for name, (r, c) in (("same_two_slots", same), ("own_copy_per_problem", own)):
            ms = ev.time(lambda: sia.run(store, W.cam, r, c, n, px, xyz, T_in, out=out), 10, warmup=3)
            res.setdefault(name, []).append(ms)
        res.setdefault("iterations", []).append(int(out.iters[0].sum().item()))
        del store
        torch.cuda.empty_cache()
    res["ratio_own_over_same"] = [a / b for a, b in zip(res["own_copy_per_problem"], res["same_two_slots"])]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
