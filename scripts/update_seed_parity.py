#!/usr/bin/env python3
"""How far is the device's DepthFilter::updateSeed (depth_filter.cpp:309-332, all float) from the CPU's on IDENTICAL
float inputs?  Prints, per output (a, b, mu, sigma2): the share of seeds with identical bits and the largest relative
deviation among the others -- the numbers the tolerances of tests/test_tracking_gpu.py::test_update_seed_batch are set
from.  Run on the GPU box."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pytrack  # noqa: E402
from rpg_svo_amd import tracking  # noqa: E402


def main(S=200000, which="orc"):
    orc = pytrack.Track(which)
    rng = np.random.default_rng(12)
    dev = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device="cuda:0")
    seeds = []
    for i in range(S):
        s = orc.seed_init(rng.uniform(0.5, 5), rng.uniform(0.2, 0.5))
        s.a, s.b = np.float32(rng.uniform(5, 30)), np.float32(rng.uniform(5, 30))
        if i % 3 == 0:   # a seed that has been updated a few times: smaller variance
            s.sigma2 = np.float32(s.sigma2 * 10.0 ** rng.uniform(-4, 0))
        seeds.append(s)
    x = np.array([1.0 / (1.0 / s.mu * rng.uniform(0.7, 1.4)) for s in seeds], dtype=np.float32)
    tau2 = (10.0 ** rng.uniform(-8, -1, size=S)).astype(np.float32)
    ss = tracking.SeedSet(a=dev([s.a for s in seeds], torch.float32), b=dev([s.b for s in seeds], torch.float32),
                          mu=dev([s.mu for s in seeds], torch.float32), z_range=dev([s.z_range for s in seeds], torch.float32),
                          sigma2=dev([s.sigma2 for s in seeds], torch.float32), batch_id=dev(np.zeros(S), torch.int32))
    tracking.DepthFilter.update_seed(dev(x, torch.float32), dev(tau2, torch.float32), ss)
    torch.cuda.synchronize()
    got = np.stack([t.cpu().numpy() for t in (ss.a, ss.b, ss.mu, ss.sigma2)], axis=1)
    want = np.array([[n.a, n.b, n.mu, n.sigma2] for n in (orc.update_seed(x[i], tau2[i], seeds[i]) for i in range(S))], dtype=np.float32)
    fin = np.isfinite(want).all(axis=1) & np.isfinite(got).all(axis=1)
    out = {"seeds": int(S), "finite": int(fin.sum()), "checker": which}
    for k, name in enumerate(("a", "b", "mu", "sigma2")):
        g, w = got[fin, k], want[fin, k]
        same = g.view(np.uint32) == w.view(np.uint32)
        rel = np.abs(g.astype(np.float64) - w) / np.maximum(np.abs(w), 1e-30)
        out[name] = {"identical_bits_frac": float(same.mean()), "max_rel_dev": float(rel.max()),
                     "p999_rel_dev": float(np.quantile(rel, 0.999)), "ulps_max": int(np.abs(g.view(np.int32).astype(np.int64) - w.view(np.int32)).max())}
    print(json.dumps(out))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200000, sys.argv[2] if len(sys.argv) > 2 else "orc")
