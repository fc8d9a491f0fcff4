"""K4 micro-benchmark on the GPU box: svo_hip_pose_optimize (wave kernel) vs
svo_hip_pose_optimize_ordered (bit-ordered checker) on B frames x N observations, plus their
agreement.  usage: python scripts/pose_bench.py [B] [N]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from rpg_svo_amd import capi, se3, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
lib = capi.load()
cam = synth.Camera(640, 480, 400.0, 400.0, 320.0, 240.0)
rng = np.random.default_rng(0)
g = torch.Generator().manual_seed(0)
T = synth.make_trajectory(B, seed=3)
px = torch.stack([torch.rand(B, N, generator=g, dtype=torch.float64) * 580 + 30,
                  torch.rand(B, N, generator=g, dtype=torch.float64) * 420 + 30], -1)
f, pos = synth.features_3d(T, cam, px)
# 0.3 px measurement noise, 3 % gross outliers, prior 2e-3 off
px_n = px + 0.3 * torch.randn(px.shape, generator=g, dtype=torch.float64)
f_n, _ = synth.features_3d(T, cam, px_n)
out = torch.rand(B, N, generator=g) < 0.03
pos = pos + out[..., None] * 0.3 * torch.randn(pos.shape, generator=g, dtype=torch.float64)
T0 = se3.mul(se3.exp(rng.normal(size=(B, 6)) * 2e-3), T)
t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
n_t = torch.full((B,), N, dtype=torch.int32, device=dev)
f_t, pos_t = f_n.to(dev).contiguous(), pos.to(dev).contiguous()
lvl = t(rng.integers(0, 3, size=(B, N)), torch.int32)
has0 = torch.ones(B, N, dtype=torch.uint8, device=dev)
T0_t = t(T0, torch.float64)
cam_c = capi.camera(cam)
st = torch.cuda.current_stream(dev).cuda_stream


def run(fn, reps=10):
    Tw, hw = T0_t.clone(), has0.clone()
    Cov = torch.zeros(B, 36, dtype=torch.float64, device=dev)
    stats = torch.zeros(B, 4, dtype=torch.float64, device=dev)
    ran = torch.zeros(B, dtype=torch.int32, device=dev)
    ms = []
    for _ in range(reps + 2):
        Tw.copy_(T0_t)
        hw.copy_(has0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        capi.check(fn(C.byref(cam_c), B, n_t.data_ptr(), N, f_t.data_ptr(), lvl.data_ptr(), pos_t.data_ptr(), hw.data_ptr(),
                      2.0, 10, Tw.data_ptr(), Cov.data_ptr(), stats.data_ptr(), ran.data_ptr(), st))
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms[2:])), Tw.cpu().numpy(), hw.cpu().numpy(), Cov.cpu().numpy(), stats.cpu().numpy(), ran.cpu().numpy()


ms_w, Tw, hw, Cw, sw, rw = run(lib.svo_hip_pose_optimize)
ms_o, To, ho, Co, so, ro = run(lib.svo_hip_pose_optimize_ordered, reps=3)
d = se3.log_norm(Tw, To)
res = {"B": B, "N": N, "ms_wave": ms_w, "ms_ordered": ms_o, "frames_per_s_wave": B / ms_w * 1e3,
       "se3_lognorm_max": float(d.max()), "se3_lognorm_median": float(np.median(d)),
       "pruning_flags_identical": bool(np.array_equal(hw, ho)), "n_pruned_mean": float((1 - ho).sum(1).mean()),
       "stats_max_rel": float(np.max(np.abs(sw - so) / np.maximum(np.abs(so), 1e-30))),
       "cov_max_rel": float(np.max(np.abs(Cw - Co)) / np.max(np.abs(Co))),
       "ran_equal": bool(np.array_equal(rw, ro)), "err_vs_gt_median": float(np.median(se3.log_norm(Tw, T))),
       "algorithmic_bytes_per_frame": N * 52 + 416,
       "achieved_GBs_wave": B * (N * 52 + 416) / (ms_w * 1e-3) / 1e9}
print(json.dumps(res))
