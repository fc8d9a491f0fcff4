"""The north star's "single hipSolver solve" as an independent cross-check: the H_ that the sparse
alignment kernel accumulates (and inverts in LDS by Gauss-Jordan) is solved through hipSOLVER's
batched Cholesky and through numpy; all three must agree."""
import ctypes as C

import numpy as np
import pytest
import torch

from rpg_svo_amd import capi, synth

from helpers import make_batch, run_hip

pytestmark = pytest.mark.gpu


def _solve(lib, H, b, dev):
    B = H.shape[0]
    dH = torch.as_tensor(H, dtype=torch.float64, device=dev).contiguous()
    db = torch.as_tensor(b, dtype=torch.float64, device=dev).contiguous()
    dx = torch.empty(B, 6, dtype=torch.float64, device=dev)
    info = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.svo_hip_solve6_hipsolver_workspace_bytes(B), dtype=torch.uint8, device=dev)
    capi.check(lib.svo_hip_solve6_hipsolver(B, dH.data_ptr(), db.data_ptr(), dx.data_ptr(), info.data_ptr(), ws.data_ptr(),
                                            ws.numel(), torch.cuda.current_stream(dev).cuda_stream), "svo_hip_solve6_hipsolver")
    torch.cuda.synchronize()
    return dx.cpu().numpy(), info.cpu().numpy()


def test_hipsolver_solves_the_kernels_normal_equations(hip_lib, gpu_device):
    seq = synth.make_sequence(33, 200)
    b = make_batch(seq, [(i, i + 1) for i in range(32)], 4)
    _, out, _ = run_hip(b, 3, 0)
    H = out.H.cpu().numpy().reshape(-1, 6, 6)
    assert np.allclose(H, H.transpose(0, 2, 1)) and (np.linalg.eigvalsh(H) > 0).all()   # J'J of >= 150 patches: SPD
    rng = np.random.default_rng(0)
    rhs = rng.normal(size=(32, 6)) * np.sqrt(np.abs(np.diagonal(H, axis1=1, axis2=2)))
    x, info = _solve(hip_lib, H, rhs, gpu_device)
    assert (info == 0).all()
    x_np = np.linalg.solve(H, rhs[..., None])[..., 0]
    assert np.allclose(x, x_np, rtol=1e-9, atol=1e-14)
    # covariance of the aligned pose (inverse Fisher information, sparse_img_align.cpp:77-82), column by column
    sigma2 = 5e-4 * 255 * 255
    cov = np.stack([_solve(hip_lib, H / sigma2, np.tile(np.eye(6)[j], (32, 1)), gpu_device)[0] for j in range(6)], axis=2)
    assert np.allclose(cov, np.linalg.inv(H / sigma2), rtol=1e-8, atol=1e-16)


def test_hipsolver_flags_indefinite_systems(hip_lib, gpu_device):
    H = np.tile(np.eye(6), (3, 1, 1))
    H[1, 2, 2] = -1.0        # not positive definite: potrf reports the failing minor
    x, info = _solve(hip_lib, H, np.ones((3, 6)), gpu_device)
    assert info[0] == 0 and info[2] == 0 and info[1] > 0
    assert np.allclose(x[0], 1.0) and np.allclose(x[2], 1.0)
    assert hip_lib.svo_hip_solve6_hipsolver(1, None, None, None, None, None, 0, None) == -1
