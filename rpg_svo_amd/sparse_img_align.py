"""Batched host mirror of svo::SparseImgAlign (svo/include/svo/sparse_img_align.h
:33-79, svo/src/sparse_img_align.cpp) over svo_hip_sparse_align.

Same constructor arguments and meaning as the reference class; `run` takes a
batch of (reference frame, current frame) problems whose pyramids live in a
PyramidStore instead of two FramePtr.  All heavy state stays on the device.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import capi, se3
from .pyramid import PyramidStore, _stream_ptr


@dataclass
class SparseAlignResult:
    T_cur_from_ref: torch.Tensor   # [B,12] f64
    H: torch.Tensor                # [B,36] f64  (H_ of the last evaluated iteration)
    n_tracked: torch.Tensor        # [B] i32     (return value of run(), n_meas_/16)
    iters: torch.Tensor            # [B,8] i32   residual evaluations per pyramid level
    chi2: torch.Tensor             # [B] f64
    status: torch.Tensor           # [B] i32     capi.SIA_STOP bit


def marshal_problem(T_ref_w: np.ndarray, T_cur_w: np.ndarray, f: np.ndarray, pos: np.ndarray):
    """What the host wrapper computes per problem before the device call
    (sparse_img_align.cpp:59 and :107-108): T_cur_from_ref and xyz_ref = f*depth."""
    T_ref_w = np.asarray(T_ref_w, dtype=np.float64)
    T_cur_w = np.asarray(T_cur_w, dtype=np.float64)
    T_cr = se3.mul(T_cur_w, se3.inv(T_ref_w))
    ref_pos = se3.inv(T_ref_w)[..., 9:]                      # Frame::pos()
    depth = np.linalg.norm(np.asarray(pos) - ref_pos[..., None, :], axis=-1)
    xyz_ref = np.asarray(f) * depth[..., None]
    return T_cr, xyz_ref


class SparseImgAlign:
    GaussNewton = 0  # vk::NLLSSolver::Method; LevenbergMarquardt is never used by SVO

    def __init__(self, max_level: int, min_level: int, n_iter: int = 30, method: int = 0,
                 display: bool = False, verbose: bool = False):
        if method != self.GaussNewton:
            raise ValueError("only GaussNewton is used by the reference pipeline (frame_handler_mono.cpp:136-137)")
        self.max_level = max_level
        self.min_level = min_level
        self.n_iter = n_iter
        self.eps = 0.000001  # sparse_img_align.cpp:40
        self.verbose = verbose
        self.lib = capi.load()
        # "auto": svo_hip_sparse_align picks the kernel (one wave per frame for batches of >= 1024 frames with <= 192
        # patches, one workgroup per frame otherwise); "workgroup": the workgroup-per-frame kernel whatever the sizes
        self.kernel = "auto"

    def params(self, cam) -> capi.SiaParams:
        d = tuple(getattr(cam, "d", (0.0,) * 5))
        return capi.SiaParams(cam.fx, cam.fy, cam.cx, cam.cy, self.max_level, self.min_level, self.n_iter,
                              int(getattr(cam, "model", 0)), self.eps, (C.c_double * 5)(*d))

    def alloc_result(self, B: int, device) -> SparseAlignResult:
        return SparseAlignResult(
            torch.empty(B, 12, dtype=torch.float64, device=device), torch.empty(B, 36, dtype=torch.float64, device=device),
            torch.empty(B, dtype=torch.int32, device=device), torch.empty(B, capi.MAX_LEVELS, dtype=torch.int32, device=device),
            torch.empty(B, dtype=torch.float64, device=device), torch.empty(B, dtype=torch.int32, device=device))

    def run(self, store: PyramidStore, cam, ref_slot: torch.Tensor, cur_slot: torch.Tensor, n: torch.Tensor,
            px: torch.Tensor, xyz_ref: torch.Tensor, T_cur_from_ref: torch.Tensor,
            valid: torch.Tensor | None = None, out: SparseAlignResult | None = None) -> SparseAlignResult:
        """All tensors on store.device.  px [B,Ns,2] f64, xyz_ref [B,Ns,3] f64,
        T_cur_from_ref [B,12] f64 prior, n [B] i32, slots [B] i32, valid [B,Ns] u8."""
        dev = store.device
        B, ns = px.shape[0], px.shape[1]
        for t, dt in ((ref_slot, torch.int32), (cur_slot, torch.int32), (n, torch.int32), (px, torch.float64),
                      (xyz_ref, torch.float64), (T_cur_from_ref, torch.float64)):
            assert t.is_cuda and t.dtype == dt and t.is_contiguous(), "device-resident contiguous inputs required"
        if valid is not None:
            assert valid.is_cuda and valid.dtype == torch.uint8 and valid.is_contiguous()
        if out is None:
            out = self.alloc_result(B, dev)
        P = self.params(cam)
        if self.kernel not in ("auto", "workgroup"):
            raise ValueError("SparseImgAlign.kernel must be 'auto' or 'workgroup'")
        entry = self.lib.svo_hip_sparse_align if self.kernel == "auto" else self.lib.svo_hip_sparse_align_workgroup
        capi.check(entry(
            C.byref(store.layout), store.ptr, B, ref_slot.data_ptr(), cur_slot.data_ptr(), n.data_ptr(), ns,
            px.data_ptr(), xyz_ref.data_ptr(), valid.data_ptr() if valid is not None else None, C.byref(P),
            T_cur_from_ref.data_ptr(), out.T_cur_from_ref.data_ptr(), out.H.data_ptr(), out.n_tracked.data_ptr(),
            out.iters.data_ptr(), out.chi2.data_ptr(), out.status.data_ptr(), _stream_ptr(dev)), "svo_hip_sparse_align")
        return out

    @staticmethod
    def fisher_information(H: torch.Tensor) -> torch.Tensor:
        """SparseImgAlign::getFisherInformation (sparse_img_align.cpp:77-82)."""
        sigma_i_sq = 5e-4 * 255 * 255
        return H / sigma_i_sq
