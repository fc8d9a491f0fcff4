"""rpg_svo_amd -- MI355X (gfx950) implementation of SVO's tracking hot path.

Host-side mirrors of the reference interfaces (svo::SparseImgAlign, image
pyramid, ...) over the C ABI of libsvo_hip.so (include/svo_hip.h).  PyTorch is
used only as plumbing: device memory, streams and torch.distributed.
"""
from . import capi  # noqa: F401
from .capi import SvoHipError  # noqa: F401

__all__ = ["capi", "SvoHipError"]
