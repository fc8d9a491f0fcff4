"""Small numpy SE(3) helpers (poses as [...,12] = R row-major | t).

Host-side bookkeeping only (marshalling T_cur_from_ref, parity metrics); the
hot path never calls these.
"""
from __future__ import annotations

import numpy as np


def identity(n: int | None = None) -> np.ndarray:
    T = np.concatenate([np.eye(3).ravel(), np.zeros(3)])
    return T if n is None else np.tile(T, (n, 1))


def split(T: np.ndarray):
    T = np.asarray(T, dtype=np.float64)
    return T[..., :9].reshape(T.shape[:-1] + (3, 3)), T[..., 9:]


def join(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    return np.concatenate([R.reshape(R.shape[:-2] + (9,)), t], axis=-1)


def mul(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    Ra, ta = split(A)
    Rb, tb = split(B)
    return join(Ra @ Rb, (Ra @ tb[..., None])[..., 0] + ta)


def inv(A: np.ndarray) -> np.ndarray:
    R, t = split(A)
    Rt = np.swapaxes(R, -1, -2)
    return join(Rt, -(Rt @ t[..., None])[..., 0])


def apply(A: np.ndarray, p: np.ndarray) -> np.ndarray:
    R, t = split(A)
    return (R @ p[..., None])[..., 0] + t


def hat(w: np.ndarray) -> np.ndarray:
    O = np.zeros(w.shape[:-1] + (3, 3))
    O[..., 0, 1], O[..., 0, 2] = -w[..., 2], w[..., 1]
    O[..., 1, 0], O[..., 1, 2] = w[..., 2], -w[..., 0]
    O[..., 2, 0], O[..., 2, 1] = -w[..., 1], w[..., 0]
    return O


def exp(xi: np.ndarray) -> np.ndarray:
    """xi = [upsilon(3), omega(3)] (Sophus ordering)."""
    xi = np.asarray(xi, dtype=np.float64)
    u, w = xi[..., :3], xi[..., 3:]
    th = np.linalg.norm(w, axis=-1)[..., None, None]
    O = hat(w)
    O2 = O @ O
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    A = np.where(small, 1.0 - th**2 / 6, np.sin(ths) / ths)
    B = np.where(small, 0.5 - th**2 / 24, (1 - np.cos(ths)) / ths**2)
    Cc = np.where(small, 1.0 / 6 - th**2 / 120, (ths - np.sin(ths)) / ths**3)
    I = np.eye(3)
    R = I + A * O + B * O2
    V = I + B * O + Cc * O2
    return join(R, (V @ u[..., None])[..., 0])


def log(T: np.ndarray) -> np.ndarray:
    R, t = split(T)
    tr = np.clip((np.trace(R, axis1=-2, axis2=-1) - 1) / 2, -1, 1)
    th = np.arccos(tr)
    w_raw = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1)
    # |w_raw| = 2 sin(th): accurate for small angles, unlike arccos of the trace
    s2 = np.linalg.norm(w_raw, axis=-1)
    th = np.where(tr > 0.9, np.arcsin(np.clip(s2 / 2, -1, 1)), th)
    small = th < 1e-8
    ths = np.where(small, 1.0, th)
    k = np.where(small, 0.5 + th**2 / 12, ths / (2 * np.sin(ths)))
    w = w_raw * k[..., None]
    O = hat(w)
    O2 = O @ O
    th2 = (th**2)[..., None, None]
    thb = ths[..., None, None]
    c = np.where(small[..., None, None], 1.0 / 12, (1 - thb / (2 * np.tan(thb / 2))) / np.where(small[..., None, None], 1.0, th2))
    Vi = np.eye(3) - 0.5 * O + c * O2
    return np.concatenate([(Vi @ t[..., None])[..., 0], w], axis=-1)


def log_norm(A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """|| log(A * B^-1) ||  per pose: the SE(3) parity metric."""
    return np.linalg.norm(log(mul(A, inv(B))), axis=-1)
