"""Synthetic textured-plane sequences for parity tests and the benchmark.

Stands in for the reference's `sin2_tex2_h1_v8_d` dataset (svo/test/README.md,
test_utils.h:30-41), which is not available offline: a textured plane z = 0
seen by a downward-looking pinhole camera at ~2 m (the reference's test
trajectory also sits at z = 2.0, svo/test/test_matcher.cpp:52-53), rendered by
exact ray/plane intersection + bilinear texture lookup, with analytic depth.
Everything is seeded and deterministic.  torch is used so the same code runs on
CPU (tests) and on the GPU (bench data generation, untimed).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch


CAM_PINHOLE, CAM_PINHOLE_RADTAN, CAM_ATAN = 0, 1, 2   # include/svo_hip.h SVO_HIP_CAM_*


@dataclass
class Camera:
    """A vk::AbstractCamera: intrinsics + model tag + distortion parameters as svo_hip_camera holds
    them (radtan: d0..d4 = k1 k2 p1 p2 k3; ATAN: s, 1/s, 2 tan(s/2), 1/(2 tan(s/2)))."""
    width: int
    height: int
    fx: float
    fy: float
    cx: float
    cy: float
    model: int = CAM_PINHOLE
    d: tuple = (0.0, 0.0, 0.0, 0.0, 0.0)
    ctor: tuple = ()   # ATAN: the normalised constructor arguments (fx, fy, cx, cy, s)

    @staticmethod
    def vga() -> "Camera":
        return Camera(640, 480, 400.0, 400.0, 320.0, 240.0)

    @staticmethod
    def radtan(width, height, fx, fy, cx, cy, d0, d1=0.0, d2=0.0, d3=0.0, d4=0.0) -> "Camera":
        """vk::PinholeCamera(width, height, fx, fy, cx, cy, d0..d4), e.g. svo_ros/param/camera_pinhole.yaml."""
        if abs(d0) <= 0.0000001:  # vikit: distortion_ = fabs(d0) > 0.0000001
            return Camera(width, height, fx, fy, cx, cy)
        return Camera(width, height, fx, fy, cx, cy, CAM_PINHOLE_RADTAN, (d0, d1, d2, d3, d4))

    @staticmethod
    def atan(width, height, fx, fy, cx, cy, s) -> "Camera":
        """vk::ATANCamera with the normalised parameters of svo_ros/param/camera_atan.yaml."""
        d = (0.0,) * 5
        if s != 0.0:
            tans = 2.0 * math.tan(s / 2.0)
            d = (s, 1.0 / s, tans, 1.0 / tans, 0.0)
        return Camera(width, height, width * fx, height * fy, cx * width - 0.5, cy * height - 0.5, CAM_ATAN, d,
                      (fx, fy, cx, cy, s))


def _lib(x):
    return torch if isinstance(x, torch.Tensor) else np


def cam_distort(cam: Camera, x, y):
    """Normalised image-plane point (x, y) = project2d(xyz) -> pixel: vk::*Camera::world2cam(uv)."""
    if cam.model == CAM_PINHOLE_RADTAN:
        d = cam.d
        r2 = x * x + y * y
        r4 = r2 * r2
        r6 = r4 * r2
        a1, a2, a3 = 2 * x * y, r2 + 2 * x * x, r2 + 2 * y * y
        cdist = 1 + d[0] * r2 + d[1] * r4 + d[4] * r6
        xd = x * cdist + d[2] * a1 + d[3] * a2
        yd = y * cdist + d[2] * a3 + d[3] * a1
        return xd * cam.fx + cam.cx, yd * cam.fy + cam.cy
    if cam.model == CAM_ATAN and cam.d[0] != 0.0:
        L = _lib(x)
        r = L.sqrt(x * x + y * y)
        rs = L.where(r < 0.001, L.ones_like(r), r)
        factor = L.where(r < 0.001, L.ones_like(r), cam.d[1] * L.arctan(rs * cam.d[2]) / rs)
        return cam.cx + cam.fx * (factor * x), cam.cy + cam.fy * (factor * y)
    return cam.fx * x + cam.cx, cam.fy * y + cam.cy


def cam_undistort(cam: Camera, u, v):
    """Pixel -> normalised image-plane point (the un-normalised cam2world), accurate inverse of
    cam_distort (the radial-tangential model is inverted by fixed-point iteration in double)."""
    x0, y0 = (u - cam.cx) / cam.fx, (v - cam.cy) / cam.fy
    if cam.model == CAM_PINHOLE_RADTAN:
        d = cam.d
        x, y = x0, y0
        for _ in range(30):
            r2 = x * x + y * y
            icdist = 1.0 / (1 + ((d[4] * r2 + d[1]) * r2 + d[0]) * r2)
            dx = 2 * d[2] * x * y + d[3] * (r2 + 2 * x * x)
            dy = d[2] * (r2 + 2 * y * y) + 2 * d[3] * x * y
            x, y = (x0 - dx) * icdist, (y0 - dy) * icdist
        return x, y
    if cam.model == CAM_ATAN and cam.d[0] != 0.0:
        L = _lib(x0)
        dist_r = L.sqrt(x0 * x0 + y0 * y0)
        r = L.tan(dist_r * cam.d[0]) * cam.d[3]
        safe = L.where(dist_r > 0.01, dist_r, L.ones_like(dist_r))
        k = L.where(dist_r > 0.01, r / safe, L.ones_like(dist_r))
        return k * x0, k * y0
    return x0, y0


TEX_SIZE = 1024        # texels
TEX_EXTENT = 8.0       # metres covered by the texture (centred on the origin)


def make_texture(size: int = TEX_SIZE, seed: int = 12345) -> torch.Tensor:
    """Band-limited noise texture in [16, 240], float32 [size, size] (CPU)."""
    g = torch.Generator().manual_seed(seed)
    tex = torch.zeros(size, size, dtype=torch.float64)
    for sigma, amp in ((1.5, 0.35), (4.0, 0.6), (12.0, 1.0), (32.0, 0.8)):
        noise = torch.rand(size, size, generator=g, dtype=torch.float64) - 0.5
        k = int(4 * sigma) | 1
        xs = torch.arange(k, dtype=torch.float64) - k // 2
        ker = torch.exp(-0.5 * (xs / sigma) ** 2)
        ker /= ker.sum()
        n4 = noise[None, None]
        n4 = torch.nn.functional.conv2d(torch.nn.functional.pad(n4, (k // 2, k // 2, 0, 0), mode="circular"), ker.view(1, 1, 1, k))
        n4 = torch.nn.functional.conv2d(torch.nn.functional.pad(n4, (0, 0, k // 2, k // 2), mode="circular"), ker.view(1, 1, k, 1))
        n4 = n4[0, 0]
        tex += amp * n4 / n4.std()
    tex = (tex - tex.mean()) / tex.std()
    tex = 128.0 + 45.0 * tex
    return tex.clamp(16.0, 240.0).to(torch.float32)


def make_trajectory(n_frames: int, seed: int = 12345, height: float = 2.0,
                    max_step: float = 0.012, max_rot_deg: float = 0.25) -> np.ndarray:
    """T_f_w for n_frames, float64 [n,12].  Smooth random walk: <= max_step*height
    metres and <= max_rot_deg degrees per frame."""
    rng = np.random.default_rng(seed)
    from . import se3
    R0 = np.diag([1.0, -1.0, -1.0])  # camera looks down: z_cam = -z_world
    T = np.zeros((n_frames, 12))
    c = np.array([0.0, 0.0, height])
    rv = np.zeros(3)
    vel = rng.normal(size=3)
    ang = rng.normal(size=3)
    for i in range(n_frames):
        Rs = se3.split(se3.exp(np.concatenate([np.zeros(3), rv])))[0]
        R = Rs @ R0
        T[i] = se3.join(R, -R @ c)
        vel = 0.85 * vel + 0.5 * rng.normal(size=3)
        ang = 0.85 * ang + 0.5 * rng.normal(size=3)
        step = vel / max(np.linalg.norm(vel), 1e-9) * max_step * height * rng.uniform(0.3, 1.0)
        step[2] *= 0.3
        c = c + step
        c[:2] = np.clip(c[:2], -1.0, 1.0)
        c[2] = np.clip(c[2], 0.85 * height, 1.15 * height)
        drv = ang / max(np.linalg.norm(ang), 1e-9) * math.radians(max_rot_deg) * rng.uniform(0.3, 1.0)
        rv = np.clip(rv + drv, -0.12, 0.12)
    return T


def _plane_points(T_f_w: torch.Tensor, cam: Camera, u: torch.Tensor, v: torch.Tensor):
    """World points on z=0 seen at pixels (u,v) of frames T_f_w [n,12]; u,v broadcast
    against the leading n.  Returns (X [...,3], cam centre [n,3])."""
    R = T_f_w[:, :9].reshape(-1, 3, 3)
    t = T_f_w[:, 9:]
    c = -(R.transpose(1, 2) @ t[..., None])[..., 0]          # camera centre in world
    dx, dy = cam_undistort(cam, u, v)
    d_c = torch.stack([dx, dy, torch.ones_like(dx)], dim=-1)  # [..., 3]
    shape = [R.shape[0]] + [1] * (d_c.dim() - 2)
    Rt = R.transpose(1, 2).reshape(shape + [3, 3])
    d_w = (Rt @ d_c[..., None])[..., 0]
    cc = c.reshape(shape + [3])
    s = -cc[..., 2] / d_w[..., 2]
    X = cc + s[..., None] * d_w
    return X, c


def render(tex: torch.Tensor, T_f_w, cam: Camera, device="cpu", chunk: int = 16) -> torch.Tensor:
    """uint8 [n, h, w] images of the textured plane."""
    T = torch.as_tensor(np.asarray(T_f_w), dtype=torch.float64, device=device)
    tex = tex.to(device=device, dtype=torch.float32)
    S = tex.shape[0]
    n = T.shape[0]
    out = torch.empty(n, cam.height, cam.width, dtype=torch.uint8, device=device)
    vv, uu = torch.meshgrid(torch.arange(cam.height, dtype=torch.float64, device=device),
                            torch.arange(cam.width, dtype=torch.float64, device=device), indexing="ij")
    for i0 in range(0, n, chunk):
        Tc = T[i0:i0 + chunk]
        X, _ = _plane_points(Tc, cam, uu[None], vv[None])
        # texture coordinates (texel units), bilinear
        tu = (X[..., 0] / TEX_EXTENT + 0.5) * (S - 1)
        tv = (X[..., 1] / TEX_EXTENT + 0.5) * (S - 1)
        tu = tu.clamp(0, S - 1.001)
        tv = tv.clamp(0, S - 1.001)
        iu = tu.floor().long()
        iv = tv.floor().long()
        fu = (tu - iu).float()
        fv = (tv - iv).float()
        a = tex[iv, iu]
        b = tex[iv, iu + 1]
        c = tex[iv + 1, iu]
        d = tex[iv + 1, iu + 1]
        img = (a * (1 - fu) + b * fu) * (1 - fv) + (c * (1 - fu) + d * fu) * fv
        out[i0:i0 + chunk] = img.round().clamp(0, 255).to(torch.uint8)
    return out


def select_features(images: torch.Tensor, n_feat: int, margin: int = 28, cell: int = 32) -> torch.Tensor:
    """Per image: strongest-gradient pixel of each cell x cell grid cell, then the
    n_feat best cells (FAST/Shi-Tomasi grid stand-in, svo/src/feature_detection.cpp
    :66-114).  Returns float64 [n, n_feat, 2] level-0 pixel coordinates (u, v)."""
    if images.shape[0] > 256:  # bound peak memory for large replay batches
        return torch.cat([select_features(images[i:i + 256], n_feat, margin, cell)
                          for i in range(0, images.shape[0], 256)], dim=0)
    n, h, w = images.shape
    img = images.float()
    gx = torch.zeros_like(img)
    gy = torch.zeros_like(img)
    gx[:, :, 1:-1] = img[:, :, 2:] - img[:, :, :-2]
    gy[:, 1:-1, :] = img[:, 2:, :] - img[:, :-2, :]
    score = gx * gx + gy * gy
    # smooth the score a little so a single noisy pixel does not win
    score = torch.nn.functional.avg_pool2d(score[:, None], 3, 1, 1)[:, 0]
    score[:, :margin, :] = -1
    score[:, h - margin:, :] = -1
    score[:, :, :margin] = -1
    score[:, :, w - margin:] = -1
    hc, wc = h // cell, w // cell
    s = score[:, :hc * cell, :wc * cell].reshape(n, hc, cell, wc, cell).permute(0, 1, 3, 2, 4).reshape(n, hc * wc, cell * cell)
    best, idx = s.max(dim=-1)
    cy = torch.arange(hc, device=images.device).repeat_interleave(wc)
    cx = torch.arange(wc, device=images.device).repeat(hc)
    v = cy[None] * cell + idx // cell
    u = cx[None] * cell + idx % cell
    if hc * wc < n_feat:
        raise ValueError(f"grid {hc}x{wc} has fewer cells than n_feat={n_feat}")
    top = best.topk(n_feat, dim=-1).indices
    top, _ = top.sort(dim=-1)
    u = torch.gather(u, 1, top)
    v = torch.gather(v, 1, top)
    return torch.stack([u, v], dim=-1).to(torch.float64)


def features_3d(T_f_w, cam: Camera, px: torch.Tensor):
    """Unit bearings f [n,N,3] and world points pos [n,N,3] (plane hit) for px [n,N,2]."""
    T = torch.as_tensor(np.asarray(T_f_w), dtype=torch.float64, device=px.device)
    X, _ = _plane_points(T, cam, px[..., 0], px[..., 1])
    dx, dy = cam_undistort(cam, px[..., 0], px[..., 1])
    f = torch.stack([dx, dy, torch.ones_like(dx)], dim=-1)
    f = f / f.norm(dim=-1, keepdim=True)  # cam2world -> normalized()
    return f, X


@dataclass
class Sequence:
    cam: Camera
    T_f_w: np.ndarray            # [n,12] ground truth
    images: torch.Tensor         # uint8 [n,h,w]
    px: torch.Tensor             # [n,N,2] features detected in each frame
    f: torch.Tensor              # [n,N,3]
    pos: torch.Tensor            # [n,N,3]


def make_sequence(n_frames: int, n_feat: int, cam: Camera | None = None, seed: int = 12345,
                  device="cpu", margin: int = 28, cell: int = 32, subpixel: bool = True, **traj_kw) -> Sequence:
    cam = cam or Camera.vga()
    tex = make_texture(seed=seed)
    T = make_trajectory(n_frames, seed=seed, **traj_kw)
    images = render(tex, T, cam, device=device)
    px = select_features(images, n_feat, margin=margin, cell=cell)
    if subpixel:  # refined features are not integer pixels (matcher output is sub-pixel)
        g = torch.Generator().manual_seed(seed + 1)
        px = px + (torch.rand(px.shape, generator=g, dtype=torch.float64) - 0.5).to(px.device)
    f, pos = features_3d(T, cam, px)
    return Sequence(cam, T, images, px, f, pos)


# ------------------------------------------------------------------------------------------
# Scenes for the steps after sparse alignment (reprojection matching, pose refinement,
# depth filter): K keyframes + one current frame looking at the same plane, map points
# observed from several keyframes, seeds with an uncertain inverse depth.
# ------------------------------------------------------------------------------------------
@dataclass
class TrackScene:
    cam: Camera
    T_f_w: np.ndarray            # [K+1,12]; index K is the current frame (ground truth)
    images: torch.Tensor         # uint8 [K+1,h,w]
    cur: int                     # index of the current frame
    T_cur_prior: np.ndarray      # [12] perturbed pose of the current frame (sparse-align output stand-in)
    pt_pos: np.ndarray           # [P,3] map points (slightly off the true surface)
    obs: list                    # per point: list of (frame, px[2], f[3], level, type, grad[2]), newest first
    px_true: np.ndarray          # [P,2] exact projection of the true surface point in the current frame
    px_init: np.ndarray          # [P,2] projection of pt_pos with T_cur_prior (the matcher's start)


def _proj(T, cam: Camera, X):
    R = T[:9].reshape(3, 3)
    p = X @ R.T + T[9:]
    u, v = cam_distort(cam, p[:, 0] / p[:, 2], p[:, 1] / p[:, 2])
    return np.stack([u, v], axis=-1), p[:, 2]


def _bearing(cam: Camera, px):
    x, y = cam_undistort(cam, px[:, 0], px[:, 1])
    f = np.stack([x, y, np.ones(len(px))], axis=-1)
    return f / np.linalg.norm(f, axis=-1, keepdims=True)


def make_track_scene(n_kf: int = 4, n_feat: int = 100, cam: Camera | None = None, seed: int = 777, kf_gap: int = 6,
                     depth_noise: float = 0.01, prior_noise: float = 2e-3, edgelet_frac: float = 0.25,
                     device="cpu") -> TrackScene:
    from . import se3
    cam = cam or Camera.vga()
    rng = np.random.default_rng(seed)
    tex = make_texture(seed=12345)
    T_all = make_trajectory(n_kf * kf_gap + 1, seed=seed, max_step=0.02, max_rot_deg=0.5)
    idx = list(range(0, n_kf * kf_gap, kf_gap)) + [n_kf * kf_gap]
    T = T_all[idx]
    images = render(tex, T, cam, device=device)
    K = n_kf
    px_kf = select_features(images[:K], n_feat, margin=40, cell=32).cpu().numpy()
    px_kf = px_kf + rng.uniform(-0.5, 0.5, size=px_kf.shape)
    imgs = images.cpu().numpy().astype(np.float32)
    pts, obs_all, px_true = [], [], []
    for k in range(K):
        f_k, X_k = features_3d(T[k:k + 1], cam, torch.as_tensor(px_kf[k:k + 1]))
        f_k, X_k = f_k[0].numpy(), X_k[0].numpy()
        c_k = -T[k, :9].reshape(3, 3).T @ T[k, 9:]
        for i in range(n_feat):
            X = X_k[i]
            # the map point is the surface point moved along the viewing ray (depth error)
            ray = X - c_k
            Xn = c_k + ray * (1.0 + depth_noise * rng.normal())
            u, v = px_kf[k, i]
            iu, iv = int(round(u)), int(round(v))
            gx = imgs[k, iv, iu + 1] - imgs[k, iv, iu - 1]
            gy = imgs[k, iv + 1, iu] - imgs[k, iv - 1, iu]
            gn = math.hypot(gx, gy)
            is_edge = rng.uniform() < edgelet_frac and gn > 1e-3
            grad = (gx / gn, gy / gn) if is_edge else (1.0, 0.0)
            o = [(k, px_kf[k, i].copy(), f_k[i].copy(), 0, 1 if is_edge else 0, grad)]
            for k2 in range(K):
                if k2 == k:
                    continue
                p2, z2 = _proj(T[k2], cam, X[None])
                if z2[0] > 0 and 12 <= p2[0, 0] < cam.width - 12 and 12 <= p2[0, 1] < cam.height - 12:
                    g2 = (1.0, 0.0)
                    if is_edge:  # an edgelet stays an edgelet in every keyframe that observes it
                        ju, jv = int(round(p2[0, 0])), int(round(p2[0, 1]))
                        hx = imgs[k2, jv, ju + 1] - imgs[k2, jv, ju - 1]
                        hy = imgs[k2, jv + 1, ju] - imgs[k2, jv - 1, ju]
                        hn = math.hypot(hx, hy)
                        g2 = (hx / hn, hy / hn) if hn > 1e-3 else grad
                    o.append((k2, p2[0].copy(), _bearing(cam, p2)[0], 0, 1 if is_edge else 0, g2))
            pc, zc = _proj(T[K], cam, X[None])
            if zc[0] <= 0:
                continue
            order = rng.permutation(len(o))
            pts.append(Xn)
            obs_all.append([o[j] for j in order])
            px_true.append(pc[0])
    pt_pos = np.array(pts)
    px_true = np.array(px_true)
    T_prior = se3.mul(se3.exp(rng.normal(size=6) * prior_noise), T[K])
    px_init, _ = _proj(T_prior, cam, pt_pos)
    return TrackScene(cam, T, images, K, T_prior, pt_pos, obs_all, px_true, px_init)
