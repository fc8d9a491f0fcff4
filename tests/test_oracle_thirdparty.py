"""Independent numpy re-derivations of the third-party arithmetic the oracle restates from published
sources (rpg_vikit, fast, boost::math are un-vendored and un-pinned): the depth-filter
closed forms, the FAST-10 detector with its score / non-max rules, and vk::shiTomasiScore.  CPU only."""
import math

import numpy as np

from oracle import pytrack


def test_update_seed_matches_float64_derivation(oracle):
    """Vogiatzis & Hernandez moment matching (depth_filter.cpp:309-332) re-derived in float64."""
    orc = pytrack.Track("orc")
    rng = np.random.default_rng(0)
    for _ in range(200):
        s = orc.seed_init(rng.uniform(0.5, 5.0), rng.uniform(0.2, 0.45))
        s.a, s.b = np.float32(rng.uniform(5, 40)), np.float32(rng.uniform(5, 40))
        s.mu = np.float32(s.mu * rng.uniform(0.8, 1.2))
        x = float(s.mu) * rng.uniform(0.9, 1.1)
        tau2 = (float(s.mu) * rng.uniform(0.005, 0.05)) ** 2
        a, b, mu, s2, zr = float(s.a), float(s.b), float(s.mu), float(s.sigma2), float(s.z_range)
        norm_scale = math.sqrt(s2 + tau2)
        pdf = math.exp(-0.5 * ((x - mu) / norm_scale) ** 2) / (norm_scale * math.sqrt(2 * math.pi))
        ss2 = 1.0 / (1.0 / s2 + 1.0 / tau2)
        m = ss2 * (mu / s2 + x / tau2)
        C1, C2 = a / (a + b) * pdf, b / (a + b) / zr
        C1, C2 = C1 / (C1 + C2), C2 / (C1 + C2)
        f = C1 * (a + 1) / (a + b + 1) + C2 * a / (a + b + 1)
        e = C1 * (a + 1) * (a + 2) / ((a + b + 1) * (a + b + 2)) + C2 * a * (a + 1) / ((a + b + 1) * (a + b + 2))
        mu_new = C1 * m + C2 * mu
        s2_new = C1 * (ss2 + m * m) + C2 * (s2 + mu * mu) - mu_new * mu_new
        a_new = (e - f) / (f - e / f)
        b_new = a_new * (1 - f) / f
        o = orc.update_seed(x, tau2, s)
        assert np.isclose(o.mu, mu_new, rtol=2e-5)
        assert abs(o.sigma2 - s2_new) <= 2e-3 * s2_new + 1e-6 * mu * mu      # float cancellation in the reference
        assert np.isclose(o.a, a_new, rtol=5e-2) and np.isclose(o.b, b_new, rtol=5e-2)


def test_compute_tau_is_the_law_of_sines(oracle):
    """depth_filter.cpp:334-350 with the reference's PI = 3.14159265 (global.h:78)."""
    orc = pytrack.Track("orc")
    rng = np.random.default_rng(1)
    for _ in range(100):
        t = rng.normal(size=3) * 0.2
        f = rng.normal(size=3); f[2] = abs(f[2]) + 1.0; f /= np.linalg.norm(f)
        z = rng.uniform(0.5, 5.0)
        ang = 2 * math.atan(1.0 / (2 * 315.5))
        T = np.concatenate([np.eye(3).reshape(9), t])
        a = f * z - t
        alpha = math.acos(f @ t / np.linalg.norm(t))
        beta = math.acos(a @ (-t) / (np.linalg.norm(t) * np.linalg.norm(a)))
        gamma_plus = 3.14159265 - alpha - (beta + ang)
        z_plus = np.linalg.norm(t) * math.sin(beta + ang) / math.sin(gamma_plus)
        assert np.isclose(orc.compute_tau(T, f, z, ang), z_plus - z, rtol=1e-10, atol=1e-14)


RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _is_corner(img, x, y, b):
    p = int(img[y, x])
    ring = [int(img[y + dy, x + dx]) for dx, dy in RING]
    for sign in (1, -1):
        flags = [(v > p + b) if sign == 1 else (v < p - b) for v in ring]
        run = best = 0
        for f in flags + flags:  # circular
            run = run + 1 if f else 0
            best = max(best, run)
        if best >= 10:
            return True
    return False


def test_fast10_score_nonmax_and_shi_tomasi_against_brute_force(oracle):
    rng = np.random.default_rng(3)
    h, w = 40, 56
    img = rng.integers(0, 256, size=(h, w)).astype(np.uint8)
    img[10:30, 12:40] = (img[10:30, 12:40] // 4 + 150).astype(np.uint8)   # a brighter block: real corners and edges
    pyr = oracle.create_img_pyramid(img, 1)
    # cell size 1: every pixel is its own grid cell, so the grid output IS the list of surviving corners
    xy, lvl, sc, n = pytrack.fast_detect_grid(pyr, 1, 1, w, h, None, 20, 0.0)
    got = {(int(x), int(y)): float(s) for (x, y), s, l in zip(xy, sc, lvl) if l >= 0}
    corners = {}
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if _is_corner(img, x, y, 20):
                lo, hi = 20, 255   # fast_corner_score_10: largest threshold that still passes
                t = (lo + hi) // 2
                while True:
                    if _is_corner(img, x, y, t):
                        lo = t
                    else:
                        hi = t
                    if lo == hi - 1 or lo == hi:
                        break
                    t = (lo + hi) // 2
                corners[(x, y)] = lo
    assert len(corners) > 30
    expect = {}
    I = img.astype(np.float64)
    for (x, y), s in corners.items():
        if any(corners.get((x + dx, y + dy), -1) >= s for dx in (-1, 0, 1) for dy in (-1, 0, 1) if (dx, dy) != (0, 0)):
            continue  # fast_nonmax_3x3: a neighbouring corner with score >= own suppresses
        if x - 4 < 1 or x + 4 >= w - 1 or y - 4 < 1 or y + 4 >= h - 1:
            continue  # vk::shiTomasiScore returns 0 next to the border: never above the threshold
        ys, xs = slice(y - 4, y + 4), slice(x - 4, x + 4)
        dx = I[ys, x - 3:x + 5] - I[ys, x - 5:x + 3]
        dy = I[y - 3:y + 5, xs] - I[y - 5:y + 3, xs]
        dxx, dyy, dxy = (dx * dx).sum() / 128, (dy * dy).sum() / 128, (dx * dy).sum() / 128
        lam = 0.5 * (dxx + dyy - math.sqrt((dxx + dyy) ** 2 - 4 * (dxx * dyy - dxy * dxy)))
        if lam > 0.0:
            expect[(x, y)] = lam
    assert set(got) == set(expect)
    for k, v in expect.items():
        assert np.isclose(got[k], v, rtol=1e-4), (k, got[k], v)
