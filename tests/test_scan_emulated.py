"""The epipolar ZMSSD scan of epi_scan_kernel on the CPU, through a small SIMT emulation.

rpg_svo_amd/csrc/epi_scan.h holds the scan of one seed by a group of eight lanes -- cross-lane moves (DPP, shuffles), an
LDS box handed over inside the wave.  tests/host/hip_emu.h runs that code with one fiber per lane; here it is checked
against a plain sequential numpy scan of the same seeds."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from helpers import FUZZ, fuzz_rng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "scan_emulated.cpp")
LIB = os.path.join(ROOT, "build", "emu", "libscan_emulated.so")
ZMSSD_THRESHOLD = 2000 * 64


def _p(a, t=None):
    return a.ctypes.data_as(C.POINTER(t or {np.dtype("float64"): C.c_double, np.dtype("int32"): C.c_int32, np.dtype("uint8"): C.c_uint8,
                                             np.dtype("int64"): C.c_longlong}[a.dtype]))


@pytest.fixture(scope="module")
def emu():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    csrc = os.path.join(ROOT, "rpg_svo_amd", "csrc")
    deps = [SRC, os.path.join(ROOT, "tests", "host", "hip_emu.h")] + [os.path.join(csrc, h) for h in
                                                                       ("epi_scan.h", "track_math.h", "device_math.h", "pyr_addr.h", "matcher_device.h",
                                                                        "warp_group.h", "warp_sample.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        cxx = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang++")
        if not os.path.exists(cxx):
            pytest.skip("no ROCm clang++ to compile the kernels' headers for the host")
        subprocess.run([cxx, "-std=c++17", "-O2", "-ffp-contract=off", "-fno-math-errno", "-fPIC", "-shared", "-pthread", "-Wall",
                        "-Wno-unknown-pragmas", "-Wno-pass-failed", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                        "-I", os.path.join(ROOT, "tests", "host"), SRC, "-o", LIB], check=True)
    return C.CDLL(LIB)


def _texture(rng, h, w):
    img = rng.uniform(0, 255, (h // 4 + 3, w // 4 + 3))
    ys, xs = np.arange(h) / 4.0, np.arange(w) / 4.0
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    a = img[y0][:, x0] * (1 - fy) * (1 - fx) + img[y0][:, x0 + 1] * (1 - fy) * fx
    b = img[y0 + 1][:, x0] * fy * (1 - fx) + img[y0 + 1][:, x0 + 1] * fy * fx
    return np.clip(a + b + rng.normal(0, 2.0, (h, w)), 0, 255).astype(np.uint8)


def _store(levels):
    """one slot of the tiled store (csrc/pyr_addr.h): 16-byte x 8-row tiles, levels back to back"""
    offs, ws, hs, ps, parts, off = [], [], [], [], [], 0
    for img in levels:
        h, w = img.shape
        pitch = (w + 15) & ~15
        nbytes = pitch * ((h + 7) & ~7)
        buf = np.zeros(nbytes, np.uint8)
        ys, xs = np.mgrid[0:h, 0:w]
        buf[((ys >> 3) * 8 * pitch + (ys & 7) * 16 + (xs >> 4) * 128 + (xs & 15)).ravel()] = img.ravel()
        offs.append(off); ws.append(w); hs.append(h); ps.append(pitch); parts.append(buf)
        off += nbytes
    store = np.concatenate(parts + [np.zeros(256, np.uint8)])  # (+ slack: window rows are read as 12-byte runs)
    return store, off, np.array(offs, np.int64), np.array(ws, np.int32), np.array(hs, np.int32), np.array(ps, np.int32)


def _seeds(rng, levels, cam, S):
    fx, fy, cx, cy = cam
    sl = rng.integers(0, len(levels), S).astype(np.int32)
    n_steps = np.empty(S, np.int32)
    B, step = np.empty((S, 2)), np.empty((S, 2))
    pwb = np.empty((S, 100), np.uint8)
    truth = np.full((S, 2), -1, np.int64)  # level pixel the template was cut at (when it was cut on the line)
    for s in range(S):
        img = levels[sl[s]]
        h, w = img.shape
        scale = 1 << sl[s]
        n = int(rng.choice([1, 2, 5, 9, 17, 40, 90, 160]))
        ang = rng.uniform(0, 2 * np.pi)
        d = 0.7 * np.array([np.cos(ang), np.sin(ang)])  # level pixels per step (matcher.cpp:264: 0.7 px)
        # first position anywhere (also outside: those positions are skipped), the line mostly inside the level
        p0 = np.array([rng.uniform(6, w - 6), rng.uniform(6, h - 6)]) - d * n * rng.uniform(0, 1)
        k = int(rng.integers(0, n + 1))
        pk = np.floor(p0 + k * d + 0.5).astype(int)
        if rng.uniform() < 0.7 and 8 <= pk[0] < w - 8 and 8 <= pk[1] < h - 8:
            pwb[s] = img[pk[1] - 5:pk[1] + 5, pk[0] - 5:pk[0] + 5].ravel()  # interior 8 x 8 = window [px-4, px+3]^2
            truth[s] = pk
        else:
            pwb[s] = rng.integers(0, 256, 100)
        n_steps[s] = n
        # the scan starts at B - step and adds step before every position but the first (matcher.cpp:264-268)
        P0 = p0 * scale  # level-0 pixels
        st = d * scale
        B[s] = ((P0 + st) - np.array([cx, cy])) / np.array([fx, fy])
        step[s] = st / np.array([fx, fy])
    return sl, n_steps, B, step, pwb, truth


ATLAS_PITCH = 16  # the templates' atlas (the last level of the test's store): one 10 x 10 patch per 16 x 16 cell


def _atlas(pwb):
    """The scan kernel warps its template itself (round 6).  The test hands it its templates as an image: an extra level
    of the store with patch s in cell s; the identity warp A_ref_cur = 2^-search_level * I at the cell's centre then
    reproduces the patch byte for byte (integer sample positions: the bilinear weights are 1, 0, 0, 0)."""
    S = len(pwb)
    n_col = 64
    img = np.zeros((ATLAS_PITCH * ((S + n_col - 1) // n_col) + 8, ATLAS_PITCH * n_col + 8), np.uint8)
    centre = np.zeros((S, 2), np.float32)
    for s in range(S):
        cx, cy = ATLAS_PITCH * (s % n_col) + 8, ATLAS_PITCH * (s // n_col) + 8
        img[cy - 5:cy + 5, cx - 5:cx + 5] = pwb[s].reshape(10, 10)
        centre[s] = (cx, cy)
    return img, centre


def _run(emu, form, S, store, slot_bytes, offs, ws, hs, ps, cam, size, sl, n_steps, B, step, centre, subpix):
    out = dict(uv_best=np.zeros((S, 2)), px_cur=np.zeros((S, 2)), px_scaled=np.zeros((S, 2)), align_active=np.zeros(S, np.uint8),
               accepted_raw=np.zeros(S, np.uint8), status=np.zeros(S, np.int32), pwb=np.full((S, 100), 7, np.uint8))
    cur_slot = np.zeros(S, np.int32)
    cam_a, B, step = np.array(cam, np.float64), np.ascontiguousarray(B), np.ascontiguousarray(step)
    A = np.zeros((S, 4), np.float32)
    A[:, 0] = A[:, 3] = 1.0 / (1 << sl)
    ref_slot, ref_level, mode = np.zeros(S, np.int32), np.full(S, len(ws) - 1, np.int32), np.full(S, 2, np.int32)  # MODE_SCAN
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    emu.scan_emulated(C.c_int(form), C.c_int(S), _p(store), C.c_longlong(slot_bytes), C.c_int(len(ws)), _p(offs), _p(ws), _p(hs), _p(ps),
                      _p(cam_a), C.c_int(size[0]), C.c_int(size[1]), C.c_int(subpix), _p(sl), _p(cur_slot), _p(n_steps),
                      _p(B), _p(step), _p(out["pwb"]), fp(A), fp(np.ascontiguousarray(centre)), _p(ref_slot), _p(ref_level), _p(mode),
                      _p(out["uv_best"]),
                      _p(out["px_cur"]), _p(out["px_scaled"]), _p(out["align_active"]), _p(out["accepted_raw"]), _p(out["status"]))
    return out


def _numpy_scan(levels, cam, sl, n_steps, B, step, pwb):
    """matcher.cpp:248-291 for one seed: sequential walk, skip repeated pixels and windows outside the level, first minimum"""
    fx, fy, cx, cy = cam
    img = levels[sl].astype(np.int64)
    h, w = img.shape
    A = pwb.reshape(10, 10)[1:9, 1:9].astype(np.int64)
    sumA, sumAA = A.sum(), (A * A).sum()
    uv = B - step
    best, best_uv, last = ZMSSD_THRESHOLD, None, (0, 0)
    for i in range(n_steps + 1):
        px = np.array([fx * uv[0] + cx, fy * uv[1] + cy])
        pxi = (int(px[0] / (1 << sl) + 0.5), int(px[1] / (1 << sl) + 0.5))
        if pxi != last:
            last = pxi
            if 8 <= pxi[0] < w - 8 and 8 <= pxi[1] < h - 8:
                Bw = img[pxi[1] - 4:pxi[1] + 4, pxi[0] - 4:pxi[0] + 4]
                sB, sBB, sAB = Bw.sum(), (Bw * Bw).sum(), (A * Bw).sum()
                z = sumAA - 2 * sAB + sBB - int((sumA * sumA - 2 * sumA * sB + sB * sB) / 64)
                if z < best:
                    best, best_uv = z, uv.copy()
        uv = uv + step
    return best, best_uv


def test_group_scan_is_the_sequential_scan(emu):
    """320 seeds over three levels, 2 to 161 positions per seed, with and without sub-pixel refinement: the eight-lane scan
    finds what a sequential numpy scan finds -- the same verdict and, to the bit, the same uv_best."""
    rng = fuzz_rng(51)
    base = _texture(rng, 240, 320)
    levels = [base]
    for _ in range(2):
        p = levels[-1].astype(np.uint16)
        levels.append(((p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]) // 4).astype(np.uint8))
    cam, size = (300.0, 300.0, 160.0, 120.0), (320, 240)
    S = 320
    sl, n_steps, B, step, pwb, truth = _seeds(rng, levels, cam, S)
    atlas, centre = _atlas(pwb)
    store, slot_bytes, offs, ws, hs, ps = _store(levels + [atlas])
    n_found = 0
    for subpix in (1, 0):
        a = _run(emu, 0, S, store, slot_bytes, offs, ws, hs, ps, cam, size, sl, n_steps, B, step, centre, subpix)
        matched = (a["align_active"] != 0) | (a["accepted_raw"] != 0)
        assert np.array_equal(matched, a["status"] == 0) and matched.sum() > 150 and (~matched).sum() > 20
        for s in range(S):  # the default form against the sequential scan
            best, best_uv = _numpy_scan(levels, cam, int(sl[s]), int(n_steps[s]), B[s], step[s], pwb[s])
            assert matched[s] == (best < ZMSSD_THRESHOLD), (s, best)
            # the warped patch reaches memory for a seed that goes on to the sub-pixel alignment, and only for it
            assert np.array_equal(a["pwb"][s], pwb[s] if matched[s] and subpix else np.full(100, 7, np.uint8)), s
            if matched[s]:
                assert np.array_equal(a["uv_best"][s], best_uv), (s, a["uv_best"][s], best_uv)
                n_found += int(truth[s][0] >= 0 and np.all(np.floor(a["px_scaled"][s] + 0.5).astype(int) == truth[s]))
    assert n_found > 150, n_found  # (templates cut on the line are found where they were cut)
