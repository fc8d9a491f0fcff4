"""CPU-only: the C-ABI library loads and exports every symbol include/svo_hip.h
declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "svo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svo_hip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(hip_lib):
    names = declared_functions()
    assert len(names) >= 20
    raw = ctypes.CDLL(os.path.join(ROOT, "rpg_svo_amd", "lib", "libsvo_hip.so"))
    for n in names:
        assert hasattr(raw, n), f"{n} declared in svo_hip.h but not exported"


def test_binding_covers_header(hip_lib):
    from rpg_svo_amd import capi
    assert sorted(capi.PROTOTYPES) == declared_functions()


def test_layout_is_host_only(hip_lib):
    from rpg_svo_amd import capi
    L = capi.pyr_layout(640, 480, 4)
    assert list(L.w[:4]) == [640, 320, 160, 80] and list(L.h[:4]) == [480, 240, 120, 60]
    assert L.tile == capi.PYR_TILED and all(p % 16 == 0 and p >= w for p, w in zip(L.pitch[:4], L.w[:4]))
    assert all(o % 128 == 0 for o in L.offset[:4])  # a 16 x 8 tile is one 128-byte line
    assert L.slot_bytes % 4096 == 0 and L.slot_bytes >= 408000
    L5 = capi.pyr_layout(752, 480, 5)
    assert list(L5.w[:5]) == [752, 376, 188, 94, 47] and list(L5.h[:5]) == [480, 240, 120, 60, 30]
    assert capi.pyr_store_bytes(L, 3) == 3 * L.slot_bytes + capi.STORE_TAIL_PAD


def test_tiled_addressing_is_a_bijection(hip_lib):
    """Every pixel of every level has its own byte inside the slot, levels do not overlap, and the bytes of a
    16 x 8 pixel tile form one 128-byte line (the host mirror of csrc/pyr_addr.h)."""
    import numpy as np
    from rpg_svo_amd import capi
    for (w, h, n) in ((640, 480, 4), (752, 480, 5), (100, 37, 3)):
        L = capi.pyr_layout(w, h, n)
        seen = np.zeros(L.slot_bytes, dtype=np.int32)
        for l in range(n):
            ys, xs = np.mgrid[0:L.h[l], 0:L.w[l]]
            off = capi.pyr_px_offset(L, l, xs, ys)
            assert off.min() >= L.offset[l] and off.max() < L.slot_bytes
            np.add.at(seen, off.ravel(), 1)
            if L.w[l] >= 32 and L.h[l] >= 16:
                tile = capi.pyr_px_offset(L, l, xs[8:16, 16:32], ys[8:16, 16:32])
                assert tile.max() - tile.min() == 127 and tile.min() % 128 == 0
        assert seen.max() == 1


def test_error_strings(hip_lib):
    assert hip_lib.svo_hip_strerror(0) == b"ok"
    assert b"invalid" in hip_lib.svo_hip_strerror(-1)
    from rpg_svo_amd import capi
    bad = capi.PyrLayout()
    assert hip_lib.svo_hip_pyr_layout_init(640, 480, 99, ctypes.byref(bad)) == -1


def test_product_never_imports_oracle():
    """The shipped package must not reference the checker."""
    pkg = os.path.join(ROOT, "rpg_svo_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in txt and "svo_oracle" not in txt and "orc_" not in txt, f


def test_product_has_no_cpu_path():
    """The kernels compile for the CPU only inside the test suite (tests/emu_build.py defines SVO_HOST_MATH_TEST / SVO_HIP_EMU
    through tests/host/hip_emu.h): nothing in the package defines those macros, loads the emulated library or mentions the
    emulator, the build recipe passes no such define, and the in-tree library carries no emulator symbol."""
    import subprocess
    pkg = os.path.join(ROOT, "rpg_svo_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "#define SVO_HOST_MATH_TEST" not in txt and "#define SVO_HIP_EMU" not in txt, f
                assert "libsvo_hip_emulated" not in txt and "emu_build" not in txt and '"hip_emu.h"' not in txt, f
    for f in ("bench.py", "__graft_entry__.py"):
        txt = open(os.path.join(ROOT, f)).read()
        assert "libsvo_hip_emulated" not in txt and "emu_build" not in txt and "SVO_HOST_MATH_TEST" not in txt, f
    lib = os.path.join(pkg, "lib", "libsvo_hip.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", "-C", lib], capture_output=True, text=True).stdout
        assert "svo_emu" not in syms
