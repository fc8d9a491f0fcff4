"""Build helper: compile the gfx950 HIP library.

`build_hip()` is what `__graft_entry__.build()` runs: hipcc cross-compiles every
`.hip` translation unit under `rpg_svo_amd/csrc/` for gfx950 into
`rpg_svo_amd/lib/libsvo_hip.so` (in-tree, so it travels to the GPU box).
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rpg_svo_amd", "csrc")
LIBDIR = os.path.join(ROOT, "rpg_svo_amd", "lib")
LIB = os.path.join(LIBDIR, "libsvo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def build_hip(force: bool = False, verbose: bool = False, defines: list[str] | None = None) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    if not force and _newer(LIB, deps):
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           # SLP packing (v_pk_*_f32) costs register pairs + moves in the pixel loops
           "-fno-slp-vectorize",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *[f"-D{d}" for d in (defines or [])], *srcs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
