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


# Compile-time flags of the DEFAULT library (the test suite compiles its CPU checks of the kernels with the same list).
# Empty since round 5: the variants that won their A/B on the GPU became the only code, the others were deleted
# (profiles/r05a_queue_drain.txt).  A change of this list, of the compiler or of its flags invalidates build/obj*/ (the
# stamp file below), not only a newer source.
DEFAULT_DEFINES: list[str] = []
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         # SLP packing (v_pk_*_f32) costs register pairs + moves in the pixel loops
         "-fno-slp-vectorize"]


def _compile_one(src: str, obj: str, defines: list[str], verbose: bool) -> None:
    defines = list(dict.fromkeys([*DEFAULT_DEFINES, *defines]))
    cmd = [HIPCC, *FLAGS, "-c", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *[f"-D{d}" for d in defines], src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)


def build_hip(force: bool = False, verbose: bool = False, defines: list[str] | None = None, out: str | None = None) -> str:
    """One object per translation unit (rebuilt only when it or a header changed), compiled in
    parallel, then linked into libsvo_hip.so.  With `defines` (a WHOLE-LIBRARY variant for A/B timing, e.g.
    ["SIA_F64_PARTIALS"]) the objects and the library go to build/obj_<defines>/ and
    build/variants/libsvo_hip_<defines>.so (or `out`); load it with SVO_HIP_LIB=<that file>."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    objdir = os.path.join(ROOT, "build", "obj" + ("_" + "_".join(defines) if defines else ""))
    lib = out or (os.path.join(ROOT, "build", "variants", "libsvo_hip_" + "_".join(defines) + ".so") if defines else LIB)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    # what the objects of this directory were compiled with: a different compiler, flag or define list rebuilds them all
    stamp_file = os.path.join(objdir, "flags.stamp")
    stamp = " ".join([HIPCC, *FLAGS, *dict.fromkeys([*DEFAULT_DEFINES, *(defines or [])])])
    if not os.path.exists(stamp_file) or open(stamp_file).read() != stamp:
        force = True
    todo, objs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not _newer(obj, [src] + hdrs):
            todo.append((src, obj))
    if todo:
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(lambda so: _compile_one(so[0], so[1], defines or [], verbose), todo))
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    if todo or force or not _newer(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-lhipsolver", "-o", lib]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    # python -m rpg_svo_amd.build [--force] [--out=FILE] [-DFOO ...]: the in-tree library, or a whole-library variant
    # (--out=<file>: where a variant goes instead of build/variants/libsvo_hip_<defines>.so)
    print(build_hip(force="--force" in sys.argv, verbose=True, defines=[a[2:] for a in sys.argv[1:] if a.startswith("-D")] or None,
                    out=next((os.path.abspath(a[6:]) for a in sys.argv[1:] if a.startswith("--out=")), None)))
