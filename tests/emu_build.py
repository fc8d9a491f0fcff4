"""TEST INFRASTRUCTURE.  The host-emulated build of the C-ABI library: the .hip translation units of rpg_svo_amd/csrc compiled by
ROCm's clang++ as plain C++ through tests/host/hip_emu.h (work-items as fibers, barriers, LDS, cross-lane rendezvous) and
linked into build/libsvo_hip_emulated[_<defines>].so.  The same entry points as libsvo_hip.so, on host memory; no timing."""
import ctypes as C
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ("common", "pyramid", "map_mirror", "matcher", "feature_align", "depth_filter", "sparse_align", "sparse_align_wave",
         "pose_optimizer_wave", "pose_optimizer", "point_optimizer", "fast_detect")


# Compile-time variants of the library the emulated parity tests run on besides the default build (round 4 queued ten of
# them for timing; round 5 measured them on the GPU, the winners became the only code and the losers were deleted:
# profiles/r05a_queue_drain.txt).  Tests take their build from this list: a new opt-in flag is one more tuple here.
# Round 6: ("ALIGN_WAVE_MAX_M_VALUE=0",) -- no batch is small enough for the wave-per-trial alignment, so the depth filter's
# small test batches take the lane-per-trial kernel WITH the seed_finish epilogue (csrc/seed_finish.h), the path of replay
# batches beyond 8192 seeds.
BUILDS = [(), ("ALIGN_WAVE_MAX_M_VALUE=0",)]
BUILD_IDS = ["default", "lane_kernel_with_finish"]


def sanitizer():
    """SVO_EMU_SANITIZE=address|thread: the emulated library instrumented by that sanitizer (the process must have the
    runtime preloaded: scripts/emu_sanitize.sh)."""
    return os.environ.get("SVO_EMU_SANITIZE", "")


def sanitizer_runtime(kind):
    cxx = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang++")
    name = {"address": "asan", "thread": "tsan", "undefined": "ubsan_standalone"}[kind]
    out = subprocess.run([cxx, f"-print-file-name=libclang_rt.{name}-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def build_race_probe():
    """tests/host/emu_race_probe.cpp (a kernel with and without the barrier it needs) with the sanitizer of SVO_EMU_SANITIZE."""
    san = sanitizer()
    cxx = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang++")
    lib_path = os.path.join(ROOT, "build", "emu", f"libemu_race_probe_{san or 'plain'}.so")
    src = os.path.join(ROOT, "tests", "host", "emu_race_probe.cpp")
    hdr = os.path.join(ROOT, "tests", "host", "hip_emu.h")
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    if not os.path.exists(lib_path) or os.path.getmtime(lib_path) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        flags = [f"-fsanitize={san}", "-shared-libsan", "-fno-omit-frame-pointer", "-g"] if san else []
        subprocess.run([cxx, "-std=c++17", "-O1", "-fPIC", "-shared", *flags, "-I", os.path.join(ROOT, "tests", "host"), src, "-o", lib_path],
                       check=True)
    lib = C.CDLL(lib_path)
    lib.probe_neighbour_sum.restype = C.c_int
    lib.probe_neighbour_sum.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    return lib


def build_emulated(defines=()):
    # SVO_EMU_EXTRA_DEFINES="A B": added to every emulated build of the run (e.g. the whole round-5 queue on the default tests)
    from rpg_svo_amd.build import DEFAULT_DEFINES   # (the default library's own flags: the emulated default build has them too)
    defines = tuple(dict.fromkeys(tuple(DEFAULT_DEFINES) + tuple(defines) + tuple(os.environ.get("SVO_EMU_EXTRA_DEFINES", "").split())))
    tag = "".join("_" + d.replace("=", "-") for d in defines)
    if len(tag) > 80:
        import hashlib
        tag = "_set" + hashlib.sha1(tag.encode()).hexdigest()[:10]
    san = sanitizer()
    san_flags = [f"-fsanitize={san}", "-shared-libsan", "-fno-omit-frame-pointer", "-g"] if san else []
    if san == "undefined":
        # + float-cast-overflow (a float -> int conversion out of range: the kernels guard theirs to reproduce cvttss2si).
        # float-divide-by-zero stays off: IEEE defines it and the kernels rely on it where the reference does (1 / tau2 with
        # tau2 = 0, the inverse of a singular H, chi2 / n_meas with nothing tracked: six sites, all the reference's own)
        san_flags += ["-fsanitize=float-cast-overflow", "-fno-sanitize=vptr,function"]
    if san:
        tag += "_" + san
    lib_path = os.path.join(ROOT, "build", "emu", f"libsvo_hip_emulated{tag}.so")
    objdir = os.path.join(ROOT, "build", "emu", f"obj{tag}")
    os.makedirs(objdir, exist_ok=True)
    csrc = os.path.join(ROOT, "rpg_svo_amd", "csrc")
    cxx = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang++")
    if not os.path.exists(cxx):
        pytest.skip("no ROCm clang++ to compile the kernels for the host")
    deps = glob.glob(os.path.join(csrc, "*.h")) + glob.glob(os.path.join(csrc, "*.hip")) + \
        [os.path.join(ROOT, "include", "svo_hip.h"), os.path.join(ROOT, "tests", "host", "hip_emu.h")]
    newest = max(os.path.getmtime(d) for d in deps)
    # (a change of flags rebuilds too: the object directory remembers what it was compiled with)
    stamp, flags_now = os.path.join(objdir, "flags.txt"), " ".join([*san_flags, *defines, "-O1"])
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        for o in glob.glob(os.path.join(objdir, "*.o")):
            os.remove(o)
        open(stamp, "w").write(flags_now)
    objs, todo = [], []
    for u in UNITS:
        src = os.path.join(ROOT, "tests", "host", f"emu_tu_{u}.cpp")
        obj = os.path.join(objdir, f"{u}.o")
        objs.append(obj)
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(newest, os.path.getmtime(src)):
            todo.append([cxx, "-std=c++17", "-O1", "-ffp-contract=off", "-fno-math-errno", "-fPIC", "-c", "-Wall", "-Wno-unknown-pragmas",
                         "-Wno-pass-failed", "-Wno-unused-function", "-Wno-unused-variable", *san_flags,
                         *[f"-D{d}" for d in defines], "-I", os.path.join(ROOT, "include"), "-I", csrc,
                         "-I", os.path.join(ROOT, "tests", "host"), src, "-o", obj])
    if todo:  # the translation units in parallel: a cold build of one variant set takes about as long as its slowest unit
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(8, len(todo))) as ex:
            list(ex.map(lambda cmd: subprocess.run(cmd, check=True), todo))
    if not os.path.exists(lib_path) or any(os.path.getmtime(o) > os.path.getmtime(lib_path) for o in objs):
        subprocess.run([cxx, "-shared", *san_flags, "-o", lib_path, *objs], check=True)
    lib = C.CDLL(lib_path)
    return lib
