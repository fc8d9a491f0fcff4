"""The product's C++ host layer (rpg_svo_amd/host/svo_hip_device.*) exercised by a plain-g++ unit
test over the C ABI: per-geometry contexts, the pinned arena (one upload / download / in-place
fetch, overflow and foreign-pointer errors), the pyramid cache (hits, LRU eviction, per-lane
pinning, re-upload with K0 rebuilding the levels), workspace growth."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "test_device.cpp")
EXE = os.path.join(ROOT, "build", "test_device")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    host = os.path.join(ROOT, "rpg_svo_amd", "host")
    lib = os.path.join(ROOT, "rpg_svo_amd", "lib")
    subprocess.run(["g++", "-std=c++11", "-O1", "-g", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", host, SRC,
                    os.path.join(host, "svo_hip_device.cpp"), "-L", lib, "-lsvo_hip", f"-Wl,-rpath,{lib}", "-pthread",
                    "-o", EXE], check=True)


MOCK_EXE = os.path.join(ROOT, "build", "test_device_mock")


def test_host_logic_against_the_mock_abi():
    """The same unit test linked against tests/host/mock_svo_hip.cpp (host memory, synchronous streams) instead of
    libsvo_hip.so: arena addressing in all its modes, slot cache / LRU / pinning, prediction and deferred-call
    bookkeeping, lanes of several threads -- the host layer's logic, checked where there is no GPU."""
    os.makedirs(os.path.dirname(MOCK_EXE), exist_ok=True)
    host = os.path.join(ROOT, "rpg_svo_amd", "host")
    subprocess.run(["g++", "-std=c++11", "-O1", "-g", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", host, SRC,
                    os.path.join(host, "svo_hip_device.cpp"), os.path.join(ROOT, "tests", "host", "mock_svo_hip.cpp"), "-pthread",
                    "-o", MOCK_EXE], check=True)
    for mode in ("hybrid", "mirrored", "mapped"):
        r = subprocess.run([MOCK_EXE], capture_output=True, text=True, timeout=120, env=dict(os.environ, SVO_HIP_ARENA=mode))
        assert r.returncode == 0 and "ALL OK" in r.stdout, (mode, r.stdout[-500:], r.stderr[-2000:])


def test_host_layer_builds_with_plain_gxx(hip_lib):
    """No HIP, Eigen or reference headers needed: this is what lets libsvo keep building with g++."""
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_device_arena_and_pyramid_cache(hip_lib, gpu_device):
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stderr
    assert "ALL OK" in r.stdout


EXAMPLE = os.path.join(ROOT, "build", "sparse_align_batch")


EXAMPLE_MAP = os.path.join(ROOT, "build", "reproject_map")


def _build_example(src="sparse_align_batch.cpp", exe=None):
    exe = exe or EXAMPLE
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    lib = os.path.join(ROOT, "rpg_svo_amd", "lib")
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", src), "-L", lib, "-lsvo_hip", f"-Wl,-rpath,{lib}",
                    "-o", exe], check=True)


def test_cxx_example_builds(hip_lib):
    _build_example()
    _build_example("reproject_map.cpp", EXAMPLE_MAP)


@pytest.mark.gpu
def test_cxx_map_mirror_example(hip_lib, gpu_device):
    """svo_hip_reproject_map from plain C++ (examples/reproject_map.cpp): a map filled through a patch, one frame
    reprojected, the visit list compared with a host walk written the reference's way."""
    _build_example("reproject_map.cpp", EXAMPLE_MAP)
    r = subprocess.run([EXAMPLE_MAP], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "OK" in r.stdout and "equals the host walk" in r.stdout


@pytest.mark.gpu
def test_cxx_example_recovers_the_motion(hip_lib, gpu_device):
    """The C ABI driven from plain C++ end to end: upload, K0, K1 on 1024 problems, poses back."""
    _build_example()
    r = subprocess.run([EXAMPLE, "1024"], capture_output=True, text=True, timeout=120)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "OK" in r.stdout
