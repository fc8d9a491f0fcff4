import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (checker only)."""
    from oracle import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def hip_lib():
    """libsvo_hip.so; built on demand where hipcc exists (cross-compiles without a GPU)."""
    from rpg_svo_amd import build, capi
    if not os.path.exists(capi.lib_path()):
        build.build_hip()
    return capi.load()


@pytest.fixture(scope="session")
def gpu_device(hip_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a visible HIP device")
    return torch.device("cuda:0")


@pytest.fixture(scope="module", params=["orc", "ref"])
def checker(request, oracle):
    """Which CPU checker a GPU parity test compares against: "orc" = oracle/libsvo_oracle.so (the C
    restatement, always present), "ref" = oracle/_ref/libsvo_ref.so (the reference's own translation
    units; built where the reference checkout exists and shipped to the GPU box prebuilt).  The
    "ref" leg closes the chain HIP <-> reference on the GPU box itself."""
    if request.param == "ref":
        from oracle import pytrack
        if not pytrack.ref_available():
            pytest.skip("oracle/_ref/libsvo_ref.so not built (needs the reference checkout at build time)")
    return request.param
