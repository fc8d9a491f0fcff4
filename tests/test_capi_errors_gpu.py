"""Error behaviour of the C ABI: bad arguments are rejected with SVO_HIP_EINVAL / ERANGE before any
launch, never with a crash (the host wrappers turn the codes into exceptions)."""
import ctypes as C

import pytest
import torch

from rpg_svo_amd import capi

pytestmark = pytest.mark.gpu
EINVAL, ERANGE = -1, -2


def test_bad_arguments_are_rejected(hip_lib, gpu_device):
    lib = hip_lib
    L = capi.pyr_layout(640, 480, 4)
    buf = torch.zeros(capi.pyr_store_bytes(L, 2), dtype=torch.uint8, device=gpu_device)
    p = buf.data_ptr()
    bad = capi.PyrLayout()
    # pyramid
    assert lib.svo_hip_pyramid_build(C.byref(bad), p, 0, 1, 2, None) == EINVAL
    assert lib.svo_hip_pyramid_build(C.byref(L), None, 0, 1, 2, None) == EINVAL
    assert lib.svo_hip_pyramid_build(C.byref(L), p, 0, 1, 7, None) == EINVAL           # unknown half-sample flavour
    assert lib.svo_hip_pyramid_build_from_images(C.byref(L), p, 0, 1, p, 640 * 480, 600, 2, None) == EINVAL   # stride < width
    assert lib.svo_hip_pyramid_build_tiled(C.byref(L), p, 0, 1, None, 0, 0, 2, 300, None) == EINVAL        # unknown tile
    # feature alignment / matcher / depth filter
    assert lib.svo_hip_align_batch(C.byref(L), p, -1, None, None, None, None, None, 10, None, None, None, None) == EINVAL
    assert lib.svo_hip_align_batch(C.byref(L), p, 0, None, None, None, None, None, 10, None, None, None, None) == 0  # empty batch
    cam = capi.Camera(400, 400, 320, 240, 640, 480)
    fr = capi.Frames(0, 0, None, None)
    ft = capi.Features(None, None, None, None, None, None)
    assert lib.svo_hip_find_match_direct(C.byref(L), p, C.byref(cam), C.byref(fr), 4, None, None, None, C.byref(ft), 3, 10,
                                         None, None, None, None, None, None, None, 0, None) < 0
    sd = capi.Seeds(None, None, None, None, None, None)
    opt = capi.DepthFilterOptions(3, 0, 200.0, 0, 10, 1000, 1, 1, 3, 0.7)
    assert lib.svo_hip_update_seeds(C.byref(L), p, C.byref(cam), C.byref(fr), 4, None, C.byref(ft), C.byref(sd), C.byref(opt),
                                    None, None, None, None, 0, None) < 0
    assert lib.svo_hip_update_seed_batch(3, None, None, C.byref(sd), None) == EINVAL
    assert lib.svo_hip_compute_tau_batch(3, None, None, None, 0.001, None, None) == EINVAL
    # the reprojector's selection rule
    assert lib.svo_hip_select_matches(None, 4, p, p, p, p, p, 120, p, p, p, p, p, p, None, 0, None) == EINVAL
    assert lib.svo_hip_select_matches(C.byref(cam), -1, p, p, p, p, p, 120, p, p, p, p, p, p, None, 0, None) == EINVAL
    assert lib.svo_hip_select_matches(C.byref(cam), 4, None, p, p, p, p, 120, p, p, p, p, p, p, None, 0, None) == EINVAL
    assert lib.svo_hip_select_matches(C.byref(cam), 4, p, p, p, p, p, -1, p, p, p, p, p, p, None, 0, None) == EINVAL
    # pose / point optimizers
    assert lib.svo_hip_pose_optimize(None, 1, None, 10, None, None, None, None, 2.0, 10, None, None, None, None, None) == EINVAL
    assert lib.svo_hip_pose_optimize(C.byref(cam), 1, p, 1 << 20, p, p, p, p, 2.0, 10, p, None, p, p, None) == ERANGE
    assert lib.svo_hip_point_optimize(None, 1, None, None, None, 5, None, None) < 0
    # detector
    assert lib.svo_hip_fast_detect(C.byref(L), p, 1, p, 9, 20, 30, 22, 16, None, 20.0, p, p, p, p, 1 << 30, None) == EINVAL   # levels > layout
    assert lib.svo_hip_fast_detect(C.byref(L), p, 1, p, 3, 20, 30, 22, 16, None, 20.0, p, p, p, p, 16, None) == ERANGE      # workspace too small
    # hipSOLVER cross-check, graphs
    assert lib.svo_hip_solve6_hipsolver(2, p, p, p, None, p, 8, None) == ERANGE
    assert lib.svo_hip_graph_end_capture(None, None) == EINVAL
    assert b"invalid" in lib.svo_hip_strerror(EINVAL) and b"limit" in lib.svo_hip_strerror(ERANGE)
    torch.cuda.synchronize()  # nothing above may have poisoned the context
    assert int(buf.sum().item()) == 0


def test_pin_calling_thread_stays_next_to_the_device(gpu_device):
    """svo_hip_pin_calling_thread: in a process of its own (the mask is inherited by everything the process starts), the calling
    thread ends up inside the device's local_cpulist -- or, on a host without NUMA information, where it was."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys, glob, json\n"
        f"sys.path.insert(0, {root!r})\n"
        "import torch\n"
        "from rpg_svo_amd import capi\n"
        "lib = capi.load()\n"
        "torch.zeros(1, device='cuda:0')\n"
        "before = sorted(os.sched_getaffinity(0))\n"
        "n = lib.svo_hip_pin_calling_thread()\n"
        "after = sorted(os.sched_getaffinity(0))\n"
        "p = torch.cuda.get_device_properties(0)\n"
        "bdf = '%04x:%02x:%02x.0' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, p.pci_device_id)\n"
        "f = '/sys/bus/pci/devices/' + bdf + '/local_cpulist'\n"
        "local = open(f).read().strip() if os.path.exists(f) else ''\n"
        "print(json.dumps(dict(n=n, before=before, after=after, local=local)))\n")
    q = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert q.returncode == 0, q.stderr[-2000:]
    import json
    r = json.loads(q.stdout.strip().splitlines()[-1])
    assert r["n"] >= 0
    if r["n"] == 0:
        assert r["after"] == r["before"]
        return
    cpus = set()
    for part in r["local"].split(","):
        lo, _, hi = part.partition("-")
        cpus |= set(range(int(lo), int(hi or lo) + 1))
    assert len(r["after"]) == r["n"] and set(r["after"]) <= cpus and set(r["after"]) <= set(r["before"])
