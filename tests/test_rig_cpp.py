"""The C++ side of the multi-GPU path (rpg_svo_amd/host/rig + tools/svo_rig_replay.cpp): one process per
camera and GPU, RCCL all-gather of the SE(3) results after every frame set.

CPU: the bootstrap that hands rank 0's RCCL unique id to the other ranks (plain TCP, three processes).
GPU: the rig tool with one rank (RCCL communicator + all-gather execute on the 1-GPU box); with two or
more GPUs visible, two ranks through torch.distributed.run --no-python."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RIG = os.path.join(ROOT, "rpg_svo_amd", "host", "rig")


def test_unique_id_broadcast_three_processes(tmp_path):
    exe = str(tmp_path / "test_bootstrap")
    subprocess.run(["g++", "-std=c++11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", RIG, "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "host", "test_bootstrap.cpp"), os.path.join(RIG, "pose_exchange.cpp"),
                    "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe], check=True)
    port = 30000 + os.getpid() % 2000
    procs = [subprocess.Popen([exe, str(r), "3", str(port)], stdout=subprocess.PIPE, text=True) for r in (2, 1, 0)]
    outs = [p.communicate(timeout=60)[0].strip() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert len(set(outs)) == 1 and outs[0] != "0"


def test_bootstrap_ignores_strangers(tmp_path):
    """Rank 0 hands the blob only to peers that introduce themselves as a rank of this job: a connection that says
    something else gets no byte of it and does not use up a rank's place."""
    import socket
    import time
    exe = str(tmp_path / "test_bootstrap")
    subprocess.run(["g++", "-std=c++11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", RIG, "-I", "/opt/rocm/include",
                    os.path.join(ROOT, "tests", "host", "test_bootstrap.cpp"), os.path.join(RIG, "pose_exchange.cpp"),
                    "-L", "/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-pthread", "-o", exe], check=True)
    port = 32000 + os.getpid() % 2000
    p0 = subprocess.Popen([exe, "0", "2", str(port)], stdout=subprocess.PIPE, text=True)
    got = None
    for _ in range(100):  # the stranger: connects as soon as rank 0 listens, sends 16 bytes that are not a hello
        try:
            with socket.create_connection(("127.0.0.1", port), timeout=2) as so:
                so.sendall(b"GET / HTTP/1.0\r\n")
                so.settimeout(5)
                got = so.recv(256)
            break
        except ConnectionRefusedError:
            time.sleep(0.05)
    assert got == b"", got  # closed without a byte
    p1 = subprocess.Popen([exe, "1", "2", str(port)], stdout=subprocess.PIPE, text=True)
    o0, o1 = p0.communicate(timeout=60)[0].strip(), p1.communicate(timeout=60)[0].strip()
    assert p0.returncode == 0 and p1.returncode == 0 and o0 == o1 and o0 != "0"


def _rig_exe():
    exe = os.path.join(ROOT, "build", "svo_rig_replay")
    if not os.path.exists(exe):
        pytest.fail("build/svo_rig_replay missing: python -c 'import __graft_entry__ as g; g.build()'")
    return exe


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_rig_replay_one_rank(gpu_device):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([_rig_exe(), "60"], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["cameras"] == 1 and r["rig_frames_per_s"] > 1000 and r["worst_translation_error_m"] < 2e-3


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rig_replay_two_ranks(gpu_device):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (the gpurun box has one)")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--no-python", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29000 + os.getpid() % 900), _rig_exe(), "60"],
                       capture_output=True, text=True, timeout=580)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["cameras"] == 2 and r["gather_bytes_per_frame_set"] == 192 and r["worst_translation_error_m"] < 2e-3
