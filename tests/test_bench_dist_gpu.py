"""bench.py's multi-GPU launch path.

CPU: `python bench.py --gpus N` from a plain shell re-launches itself under torch.distributed.run
with the contract's arguments (checked through SVO_BENCH_DRY_SPAWN, nothing is started).
GPU: the RCCL code path (process group, double-buffered pose all-gather, per-frame rig gather)
executes on the 1-GPU box with a single rank (SVO_BENCH_FORCE_DIST=1); with two or more GPUs
visible the real `--gpus 2` launch runs, otherwise that test skips."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _last_json(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_plain_shell_launch_spawns_one_rank_per_gpu():
    env = dict(os.environ, SVO_BENCH_DRY_SPAWN="1")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "7", "--warmup", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    cmd = _last_json(p.stdout)["spawn"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index(BENCH) + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_path_runs_with_one_rank(gpu_device):
    env = dict(os.environ, SVO_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--batch", "512", "--no-cpu-baseline",
                        "--extras", "rig"], env=env, capture_output=True, text=True, timeout=580)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _last_json(p.stdout)
    assert r["n_gpus"] == 1 and r["value"] > 0
    g = r["gather"]
    assert g["collective"].startswith("all_gather") and g["bytes_per_rank_per_step"] == 512 * 96 and g["ms_blocking_avg"] > 0
    rig = r["rig_replay"]
    assert rig["cameras"] == 1 and rig["rig_frames_per_s"] > 0 and rig["gather_bytes_per_frame_set"] == 96


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_gpu_launch_from_plain_shell(gpu_device):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (the gpurun box has one)")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "512",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=880)
    assert p.returncode == 0, p.stderr[-3000:]
    r = _last_json(p.stdout)
    assert r["n_gpus"] == 2 and r["gather"]["bytes_gathered_per_step"] == 2 * 512 * 96
    assert r["rig_replay"]["cameras"] == 2
