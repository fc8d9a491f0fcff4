"""N>1 path on CPU: world-size-2 gloo processes shard a batch of alignment problems,
solve their shard (with the oracle standing in for the device kernel -- there is no
GPU here) and gather the poses; the result must equal the single-process run."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _worker(rank, world, port, n_total, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle
    from rpg_svo_amd import synth
    from rpg_svo_amd.dist import gather_poses, shard_range
    from helpers import make_batch, run_oracle
    seq = synth.make_sequence(n_total + 1, 60, cam=synth.Camera(320, 240, 200.0, 200.0, 160.0, 120.0), margin=16, cell=24)
    lo, hi = shard_range(n_total, rank, world)
    b = make_batch(seq, [(i, i + 1) for i in range(lo, hi)], 3)
    T, _, _ = run_oracle(pyoracle, b, 2, 0, n_threads=1)
    counts = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    allT = gather_poses(torch.from_numpy(T), counts)
    if rank == 0:
        np.save(os.path.join(tmp, "gathered.npy"), allT.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from rpg_svo_amd.dist import shard_range
    for n in (0, 1, 7, 8, 9):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gather_matches_single_process(tmp_path, oracle):
    from rpg_svo_amd import synth
    from helpers import make_batch, run_oracle
    n_total = 5  # ragged: 3 + 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    seq = synth.make_sequence(n_total + 1, 60, cam=synth.Camera(320, 240, 200.0, 200.0, 160.0, 120.0), margin=16, cell=24)
    b = make_batch(seq, [(i, i + 1) for i in range(n_total)], 3)
    T, _, _ = run_oracle(oracle, b, 2, 0, n_threads=1)
    assert got.shape == (n_total, 12)
    assert np.array_equal(got, T)


def _overlap_worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rpg_svo_amd.dist import OverlappedPoseGather
    g = OverlappedPoseGather(3, width=12, device="cpu")
    got = []
    for i in range(5):  # more steps than buffers: every buffer is reused after its gather
        T = g.local(i)
        T.copy_(torch.full((3, 12), float(100 * i + rank), dtype=torch.float64))
        g.submit(i)
        if i >= 1:
            got.append(g.result(i - 1).clone())  # consume one step behind, as the bench does
    got.append(g.result(4).clone())
    g.drain()
    if rank == 1:
        np.save(os.path.join(tmp, "overlap.npy"), torch.stack(got).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_overlapped_pose_gather_two_ranks(tmp_path):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_overlap_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "overlap.npy"))
    assert got.shape == (5, 6, 12)
    for i in range(5):
        assert (got[i, :3] == 100 * i).all() and (got[i, 3:] == 100 * i + 1).all()
