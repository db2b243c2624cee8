"""Multi-process host logic on CPU (gloo, world size 2 and 3): row-sharded
brute-force top-2 with one all-gather of packed words and a merge.  The scan
and the merge kernels cannot run without a GPU, so the oracle stands in for
them here; what is under test is the sharding, the packed exchange format, the
collective call and the merge RULE (re-stated in numpy below), which the GPU
tests then check against the HIP merge kernel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from orb_slam2_amd.sharded import ShardedBruteForce, pack_top2, shard_range, unpack_top2  # noqa: E402
from orb_slam2_amd.synth import synth_descriptors, synth_queries  # noqa: E402


def merge_rule(gathered):
    """Reference statement of the merge: best = min (distance, row); second =
    second-smallest distance with multiplicity over every shard's best and second."""
    g = gathered.numpy().view(np.uint64)
    world, nq = g.shape
    idx = np.full(nq, -1, np.int64)
    best = np.full(nq, 0xFFFF, np.int64)
    second = np.full(nq, 0xFFFF, np.int64)
    for s in range(world):
        i, b, d2 = unpack_top2(g[s])
        b = b.astype(np.int64)
        d2 = d2.astype(np.int64)
        i = i.astype(np.int64)
        has = b != 0xFFFF
        better = has & ((b < best) | ((b == best) & (i < idx)))
        second = np.where(better, best, np.where(has, np.minimum(second, b), second))
        best = np.where(better, b, best)
        idx = np.where(better, i, idx)
        second = np.where(has, np.minimum(second, d2), second)
    return idx.astype(np.int32), best.astype(np.uint16), second.astype(np.uint16)


def _worker(rank, world, port, db, q, out):
    import oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(len(db), rank, world)

    def scan(queries):
        idx, b, s = oracle.bf_top2(queries.numpy(), db[lo:hi])
        gidx = np.where(idx >= 0, idx.astype(np.int64) + lo, 0xFFFFFFFF)
        return torch.from_numpy(pack_top2(b, s, gidx).view(np.int64))

    sb = ShardedBruteForce(scan, merge_rule, always_gather=(world == 1))  # a group of one still runs the collective
    idx, best, second = sb.query(torch.from_numpy(q))
    if rank == world - 1:  # every rank holds the full answer; check a non-zero rank
        np.save(out, np.stack([idx.astype(np.int64), best.astype(np.int64), second.astype(np.int64)]))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_bruteforce_gloo(world, tmp_path, oracle):
    db = synth_descriptors(3001, 31)
    q = synth_queries(db, 200, 32)
    db[2500:2520] = db[10:30]  # duplicates of the best row in another shard
    q[:20] = db[10:30]
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), db, q, out), nprocs=world, join=True)
    got = np.load(out)
    io, bo, so = oracle.bf_top2(q, db)
    np.testing.assert_array_equal(got[0], io)
    np.testing.assert_array_equal(got[1], bo)
    np.testing.assert_array_equal(got[2], so)
    assert (got[1][:20] == 0).all() and (got[2][:20] == 0).all() and (got[0][:20] == np.arange(10, 30)).all()


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 64, 1000, 10_000_000):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip():
    b = np.array([0, 17, 256, 0xFFFF], np.uint16)
    s = np.array([0, 40, 256, 0xFFFF], np.uint16)
    i = np.array([0, 123456789, 4294967294, -1], np.int64)
    idx, bb, ss = unpack_top2(pack_top2(b, s, i))
    assert list(bb) == list(b) and list(ss) == list(s)
    assert list(idx[:2]) == [0, 123456789] and idx[3] == -1
