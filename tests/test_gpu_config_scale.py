"""Parity at the sizes the benchmark configurations run at, where the smaller parity cases leave paths
untested: the chunk geometry of a 100 000 x 10 000 000 brute-force scan (the timed launch itself, a slice
of its queries against the CPU oracle, src/ORBmatcher.cc:53-76 applied densely), a batch of distinct
3840x2160 frames (quadtree levels that run from global memory, several frames per launch, both frame
groups; src/ORBextractor.cc:571-796), and a batch of 64 frames of 1241x376 with 2000 features.
Everything here is integer / byte / index work: bit-exact."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from orb_slam2_amd.synth import frame_seed, synth_scene

pytestmark = pytest.mark.gpu

FORK = 0x1F


def _oracle_top2_threads(oracle, q, db, threads=8):
    """oracle.bf_top2 over slices of the queries on several host threads (ctypes drops the GIL)."""
    parts = np.array_split(np.arange(len(q)), threads)
    with ThreadPoolExecutor(threads) as ex:
        res = list(ex.map(lambda ix: oracle.bf_top2(q[ix], db), parts))
    return tuple(np.concatenate([r[k] for r in res]) for k in range(3))


def _unpack(p):
    p = p.view(np.uint64)
    return ((p & np.uint64(0xFFFFFFFF)).astype(np.int64), (p >> np.uint64(48)).astype(np.uint16),
            ((p >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.uint16))


def test_brute_force_ten_million_rows(oracle):
    import torch

    from orb_slam2_amd import binding

    nq, nd = 100_000, 10_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(0x0B5EED05)
    d_db = torch.randint(0, 256, (nd, 32), dtype=torch.uint8, device="cuda", generator=g)
    d_q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device="cuda", generator=g)
    # checked queries: every 199th (503 of them, spread over all query workgroups)
    sel = torch.arange(0, nq, 199, device="cuda")
    ns = len(sel)
    # a third of them copy a database row with a few bits flipped (a clear best, random second)
    near = sel[: ns // 3]
    rows = (torch.arange(len(near), device="cuda") * 29_989 + 17) % nd
    d_q[near] = d_db[rows]
    d_q[near, 3] ^= 0x11
    # ties for best across chunks: the same row planted at two places millions of rows apart, and at
    # neighbouring rows; the lower index must win and second == best
    tie = sel[ns // 3 : ns // 3 + 64]
    a = (torch.arange(64, device="cuda") * 71_993 + 5) % (nd // 2 - 2000)
    d_db[a + nd // 2 + 1234] = d_db[a]
    d_q[tie] = d_db[a]
    tie2 = sel[ns // 3 + 64 : ns // 3 + 96]
    b = (torch.arange(32, device="cuda") * 300_007 + 11) % (nd - 2)
    d_db[b + 1] = d_db[b]
    d_q[tie2] = d_db[b]
    # the last rows of the database (tile loads clamp there) hold the best for some queries
    last = sel[ns - 8 :]
    d_q[last] = d_db[nd - 8 :]
    bf = binding.BruteForce(0)
    bf.set_database_device(d_db, 0)
    d_packed = torch.zeros(nq, dtype=torch.int64, device="cuda")
    bf.query_packed_device(d_q, d_packed)
    torch.cuda.synchronize()
    db = d_db.cpu().numpy()
    q = d_q[sel].cpu().numpy()
    idx, best, second = _unpack(d_packed[sel].cpu().numpy())
    io, bo, so = _oracle_top2_threads(oracle, q, db)
    np.testing.assert_array_equal(idx, io.astype(np.int64))
    np.testing.assert_array_equal(best, bo)
    np.testing.assert_array_equal(second, so)
    t0 = ns // 3
    assert (best[t0 : t0 + 96] == 0).all() and (second[t0 : t0 + 96] == 0).all()
    assert (idx[t0 : t0 + 64] == a.cpu().numpy()).all()
    assert (idx[ns - 8 :] == np.arange(nd - 8, nd)).all()
    # a database that is not a multiple of anything, fewer queries (another chunk geometry), global base
    nd2 = 9_999_991
    bf.set_database_device(d_db[:nd2], 7)
    q2 = d_q[sel[:256]].contiguous()
    p2 = torch.zeros(256, dtype=torch.int64, device="cuda")
    bf.query_packed_device(q2, p2)
    torch.cuda.synchronize()
    idx, best, second = _unpack(p2.cpu().numpy())
    io, bo, so = _oracle_top2_threads(oracle, q[:256], db[:nd2])
    np.testing.assert_array_equal(idx, io.astype(np.int64) + 7)
    np.testing.assert_array_equal(best, bo)
    np.testing.assert_array_equal(second, so)
    bf.close()


def _assert_frame_equal(binding, got, ref, what):
    kg, dg = got
    ko, do = ref
    assert len(kg) == len(ko), "%s: %d keypoints, oracle %d" % (what, len(kg), len(ko))
    for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
        np.testing.assert_array_equal(kg[f], ko[f], err_msg="%s field %s" % (what, f))
    np.testing.assert_array_equal(dg, do, err_msg="%s descriptors" % what)


def test_batch_of_distinct_4k_frames(oracle):
    """Sixteen distinct 3840x2160 frames in one batch (two frame groups of eight): every frame against the
    oracle's result for that frame."""
    from orb_slam2_amd import binding

    w, h, nf, n = 3840, 2160, 4000, 16
    # four scenes drawn in full, the others are shifted copies with new noise: distinct key sets, cheap to make
    base = [synth_scene(w, h, frame_seed(4, 10 + i)) for i in range(4)]
    imgs = []
    for f in range(n):
        if f < 4:
            imgs.append(base[f])
        else:
            imgs.append(synth_scene(w, h, frame_seed(4, 10 + f % 4), shift=(f, 2 * f - 7), noise_seed=1000 + f))
    c = binding.Context(nf, 1.2, 8, 20, 7, w, h, n, 0, FORK)
    res = c.extract_batch(imgs)
    def work(t):  # one oracle extractor per host thread, frames t, t + 4, ...
        e = oracle.Extractor(nf, 1.2, 8, 20, 7, FORK)
        return [(f, e.extract(imgs[f])) for f in range(t, n, 4)]

    with ThreadPoolExecutor(4) as ex:
        refs = dict(sum(ex.map(work, range(4)), []))
    for f in range(n):
        _assert_frame_equal(binding, res[f], refs[f], "4K frame %d" % f)
    assert len({r[1].tobytes() for r in refs.values()}) == n  # the frames really differ
    # the same frames one at a time on the same context
    for f in (0, 9):
        k1, d1 = c.extract(imgs[f])
        assert (k1 == res[f][0]).all() and (d1 == res[f][1]).all()
    c.close()


def test_batch_of_64_kitti_frames(oracle):
    """64 frames of 1241x376, 2000 features: every frame of the batch equals the same frame extracted alone,
    and a sample of them the oracle."""
    from orb_slam2_amd import binding

    w, h, nf, n = 1241, 376, 2000, 64
    imgs = [synth_scene(w, h, frame_seed(2, f)) for f in range(n)]
    c = binding.Context(nf, 1.2, 8, 20, 7, w, h, n, 0, FORK)
    res = c.extract_batch(imgs)
    for f in range(n):
        k1, d1 = c.extract(imgs[f])
        assert len(k1) == len(res[f][0]) and (k1 == res[f][0]).all() and (d1 == res[f][1]).all(), "frame %d" % f
    e = oracle.Extractor(nf, 1.2, 8, 20, 7, FORK)
    for f in (0, 21, 42, 63):
        _assert_frame_equal(binding, res[f], e.extract(imgs[f]), "KITTI-shaped frame %d" % f)
    c.close()


def test_sharded_frame_job_does_not_depend_on_the_ranks(oracle):
    """The benchmark's sharded frame job (tools/bench_configs.py config4: contiguous frame blocks per rank, no
    exchange; the unit being sharded is one ORBextractor::operator() call, src/Frame.cc:67-74): the sum of the
    per-frame output CRCs is the same for one rank and for three ranks run one after the other here (uneven blocks,
    one of them shorter than a step), and a frame of the job equals the oracle's extraction of it."""
    import os
    import sys

    import torch

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import bench_configs as bc

    kw = dict(job=20, step_frames=6, w=1280, h=720, nf=1500, scenes=3)
    one = bc.config4(bc.Ranks(), **kw)
    assert one["frames_per_rank"] == 20 and one["mean_keypoints"] > 1000
    got = []
    for r in range(3):
        # (a rank's "sum over ranks" here just records its share)
        res = bc.config4(bc.Ranks(rank=r, world=3, sum_over_ranks=lambda x: x), **kw)
        got.append(res)
    assert [g["frames_per_rank"] for g in got] == [6, 7, 7]
    assert sum(g["sum_of_frame_crc32"] for g in got) == one["sum_of_frame_crc32"]
    # frame 7 of the job (scene 1 rolled by 10 columns) against the oracle
    from orb_slam2_amd import binding

    d = bc.job_frames(7, 8, 1280, 720, 4, 3, torch.device("cuda", 0))
    img = d[0].cpu().numpy()
    np.testing.assert_array_equal(img, np.roll(synth_scene(1280, 720, frame_seed(4, 1)), 10, axis=1))
    ctx = binding.Context(1500, 1.2, 8, 20, 7, 1280, 720, 1, 0, FORK)
    k, dsc = ctx.extract(img)
    ko, do = oracle.Extractor(1500, 1.2, 8, 20, 7, oracle.QUIRKS_FORK).extract(img)
    assert len(k) == len(ko) and (k == ko).all() and (dsc == do).all()
    ctx.close()
