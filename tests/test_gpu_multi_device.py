"""The in-library multi-device entry points (include/orb_hip.h: orb_bf_multi_*, orb_extract_batch_multi) on the
GPUs at hand.  With one GPU, a device listed several times stands in for several (packed words then travel by
device-to-device copies), and a single device goes through RCCL (a communicator of one rank), so that both
exchange paths run.  Results must equal the CPU oracle and the single-shard entry points bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

from orb_slam2_amd.synth import frame_seed, synth_descriptors, synth_queries, synth_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _devices():
    import torch

    return list(range(torch.cuda.device_count()))


def test_multi_device_brute_force_equals_oracle(oracle):
    from orb_slam2_amd import binding

    db = synth_descriptors(50021, 41)
    q = synth_queries(db, 1300, 42)
    db[40000:40050] = db[300:350]  # the best row exists twice, in different shards: lower index wins, second == best
    q[:50] = db[300:350]
    io, bo, so = oracle.bf_top2(q, db)
    real = _devices()
    layouts = [[0], [0, 0], [0, 0, 0, 0, 0]] + ([real] if len(real) > 1 else [])
    for devs in layouts:
        m = binding.MultiBruteForce(devs)
        assert m.exchange == ("rccl" if len(set(devs)) == len(devs) else "copy")
        m.set_database(db)
        for nq in (1300, 77):  # second call with fewer queries on the same buffers
            ig, bg, sg = m.query_top2(q[:nq])
            np.testing.assert_array_equal(ig, io[:nq], err_msg="devices %s" % devs)
            np.testing.assert_array_equal(bg, bo[:nq])
            np.testing.assert_array_equal(sg, so[:nq])
        assert (bg[:50] == 0).all() and (sg[:50] == 0).all()
        # fewer rows than shards, and an empty database
        for nd in (3, 0):
            m.set_database(db[:nd])
            ig, bg, sg = m.query_top2(q[:64])
            i2, b2, s2 = oracle.bf_top2(q[:64], db[:nd])
            np.testing.assert_array_equal(ig, i2)
            np.testing.assert_array_equal(bg, b2)
            np.testing.assert_array_equal(sg, s2)
        m.close()


def test_frame_blocks_over_several_contexts(oracle):
    from orb_slam2_amd import binding

    w, h, n = 640, 480, 11
    imgs = [synth_scene(w, h, frame_seed(1, 300 + f)) for f in range(n)]
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, 0x1F)
    real = _devices()
    for devs in ([0], [0, 0, 0]) + ((real,) if len(real) > 1 else ()):
        ctxs = [binding.Context(1000, 1.2, 8, 20, 7, w, h, n, d, 0x1F) for d in devs]
        res = binding.extract_batch_multi(ctxs, imgs)
        for f in (0, 3, 4, 10):
            ko, do = e.extract(imgs[f])
            assert len(res[f][0]) == len(ko) and (res[f][0] == ko).all() and (res[f][1] == do).all(), "frame %d, devices %s" % (f, devs)
        for c in ctxs:
            c.close()


def test_cpp_multi_device_demo():
    exe = os.path.join(ROOT, "examples", "orb_bf_multi")
    assert os.path.exists(exe), "examples/orb_bf_multi is built by __graft_entry__.build()"
    for devs in ("0", "0,0,0"):
        r = subprocess.run([exe, devs, "300000", "1500"], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out["differ_from_single_shard"] == 0 and out["exact_ties"] >= 70
        assert out["exchange"] == ("rccl" if devs == "0" else "copy")
