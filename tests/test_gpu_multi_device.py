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
        for nq in (100, 1300, 77, 900):  # the buffers grow (100 -> 1300), then serve fewer queries as they are
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


def _same(res, imgs, e, frames):
    for f in frames:
        ko, do = e.extract(imgs[f])
        assert len(res[f][0]) == len(ko) and (res[f][0] == ko).all() and (res[f][1] == do).all(), "frame %d" % f


def test_page_locked_frames_skip_the_staging_copy(oracle):
    """Frames in page-locked memory (orb_alloc_host) are fetched by the DMA engine from where they are: as one
    block (frames stride*h apart), with padded rows (two-dimensional copy), scattered (one copy per frame), a
    single frame, and over several contexts enqueued from one thread.  Results equal the pageable path's and the
    oracle's (unit: ORBextractor::operator(), src/ORBextractor.cc:448-509)."""
    from orb_slam2_amd import binding

    w, h, n = 640, 480, 13
    scenes = [synth_scene(w, h, frame_seed(1, 500 + f)) for f in range(n)]
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, 0x1F)
    c = binding.Context(1000, 1.2, 8, 20, 7, w, h, n, 0, 0x1F)
    ref = c.extract_batch(scenes)  # pageable
    _same(ref, scenes, e, (0, 5, 12))
    block = binding.alloc_host((n, h, w))
    block[:] = np.stack(scenes)
    got = c.extract_batch([block[f] for f in range(n)])
    for f in range(n):
        assert (got[f][0] == ref[f][0]).all() and (got[f][1] == ref[f][1]).all()
    # rows 704 bytes apart (the context's own pitch is 640): one 2-D copy; the padding holds other values
    padded = binding.alloc_host((n, h, 704))
    padded[:] = 0xAB
    padded[:, :, :w] = block
    ptrs = [padded[f] for f in range(n)]
    kps = np.zeros((n, c.cap), binding.KP_DTYPE)
    desc = np.zeros((n, c.cap, 32), np.uint8)
    cnt = np.zeros(n, np.int32)
    import ctypes as C

    arr = (C.c_void_p * n)(*[p.ctypes.data for p in ptrs])
    rc = binding.lib().orb_extract_batch(c.h, arr, n, w, h, 704, kps.ctypes.data_as(C.c_void_p), desc.ctypes.data_as(C.c_void_p),
                                         c.cap, cnt.ctypes.data_as(C.c_void_p))
    assert rc == 0, binding.lib().orb_last_error()
    for f in range(n):
        assert cnt[f] == len(ref[f][0]) and (kps[f, : cnt[f]] == ref[f][0]).all() and (desc[f, : cnt[f]] == ref[f][1]).all()
    # scattered page-locked frames, in another order than they lie in memory; then one frame alone
    order = [7, 2, 11, 0, 5, 9, 1]
    got = c.extract_batch([block[f] for f in order])
    for k, f in enumerate(order):
        assert (got[k][0] == ref[f][0]).all() and (got[k][1] == ref[f][1]).all()
    k1, d1 = c.extract(block[4])
    assert (k1 == ref[4][0]).all() and (d1 == ref[4][1]).all()
    # a mix of page-locked and pageable frames takes the staging path
    got = c.extract_batch([block[0], scenes[1], block[2], scenes[3]])
    for f in range(4):
        assert (got[f][0] == ref[f][0]).all() and (got[f][1] == ref[f][1]).all()
    c.close()
    # several contexts, page-locked frames: every block enqueued by the calling thread
    devs = _devices()
    for layout in ([0, 0, 0], devs if len(devs) > 1 else [0, 0]):
        ctxs = [binding.Context(1000, 1.2, 8, 20, 7, w, h, n, d, 0x1F) for d in layout]
        res = binding.extract_batch_multi(ctxs, [block[f] for f in range(n)])
        for f in range(n):
            assert (res[f][0] == ref[f][0]).all() and (res[f][1] == ref[f][1]).all(), "frame %d over %s" % (f, layout)
        for x in ctxs:
            x.close()


def test_few_host_frames_with_unaligned_rows(oracle):
    """Calls of fewer than four frames fetch their frames with a kernel that reads the page-locked memory over the
    link, whole dwords at a time: frames whose rows start at odd addresses, are an odd number of bytes apart and an
    odd number of pixels wide, page-locked (read where they lie) and pageable (staged first), one to three per
    call, with other values in the padding.  Results equal the oracle's
    (unit: ORBextractor::operator(), src/ORBextractor.cc:448-509)."""
    import ctypes as C

    from orb_slam2_amd import binding

    w, h, stride = 637, 477, 701
    scenes = [np.ascontiguousarray(synth_scene(640, 480, frame_seed(1, 700 + f))[:h, :w]) for f in range(3)]
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, 0x1F)
    c = binding.Context(1000, 1.2, 8, 20, 7, w, h, 4, 0, 0x1F)
    ref = c.extract_batch(scenes)
    _same(ref, scenes, e, (0, 1, 2))
    flat = binding.alloc_host((3 * stride * h + 64,))
    flat[:] = 0x5C
    heap = np.full(3 * stride * h + 64, 0xC5, np.uint8)
    for buf in (flat, heap):
        views = []
        for f in range(3):
            v = np.lib.stride_tricks.as_strided(buf[1 + f * stride * h:], shape=(h, w), strides=(stride, 1))
            v[:] = scenes[f]
            views.append(v)
        for n in (1, 2, 3):
            kps = np.zeros((n, c.cap), binding.KP_DTYPE)
            desc = np.zeros((n, c.cap, 32), np.uint8)
            cnt = np.zeros(n, np.int32)
            arr = (C.c_void_p * n)(*[views[f].ctypes.data for f in range(n)])
            rc = binding.lib().orb_extract_batch(c.h, arr, n, w, h, stride, kps.ctypes.data_as(C.c_void_p),
                                                 desc.ctypes.data_as(C.c_void_p), c.cap, cnt.ctypes.data_as(C.c_void_p))
            assert rc == 0, binding.lib().orb_last_error()
            for f in range(n):
                assert cnt[f] == len(ref[f][0]) and (kps[f, : cnt[f]] == ref[f][0]).all() and (desc[f, : cnt[f]] == ref[f][1]).all()
    c.close()


def test_device_resident_blocks_over_several_contexts(oracle):
    """orb_extract_batch_multi_device: contiguous frame blocks that already lie on their contexts' devices, one
    enqueue per context from the calling thread (own streams and the contexts' streams); a 3840x2160 block
    exercises the large-LDS quadtree plan on every context."""
    import torch

    from orb_slam2_amd import binding

    devs = _devices()
    for (w, h, nf, n, layout) in ((640, 480, 1000, 11, [0, 0, 0]), (3840, 2160, 4000, 3, [0, 0]),
                                  (640, 480, 1000, 11, devs if len(devs) > 1 else [0])):
        imgs = [synth_scene(w, h, frame_seed(1, 700 + f)) for f in range(n)]
        e = oracle.Extractor(nf, 1.2, 8, 20, 7, 0x1F)
        R = len(layout)
        ctxs = [binding.Context(nf, 1.2, 8, 20, 7, w, h, n, d, 0x1F) for d in layout]
        cap = ctxs[0].cap
        lo = [n * r // R for r in range(R + 1)]
        blocks, kps, desc, cnt = [], [], [], []
        for r, d in enumerate(layout):
            dev = torch.device("cuda", d)
            m = lo[r + 1] - lo[r]
            blocks.append(torch.from_numpy(np.stack(imgs[lo[r] : lo[r + 1]])).to(dev) if m else torch.empty((0, h, w), dtype=torch.uint8, device=dev))
            kps.append(torch.zeros((m, cap, 28), dtype=torch.uint8, device=dev))
            desc.append(torch.zeros((m, cap, 32), dtype=torch.uint8, device=dev))
            cnt.append(torch.zeros(m, dtype=torch.int32, device=dev))
        for own_streams in (False, True):
            for t in kps + desc + cnt:
                t.zero_()
            streams = None
            if own_streams:
                streams = [torch.cuda.Stream(device=torch.device("cuda", d)) for d in layout]
            for d in set(layout):
                torch.cuda.synchronize(d)
            binding.extract_batch_multi_device(ctxs, blocks, kps, desc, cnt, streams)
            if streams:
                for s in streams:
                    s.synchronize()
            for x in ctxs:
                x.check_last_batch()
            check = (0, n - 1) if w > 1000 else (0, 3, 4, n - 1)
            for f in check:
                r = max(i for i in range(R) if lo[i] <= f)
                k = f - lo[r]
                m = int(cnt[r][k].item())
                kg = kps[r][k, :m].cpu().numpy().view(binding.KP_DTYPE).reshape(-1)
                ko, do = e.extract(imgs[f])
                assert m == len(ko) and (kg == ko).all() and (desc[r][k, :m].cpu().numpy() == do).all(), "frame %d, %s" % (f, layout)
        for x in ctxs:
            x.close()
