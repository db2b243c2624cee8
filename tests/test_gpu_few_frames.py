"""Calls of a few frames take latency-oriented paths (one launch for the whole scale pyramid, results written
straight to host memory); batches take the throughput paths.  Both must return the same bytes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(640, 480), (752, 480), (1241, 376), (253, 249), (333, 251), (1920, 1080), (2047, 1153)]


def _frame(w, h, seed):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (h // 8 + 2, w // 8 + 2), dtype=np.uint8)
    img = np.kron(base, np.ones((8, 8), dtype=np.uint8))[:h, :w].astype(np.int32)
    img += rng.integers(-20, 21, (h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("quirks", [0x1F, 0x1E], ids=["area", "linear"])
@pytest.mark.parametrize("w,h", SIZES)
def test_few_frame_paths_match_the_batch_paths(w, h, quirks):
    from orb_slam2_amd import binding

    frames = [_frame(w, h, 100 + i) for i in range(3)]
    big = binding.Context(1000, 1.2, 8, 20, 7, w, h, 8, 0, quirks)
    res = big.extract_batch(frames + [frames[0]] * 5)  # eight frames: level-by-level pyramid, results by copy
    kb, db = [r[0] for r in res], [r[1] for r in res]
    ref_levels = [[big.pyramid_level(f, l) for l in range(8)] for f in range(3)]
    one = binding.Context(1000, 1.2, 8, 20, 7, w, h, 3, 0, quirks)
    for f in range(3):  # one frame per call
        k, d = one.extract(frames[f])
        for l in range(8):
            assert np.array_equal(one.pyramid_level(0, l), ref_levels[f][l]), (f, l)
        assert np.array_equal(k, kb[f]) and np.array_equal(d, db[f])
    res = one.extract_batch(frames)  # three frames in one call
    k3, d3 = [r[0] for r in res], [r[1] for r in res]
    for f in range(3):
        for l in range(8):
            assert np.array_equal(one.pyramid_level(f, l), ref_levels[f][l]), (f, l)
        assert np.array_equal(k3[f], kb[f]) and np.array_equal(d3[f], db[f])
    big.close()
    one.close()


def test_few_frame_path_reports_the_callers_capacity():
    """The flag word reaches the host without a copy of its own on this path."""
    import ctypes as C

    from orb_slam2_amd import binding

    img = _frame(640, 480, 7)
    ctx = binding.Context(1000, 1.2, 8, 20, 7, 640, 480, 1, 0, 0x1F)
    k, d = ctx.extract(img)
    assert len(k) > 100
    cap = len(k) - 10
    kps = np.zeros(cap, binding.KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int(0)
    rc = binding.lib().orb_extract(ctx.h, img.ctypes.data_as(C.c_void_p), 640, 480, 640, kps.ctypes.data_as(C.c_void_p),
                                   desc.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    assert rc != 0
    k2, d2 = ctx.extract(img)  # and the context works on
    assert np.array_equal(k, k2) and np.array_equal(d, d2)
    ctx.close()


def test_batch_into_arrays_the_caller_keeps():
    from orb_slam2_amd import binding

    frames = [_frame(640, 480, 300 + i) for i in range(9)]
    ctx = binding.Context(1000, 1.2, 8, 20, 7, 640, 480, 9, 0, 0x1F)
    ref = ctx.extract_batch(frames)
    kps = np.zeros((9, ctx.cap), binding.KP_DTYPE)
    desc = np.zeros((9, ctx.cap, 32), np.uint8)
    cnt = np.zeros(9, np.int32)
    for _ in range(2):  # the second call finds the first one's rows in the arrays
        ctx.extract_batch_into(frames, kps, desc, cnt)
        for f in range(9):
            assert cnt[f] == len(ref[f][0])
            assert np.array_equal(kps[f, : cnt[f]], ref[f][0]) and np.array_equal(desc[f, : cnt[f]], ref[f][1])
    ctx.close()


def _crowded(w, h, seed):
    """Flat frame with one textured patch: every keypoint of a level sits in a small part of it, so the quadtree has to
    divide far below the depth ordinary frames need."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 128, np.uint8)
    side = int(rng.integers(90, 180))
    x0, y0 = int(rng.integers(30, w - side - 30)), int(rng.integers(30, h - side - 30))
    img[y0 : y0 + side, x0 : x0 + side] = _frame(side, side, seed + 1)
    return img


@pytest.mark.parametrize("quirks", [0x1F, 0x00], ids=["fork", "canonical"])
def test_quadtree_second_pass_takes_the_crowded_levels(oracle, quirks, monkeypatch):
    """A large batch runs DistributeOctTree (src/ORBextractor.cc:571-796) in two passes; which pass settles a level
    must not show in the result.  (The threshold is lowered so that twelve frames count as large.)"""
    from orb_slam2_amd import binding

    monkeypatch.setenv("ORBHIP_QT_TWO_PASS_MIN", "16")
    w, h = 640, 480
    frames = [_crowded(w, h, 500 + 2 * i) for i in range(6)] + [_frame(w, h, 600 + i) for i in range(6)]
    ctx = binding.Context(1000, 1.2, 8, 20, 7, w, h, 12, 0, quirks)
    res = ctx.extract_batch(frames)
    second = ctx.quadtree_second_pass_levels()
    assert 0 < second < 12 * 8, second  # the crowded frames' levels, not everything
    one = binding.Context(1000, 1.2, 8, 20, 7, w, h, 1, 0, quirks)
    for f in range(12):
        k, d = one.extract(frames[f])  # one pass, full depth
        assert np.array_equal(k, res[f][0]) and np.array_equal(d, res[f][1]), f
    assert one.quadtree_second_pass_levels() == 0
    ext = oracle.Extractor(1000, 1.2, 8, 20, 7, quirks)
    for f in (0, 3, 7):
        ko, do = ext.extract(frames[f])
        assert len(ko) == len(res[f][0]) and (ko == res[f][0]).all() and (do == res[f][1]).all(), f
    # the second pass leaves its lists empty for the next batch: ordinary frames next (fewer levels listed, if any),
    # then the first batch again, with the frames in another order (other frame groups hold the crowded ones)
    plain = [_frame(w, h, 700 + i) for i in range(12)]
    res2 = ctx.extract_batch(plain)
    assert ctx.quadtree_second_pass_levels() < second
    for f in (0, 5, 11):
        k, d = one.extract(plain[f])
        assert np.array_equal(k, res2[f][0]) and np.array_equal(d, res2[f][1]), f
    order = [11, 0, 7, 3, 9, 1, 6, 2, 10, 4, 8, 5]
    res3 = ctx.extract_batch([frames[f] for f in order])
    assert ctx.quadtree_second_pass_levels() == second
    for k_, f in enumerate(order):
        assert np.array_equal(res3[k_][0], res[f][0]) and np.array_equal(res3[k_][1], res[f][1]), f
    ctx.close()
    one.close()
