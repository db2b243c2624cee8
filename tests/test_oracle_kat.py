"""Known-answer tests that pin the CPU oracle.

The reference has no tests or golden vectors for this path (parity unpinned),
so the oracle is pinned by values derived by hand from the reference's text
(src/ORBextractor.cc:369-436, src/ORBmatcher.cc:413-476) and from the
published definitions of the OpenCV primitives it calls.
"""
import numpy as np
import pytest

from orb_slam2_amd.synth import frame_seed, synth_descriptors, synth_scene


def test_level_sizes_and_quotas(oracle):
    e = oracle.Extractor(1000, 1.2, 8, 20, 7)
    e.compute_pyramid(np.zeros((480, 640), np.uint8))
    sizes = [e.level(l).shape[::-1] for l in range(8)]
    assert sizes == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    t = e.tables()
    assert list(t["per_level"]) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert t["per_level"].sum() == 1000
    assert t["scale"][0] == 1.0 and np.float32(t["scale"][1]) == np.float32(1.2)
    np.testing.assert_array_equal(t["inv_scale"], np.float32(1.0) / t["scale"])
    e2 = oracle.Extractor(2000, 1.2, 8, 20, 7)
    assert list(e2.tables()["per_level"]) == [434, 362, 302, 251, 209, 175, 145, 122]
    e4 = oracle.Extractor(4000, 1.2, 8, 20, 7)
    assert list(e4.tables()["per_level"]) == [869, 724, 603, 503, 419, 349, 291, 242]


def test_umax_tables(oracle):
    fork = oracle.Extractor(quirks=oracle.QUIRKS_FORK).tables()["umax"]
    canon = oracle.Extractor(quirks=oracle.QUIRKS_CANONICAL).tables()["umax"]
    assert list(fork) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 4, 3, 2, 1, 0]
    assert list(canon) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def test_gaussian_kernel(oracle):
    k = np.array([18, 34, 48, 56, 48, 34, 18])
    assert k.sum() == 256
    const = np.full((20, 24), 77, np.uint8)
    np.testing.assert_array_equal(oracle.gaussian(const), const)
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    out = oracle.gaussian(imp)
    expect = (np.outer(k, k) * 255 + 32768) >> 16
    np.testing.assert_array_equal(out[7:14, 7:14], expect)
    assert out.sum() == expect.sum()
    # REFLECT_101 at the border: a ramp in x reflects about column 0
    ramp = np.tile(np.arange(0, 40, 2, dtype=np.uint8), (9, 1))
    out = oracle.gaussian(ramp)
    h0 = sum(kk * ramp[0, abs(i - 3)] for i, kk in enumerate(k))
    assert out[4, 0] == (256 * h0 + 32768) >> 16
    kl = np.array([18, 34, 49, 55, 49, 34, 18])
    outl = oracle.gaussian(imp, legacy=True)
    np.testing.assert_array_equal(outl[7:14, 7:14], np.minimum((np.outer(kl, kl) * 255 + 32768) >> 16, 255))


def test_fast_atan2(oracle):
    assert oracle.fast_atan2(0, 1) == 0.0
    assert oracle.fast_atan2(1, 0) == 90.0
    assert oracle.fast_atan2(0, -1) == 180.0
    assert oracle.fast_atan2(-1, 0) == 270.0
    f = np.float32
    k = f(180 / np.pi)
    p1, p3, p5, p7 = (f(c) * k for c in (f(0.9997878412794807), f(-0.3258083974640975), f(0.1555786518463281), f(-0.04432655554792128)))
    c = f(1) / (f(1) + f(2.220446049250313e-16))
    c2 = c * c
    expect = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    got = oracle.fast_atan2(1, 1)
    assert got == expect and abs(got - 44.9905) < 1e-3 and got != 45.0
    rng = np.random.default_rng(1)
    for y, x in rng.integers(-3000000, 3000000, size=(2000, 2)):
        a = oracle.fast_atan2(float(y), float(x))
        true = np.degrees(np.arctan2(float(y), float(x))) % 360.0
        assert min(abs(a - true), 360 - abs(a - true)) < 0.02


def test_sincos_portable_accuracy(oracle):
    xs = np.linspace(0, 2 * np.pi * 1.001, 20001).astype(np.float32)
    for x in xs[::7]:
        s, c = oracle.sincos(float(x))
        # correctly rounded (or off by the double-rounding case, < 1 ulp)
        assert abs(np.float64(s) - np.sin(np.float64(x))) <= np.spacing(np.float32(abs(s))) * 0.5000001 + 1e-45
        assert abs(np.float64(c) - np.cos(np.float64(x))) <= np.spacing(np.float32(abs(c))) * 0.5000001 + 1e-45
    assert oracle.sincos(0.0) == (0.0, 1.0)


def test_hamming(oracle):
    d = synth_descriptors(300, 7)
    for i in range(0, 299):
        a, b = d[i], d[i + 1]
        pop = int(np.unpackbits(a ^ b).sum())
        assert oracle.hamming(a, b, bithack=True) == pop
        assert oracle.hamming(a, b, bithack=False) == pop
    assert oracle.hamming(d[0], d[0]) == 0
    assert oracle.hamming(d[0], ~d[0]) == 256


def _flat(v=100, n=31):
    return np.full((n, n), v, np.uint8)


def test_fast_primitive(oracle):
    img = _flat()
    assert len(oracle.fast(img, 20)) == 0
    img[15, 15] = 150  # isolated bright pixel: all 16 ring pixels darker by 50
    k = oracle.fast(img, 20)
    assert len(k) == 1 and (k[0]["x"], k[0]["y"]) == (15, 15)
    assert k[0]["response"] == 49 and k[0]["size"] == 7 and k[0]["angle"] == -1
    assert oracle.fast_score(img, 15, 15, 20) == 49
    assert oracle.fast_score(img, 15, 15, 7) == 49  # independent of the threshold
    # step edge: no corner
    edge = _flat()
    edge[:, 16:] = 200
    assert len(oracle.fast(edge, 20)) == 0
    # bright quadrant: corner at the apex
    quad = _flat()
    quad[15:, 15:] = 200
    k = oracle.fast(quad, 20, nms=False)  # (neighbours tie on score, so NMS would drop the apex)
    assert any((p["x"], p["y"]) == (15, 15) for p in k)
    # border: pixels closer than 3 to the ROI edge are never reported
    b = _flat()
    b[2, 2] = 200
    assert len(oracle.fast(b, 20)) == 0
    # tiny ROI
    assert len(oracle.fast(np.zeros((6, 30), np.uint8), 5)) == 0


def test_fast_score_is_max_threshold(oracle):
    """score == largest t at which the pixel is still a corner (so one dense
    score map serves both the iniThFAST and the minThFAST pass)."""
    img = synth_scene(96, 80, 11)
    k7 = oracle.fast(img, 7, nms=False)
    assert len(k7) > 20
    for p in k7[:: max(1, len(k7) // 60)]:
        x, y = int(p["x"]), int(p["y"])
        s = oracle.fast_score(img, x, y, 7)
        pts = lambda t: {(int(q["x"]), int(q["y"])) for q in oracle.fast(img, t, nms=False)}
        assert (x, y) in pts(s)
        assert (x, y) not in pts(s + 1)


def test_three_maxima(oracle):
    c = [0] * 30
    assert oracle.three_maxima(c) == (-1, -1, -1)
    c[3], c[7], c[20] = 10, 50, 30
    assert oracle.three_maxima(c) == (7, 20, 3)
    c[3] = 4  # below 10% of the maximum
    assert oracle.three_maxima(c) == (7, 20, -1)
    c[20] = 4
    assert oracle.three_maxima(c) == (7, -1, -1)
    c = [5] * 30  # ties: strict '>' keeps the first three
    assert oracle.three_maxima(c) == (0, 1, 2)


def _keys(pts, resp=None):
    import oracle as O

    k = np.zeros(len(pts), dtype=O.KP_DTYPE)
    for i, (x, y) in enumerate(pts):
        k[i] = (x, y, 7, -1, (resp[i] if resp else 10 + i), 0, -1)
    return k


@pytest.mark.parametrize("quirks", ["fork", "canonical"])
def test_quadtree_small_cases(oracle, quirks):
    q = oracle.QUIRKS_FORK if quirks == "fork" else oracle.QUIRKS_CANONICAL
    e = oracle.Extractor(quirks=q)
    # one key
    r = e.distribute(_keys([(100, 50)]), 16, 624, 16, 464, 217)
    assert len(r) == 1 and (r[0]["x"], r[0]["y"]) == (100, 50)
    # four well separated keys, quota 4
    pts = [(10, 10), (500, 10), (10, 400), (500, 400)]
    r = e.distribute(_keys(pts), 16, 624, 16, 464, 4)
    assert sorted((int(p["x"]), int(p["y"])) for p in r) == sorted(pts)
    # two keys in one final node: the larger response survives, first wins on ties
    r = e.distribute(_keys([(10, 10), (11, 10), (500, 400)], [5, 9, 1]), 16, 624, 16, 464, 2)
    assert sorted(int(p["response"]) for p in r) == [1, 9]
    r = e.distribute(_keys([(10, 10), (11, 10), (500, 400)], [9, 9, 1]), 16, 624, 16, 464, 2)
    assert sorted((int(p["x"]), int(p["response"])) for p in r) == [(10, 9), (500, 1)]


def test_quadtree_respects_quota_lower_bound(oracle):
    img = synth_scene(640, 480, frame_seed(1, 3))
    for q in (oracle.QUIRKS_FORK, oracle.QUIRKS_CANONICAL):
        e = oracle.Extractor(quirks=q)
        e.extract(img)
        per = e.tables()["per_level"]
        for l in range(8):
            n_in, n_out = len(e.level_keys(l, False)), len(e.level_keys(l, True))
            assert n_out <= n_in
            if q == oracle.QUIRKS_CANONICAL and n_in >= per[l] * 4:
                assert per[l] <= n_out <= per[l] + 3


def test_brute_force_top2_ties(oracle):
    db = synth_descriptors(500, 3)
    q = db[[10, 20]].copy()
    db[400] = db[10]  # exact duplicate later in the scan
    idx, d1, d2 = oracle.bf_top2(q, db)
    assert idx[0] == 10 and d1[0] == 0 and d2[0] == 0  # tie for best: second == best, lower index wins
    assert idx[1] == 20 and d1[1] == 0 and d2[1] > 0


def test_distinctive_descriptor_known_answers(oracle):
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:186-214) on cases worked by hand."""
    z = np.zeros(32, np.uint8)

    def d(nbits):  # descriptor with the lowest `nbits` bits set: distance to z is nbits
        v = np.zeros(32, np.uint8)
        v[: nbits // 8] = 255
        if nbits % 8:
            v[nbits // 8] = (1 << (nbits % 8)) - 1
        return v

    # one observation: itself, median 0
    assert oracle.distinctive_descriptor(np.stack([d(9)])) == (0, 0)
    # two observations: rows {0, 9} and {9, 0}; k = (size_t)(0.5 * 1) = 0 -> both medians 0, first wins
    assert oracle.distinctive_descriptor(np.stack([z, d(9)])) == (0, 0)
    # three: distances z-d(4) = 4, z-d(10) = 10, d(4)-d(10) = 6; sorted rows {0,4,10} {0,4,6} {0,6,10}, k = 1:
    # medians 4, 4, 6 -> row 0 (strict '<' keeps the first)
    assert oracle.distinctive_descriptor(np.stack([z, d(4), d(10)])) == (0, 4)
    # four: k = (size_t)1.5 = 1; rows of d(0), d(8), d(10), d(40): {0,8,10,40} {0,2,8,32} {0,2,10,30} {0,30,32,40}
    # medians 8, 2, 2, 30 -> row 1
    assert oracle.distinctive_descriptor(np.stack([z, d(8), d(10), d(40)])) == (1, 2)
    # the fork's self-comparison at :195 makes every distance 0
    assert oracle.distinctive_descriptor(np.stack([d(40), d(8), z]), self_distance=True) == (0, 0)
    assert oracle.distinctive_descriptor(np.zeros((0, 32), np.uint8)) == (-1, None)


def test_undistortion_known_answers(oracle):
    """Frame::UndistortKeyPoints / ComputeImageBounds through cv::undistortPoints(.., K, D, noArray(), K)."""
    K = (517.306408, 516.469215, 318.643040, 255.313989)      # Examples/Monocular/TUM1.yaml:8-11
    D = (0.262383, -0.953104, -0.005358, 0.002628, 1.163314)  # :13-17
    # no distortion: bounds are the image rectangle, keypoints are copied
    assert list(oracle.compute_image_bounds(640, 480, K, (0, 0, 0, 0, 0))) == [0.0, 640.0, 0.0, 480.0]
    kps = np.zeros(3, oracle.KP_DTYPE)
    kps["x"], kps["y"], kps["octave"] = [10.5, 320, 600], [20.25, 240, 470], [0, 3, 7]
    assert (oracle.undistort_keypoints(kps, K, (0, 0, 0, 0, 0)) == kps).all()
    # the principal point is a fixed point of the model; every other field is copied
    pp = oracle.undistort_points([[K[2], K[3]]], K, D)
    assert abs(pp[0, 0] - np.float32(K[2])) < 1e-4 and abs(pp[0, 1] - np.float32(K[3])) < 1e-4
    un = oracle.undistort_keypoints(kps, K, D)
    assert (un["octave"] == kps["octave"]).all() and (un["x"] != kps["x"]).any()
    # forward model applied to the result returns the input (the five fixed-point iterations converge near the axis)
    x = (np.float64(un["x"][1]) - K[2]) / K[0]
    y = (np.float64(un["y"][1]) - K[3]) / K[1]
    r2 = x * x + y * y
    rad = 1 + D[0] * r2 + D[1] * r2 * r2 + D[4] * r2 ** 3
    xd = x * rad + 2 * D[2] * x * y + D[3] * (r2 + 2 * x * x)
    yd = y * rad + D[2] * (r2 + 2 * y * y) + 2 * D[3] * x * y
    assert abs(xd * K[0] + K[2] - 320) < 1e-3 and abs(yd * K[1] + K[3] - 240) < 1e-3
    # bounds: min of the left corners' x, max of the right corners' x, likewise for y
    b = oracle.compute_image_bounds(640, 480, K, D)
    c = oracle.undistort_points([[0, 0], [640, 0], [0, 480], [640, 480]], K, D)
    assert b[0] == min(c[0, 0], c[2, 0]) and b[1] == max(c[1, 0], c[3, 0])
    assert b[2] == min(c[0, 1], c[1, 1]) and b[3] == max(c[2, 1], c[3, 1])
