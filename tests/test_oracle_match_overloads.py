"""Known-answer tests of the oracle's restatements of the matcher overloads the reference only declares
(include/ORBmatcher.h:39,110,133-184).  The reference holds no body and no test vector for them, so the
expected values are worked out by hand from the rules of upstream ORB-SLAM2 that the fork's header
comments summarise: parity with the reference is unpinned for these functions."""
import numpy as np

W, H = 640, 480


def _desc_with_bits(n):
    """A 256-bit descriptor whose first n bits are set."""
    d = np.zeros(32, np.uint8)
    bits = np.arange(n)
    np.bitwise_or.at(d, bits >> 3, (1 << (bits & 7)).astype(np.uint8))
    return d


def _kps(oracle, rows):
    k = np.zeros(len(rows), oracle.KP_DTYPE)
    for i, (x, y, octave) in enumerate(rows):
        k[i]["x"], k[i]["y"], k[i]["octave"] = x, y, octave
    return k


def test_local_map_ratio_applies_on_equal_levels_only(oracle):
    q = np.zeros((1, 32), np.uint8)
    d2 = np.stack([_desc_with_bits(10), _desc_with_bits(11)])
    u, v, r = [100.0], [100.0], [5.0]
    # both candidates on level 1: 10 > 0.8 * 11 rejects the query
    k2 = _kps(oracle, [(100, 100, 1), (101, 100, 1)])
    n, m = oracle.search_local_map(u, v, r, [0], [1], q, None, k2, d2, W, H, None, 100, 0.8, 0)
    assert n == 0 and list(m) == [-1, -1]
    # second best on another level: no ratio test
    k2 = _kps(oracle, [(100, 100, 1), (101, 100, 0)])
    n, m = oracle.search_local_map(u, v, r, [0], [1], q, None, k2, d2, W, H, None, 100, 0.8, 0)
    assert n == 1 and list(m) == [0, -1]
    # second best outside the level range is not a candidate at all
    k2 = _kps(oracle, [(100, 100, 1), (101, 100, 2)])
    n, m = oracle.search_local_map(u, v, r, [0], [1], q, None, k2, d2, W, H, None, 100, 0.8, 0)
    assert n == 1 and list(m) == [0, -1]
    # the best one holds an observed map point: the other is alone and accepted
    k2 = _kps(oracle, [(100, 100, 1), (101, 100, 1)])
    n, m = oracle.search_local_map(u, v, r, [0], [1], q, None, k2, d2, W, H, [1, 0], 100, 0.8, 0)
    assert n == 1 and list(m) == [-1, 0]
    # above TH_HIGH
    n, m = oracle.search_local_map(u, v, r, [0], [1], q, None, k2, d2, W, H, [1, 0], 10, 0.8, 0)
    assert n == 0


def test_local_map_later_queries_see_earlier_assignments(oracle):
    q = np.zeros((2, 32), np.uint8)
    d2 = np.stack([_desc_with_bits(10), _desc_with_bits(30)])
    k2 = _kps(oracle, [(100, 100, 1), (101, 100, 1)])
    u, v, r = [100.0, 100.0], [100.0, 100.0], [5.0, 5.0]
    # map points with observations: the first query takes keypoint 0, the second finds it taken
    n, m = oracle.search_local_map(u, v, r, [0, 0], [1, 1], q, [1, 1], k2, d2, W, H, None, 100, 0.8, 0)
    assert n == 2 and list(m) == [0, 1]
    # a map point without observations does not block: the second query overwrites the first
    n, m = oracle.search_local_map(u, v, r, [0, 0], [1, 1], q, [0, 1], k2, d2, W, H, None, 100, 0.8, 0)
    assert n == 2 and list(m) == [1, -1]


def test_search_best_gate_and_ties(oracle):
    q = np.zeros((1, 32), np.uint8)
    inv_sigma2 = (1.0 / (1.2 ** np.arange(8)) ** 2).astype(np.float32)
    # two candidates 3 px away, equal distance: level 0 fails the gate (9 * 1 > 5.99), level 2 passes (9 * 0.48)
    k2 = _kps(oracle, [(103, 100, 0), (100, 103, 2)])
    d2 = np.stack([_desc_with_bits(5), _desc_with_bits(5)])
    idx, dist = oracle.search_best([100.0], [100.0], [6.0], [-1], [-1], q, k2, d2, W, H, inv_sigma2, 5.99, 50, 0)
    assert list(idx) == [1] and list(dist) == [5]
    # without the gate the first candidate in window order wins the tie: cells are visited column by
    # column, and keypoint 1 (x = 100) lies in the same column as or left of keypoint 0 (x = 103)
    idx, dist = oracle.search_best([100.0], [100.0], [6.0], [-1], [-1], q, k2, d2, W, H, None, 5.99, 50, 0)
    assert dist[0] == 5 and idx[0] in (0, 1)
    same_cell = _kps(oracle, [(101, 100, 0), (100, 100, 0)])
    idx, _ = oracle.search_best([100.0], [100.0], [6.0], [-1], [-1], q, same_cell, d2, W, H, None, 5.99, 50, 0)
    assert list(idx) == [0]  # same bucket: lower index first
    # threshold and empty window
    idx, dist = oracle.search_best([100.0], [100.0], [6.0], [-1], [-1], q, k2, d2, W, H, None, 5.99, 4, 0)
    assert list(idx) == [-1] and list(dist) == [5]
    idx, dist = oracle.search_best([400.0], [300.0], [6.0], [-1], [-1], q, k2, d2, W, H, None, 5.99, 50, 0)
    assert list(idx) == [-1] and list(dist) == [256]


def test_bow_keyframes_strict_threshold_and_masks(oracle):
    kp = np.zeros(2, oracle.KP_DTYPE)
    fv = ([7], [0, 2], [0, 1])
    ones = np.ones(2, np.uint8)
    a = np.zeros((2, 32), np.uint8)
    a[1] = 0xFF
    b = a.copy()
    b[0] = _desc_with_bits(50)  # distance exactly TH_LOW to a[0]
    n, m = oracle.search_by_bow_keyframes(kp, a, ones, fv, kp, b, ones, fv, 0.9, False)
    assert n == 1 and list(m) == [-1, 1]
    n_f, m_f = oracle.search_by_bow(kp, a, ones, fv, kp, b, fv, 0.9, False)
    assert n_f == 2 and list(m_f) == [0, 1]
    b[0] = _desc_with_bits(49)
    n, m = oracle.search_by_bow_keyframes(kp, a, ones, fv, kp, b, ones, fv, 0.9, False)
    assert n == 2 and list(m) == [0, 1]
    # second-side feature 0 without a map point: feature 0 of the first keyframe finds nothing close
    n, m = oracle.search_by_bow_keyframes(kp, a, ones, fv, kp, b, [0, 1], fv, 0.9, False)
    assert n == 1 and list(m) == [-1, 1]
    # a feature of keyframe 2 is given away once: two identical first-side features, one counterpart
    a2 = np.zeros((2, 32), np.uint8)
    b2 = np.stack([_desc_with_bits(3), np.full(32, 0xFF, np.uint8)])
    n, m = oracle.search_by_bow_keyframes(kp, a2, ones, fv, kp, b2, ones, fv, 0.9, False)
    assert n == 1 and list(m) == [0, -1]


def test_triangulation_epipolar_gate_and_last_among_equals(oracle):
    """View 2 = view 1 moved by (4, 0): epipolar lines are horizontal through the first keypoint."""
    F = np.array([[0, 0, 0], [0, 0, 4], [0, -4, 0]], np.float32)
    scale = (1.2 ** np.arange(8)).astype(np.float32)
    sigma2 = (scale * scale).astype(np.float32)
    k1 = _kps(oracle, [(100, 100, 0)])
    d1 = np.zeros((1, 32), np.uint8)
    fv1 = ([3], [0, 1], [0])
    ones = lambda n: np.ones(n, np.uint8)
    # candidates: on the line / 1 px off (1 < 3.84) / 3 px off (9 > 3.84) / 3 px off on level 4 (9 < 3.84 * 4.3)
    k2 = _kps(oracle, [(104, 100, 0), (104, 101, 0), (104, 103, 0), (104, 103, 4)])
    d2 = np.stack([_desc_with_bits(20), _desc_with_bits(10), _desc_with_bits(5), _desc_with_bits(8)])
    fv2 = ([3], [0, 4], [0, 1, 2, 3])
    n, m = oracle.search_for_triangulation(k1, d1, ones(1), fv1, k2, d2, ones(4), fv2, F, None, scale, sigma2, False, False)
    assert n == 1 and list(m) == [3]  # distance 8 on level 4 beats 10; the distance-5 one fails the line test
    n, m = oracle.search_for_triangulation(k1, d1, ones(1), fv1, k2[:3], d2[:3], ones(3), ([3], [0, 3], [0, 1, 2]), F, None,
                                           scale, sigma2, False, False)
    assert list(m) == [1]
    # equal distances: the later candidate replaces the earlier one
    d2e = np.stack([_desc_with_bits(10), _desc_with_bits(10)])
    n, m = oracle.search_for_triangulation(k1, d1, ones(1), fv1, k2[:2], d2e, ones(2), ([3], [0, 2], [0, 1]), F, None, scale,
                                           sigma2, False, False)
    assert list(m) == [1]
    # a candidate closer to the epipole than 10 px (level 0) is skipped: (104,101) is 9.5 px away, (104,100) 10.5 px
    n, m = oracle.search_for_triangulation(k1, d1, ones(1), fv1, k2[:2], d2e, ones(2), ([3], [0, 2], [0, 1]), F,
                                           (104.0, 110.5), scale, sigma2, False, False)
    assert list(m) == [0]
    # features that already hold a map point take no part, on either side
    n, m = oracle.search_for_triangulation(k1, d1, ones(1), fv1, k2[:2], d2e, [1, 0], ([3], [0, 2], [0, 1]), F, None, scale,
                                           sigma2, False, False)
    assert list(m) == [0]
    n, m = oracle.search_for_triangulation(k1, d1, [0], fv1, k2[:2], d2e, ones(2), ([3], [0, 2], [0, 1]), F, None, scale,
                                           sigma2, False, False)
    assert n == 0 and list(m) == [-1]
    # two first-side features, one counterpart: upstream gives it to both, with marking on to the first only
    k1b = _kps(oracle, [(100, 100, 0), (100, 100, 0)])
    d1b = np.zeros((2, 32), np.uint8)
    fv1b = ([3], [0, 2], [0, 1])
    n, m = oracle.search_for_triangulation(k1b, d1b, ones(2), fv1b, k2[:1], d2e[:1], ones(1), ([3], [0, 1], [0]), F, None,
                                           scale, sigma2, False, False)
    assert n == 2 and list(m) == [0, 0]
    n, m = oracle.search_for_triangulation(k1b, d1b, ones(2), fv1b, k2[:1], d2e[:1], ones(1), ([3], [0, 1], [0]), F, None,
                                           scale, sigma2, True, False)
    assert n == 1 and list(m) == [0, -1]
