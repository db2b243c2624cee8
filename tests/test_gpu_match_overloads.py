"""GPU parity of the matcher overloads the reference declares without a body (include/ORBmatcher.h:39,
110,133-184): the local-map projection search, the candidate loop of Fuse / SearchBySim3, the
keyframe-to-keyframe SearchByBoW and SearchForTriangulation, through the C ABI against the CPU oracle.
Indices bit-exact."""
import numpy as np
import pytest

from orb_slam2_amd.synth import frame_seed, splitmix64, synth_descriptors, synth_scene, synth_vocabulary

pytestmark = pytest.mark.gpu

PRESETS = {"fork": 0x1F, "canonical": 0}


@pytest.fixture(scope="module")
def hip():
    from orb_slam2_amd import binding

    binding.lib()
    return binding


def _frames(oracle, w, h, nfeat, seed, preset):
    e = oracle.Extractor(nfeat, 1.2, 8, 20, 7, PRESETS[preset])
    k1, d1 = e.extract(synth_scene(w, h, seed))
    k2, d2 = e.extract(synth_scene(w, h, seed, shift=(3, 2), noise_seed=seed + 1))
    return e, k1, d1, k2, d2


def _queries(k1, scale, seed, th):
    """Projections of frame-1 keypoints with a pixel of jitter, predicted level = the keypoint's own."""
    nq = len(k1)
    r = splitmix64(seed, nq * 3)
    jitter = ((r[: 2 * nq] % np.uint64(7)).astype(np.float32) - 3).reshape(nq, 2)
    u = k1["x"] + 3 + jitter[:, 0]
    v = k1["y"] + 2 + jitter[:, 1]
    lvl = k1["octave"].astype(np.int32)
    radius = (th * scale[lvl]).astype(np.float32)
    return u, v, radius, lvl, r[2 * nq:]


@pytest.mark.parametrize("preset", ["fork", "canonical"])
def test_search_local_map(hip, oracle, preset):
    w, h = 640, 480
    e, k1, d1, k2, d2 = _frames(oracle, w, h, 1000, frame_seed(1, 60), preset)
    scale = e.tables()["scale"]
    c = hip.Context(1000, 1.2, 8, 20, 7, w, h, 1, 0, PRESETS[preset])
    total = 0
    for th, ratio, th_high in ((4.0, 0.8, 100), (10.0, 0.8, 100), (2.5, 0.6, 60), (25.0, 0.95, 100)):
        u, v, radius, lvl, r = _queries(k1, scale, 99 + int(th), th)
        u[::41] = -7.0  # outside the image bounds
        qobs = (r % np.uint64(6) != 0).astype(np.uint8)
        occ = (splitmix64(17, len(k2)) % np.uint64(9) == 0).astype(np.uint8)
        for o2, qo in ((occ, qobs), (None, None)):
            ng, mg = c.match_local_map(u, v, radius, lvl - 1, lvl, d1, qo, k2, d2, w, h, o2, th_high, ratio)
            no, mo = oracle.search_local_map(u, v, radius, lvl - 1, lvl, d1, qo, k2, d2, w, h, o2, th_high, ratio, PRESETS[preset])
            np.testing.assert_array_equal(mg, mo, err_msg="th=%g ratio=%g" % (th, ratio))
            assert ng == no
            total += no
    assert total > 400
    # the ratio test really decides: with it off (second best never on the same level) more queries pass
    u, v, radius, lvl, r = _queries(k1, scale, 5, 25.0)
    n_on, _ = c.match_local_map(u, v, radius, lvl - 1, lvl, d1, None, k2, d2, w, h, None, 100, 0.5)
    n_off, _ = c.match_local_map(u, v, radius, lvl - 1, lvl, d1, None, k2, d2, w, h, None, 100, 10.0)
    assert n_on < n_off
    # no queries, no keypoints
    n, m = c.match_local_map(u[:0], v[:0], radius[:0], lvl[:0], lvl[:0], d1[:0], None, k2, d2, w, h)
    assert n == 0 and (m == -1).all()
    n, m = c.match_local_map(u, v, radius, lvl - 1, lvl, d1, None, k2[:0], d2[:0], w, h)
    assert n == 0 and len(m) == 0
    c.close()


@pytest.mark.parametrize("preset", ["fork", "canonical"])
def test_search_best_fuse_and_sim3(hip, oracle, preset):
    w, h = 752, 480
    e, k1, d1, k2, d2 = _frames(oracle, w, h, 2000, frame_seed(3, 9), preset)
    tabs = e.tables()
    scale, inv_sigma2 = tabs["scale"], tabs["inv_sigma2"]
    c = hip.Context(2000, 1.2, 8, 20, 7, w, h, 1, 0, PRESETS[preset])
    kept = 0
    for th_win, th, sig, chi2 in ((3.0, 50, inv_sigma2, 5.99), (3.0, 50, None, 5.99), (7.5, 100, None, 5.99),
                                  (12.0, 50, inv_sigma2, 40.0), (3.0, 50, inv_sigma2[:3], 5.99)):
        u, v, radius, lvl, _ = _queries(k1, scale, 1234 + th, th_win)
        ig, dg = c.match_best(u, v, radius, lvl - 1, lvl, d1, k2, d2, w, h, sig, chi2, th)
        io, do = oracle.search_best(u, v, radius, lvl - 1, lvl, d1, k2, d2, w, h, sig, chi2, th, PRESETS[preset])
        np.testing.assert_array_equal(ig, io)
        np.testing.assert_array_equal(dg, do)
        kept += int((io >= 0).sum())
    assert kept > 800
    # the gate removes candidates: fewer queries keep a match with it than without
    u, v, radius, lvl, _ = _queries(k1, scale, 8, 12.0)
    with_gate, _ = c.match_best(u, v, radius, lvl - 1, lvl, d1, k2, d2, w, h, inv_sigma2, 0.5, 100)
    without, _ = c.match_best(u, v, radius, lvl - 1, lvl, d1, k2, d2, w, h, None, 0.5, 100)
    assert (with_gate >= 0).sum() < (without >= 0).sum()
    ig, dg = c.match_best(u, v, radius, lvl - 1, lvl, d1, k2[:0], d2[:0], w, h)
    assert (ig == -1).all() and (dg == 256).all()
    ig, dg = c.match_best(u[:0], v[:0], radius[:0], lvl[:0], lvl[:0], d1[:0], k2, d2, w, h)
    assert len(ig) == 0
    c.close()


@pytest.mark.parametrize("check_ori", [True, False])
def test_search_by_bow_keyframes(hip, oracle, check_ori):
    w, h = 752, 480
    e = oracle.Extractor(2000, 1.2, 8, 20, 7, oracle.QUIRKS_FORK)
    k1, d1 = e.extract(synth_scene(w, h, frame_seed(3, 2)))
    k2, d2 = e.extract(synth_scene(w, h, frame_seed(3, 2), shift=(2, 3), noise_seed=5))
    arrays = synth_vocabulary(10, 4, seed=41, prune=0.05, stop=0.02)
    g = hip.Vocabulary.from_arrays(10, 4, 0, 0, *arrays)
    _, fv1 = g.transform(d1, 4)
    _, fv2 = g.transform(d2, 4)
    rng = np.random.default_rng(6)
    v1 = (rng.random(len(k1)) < 0.8).astype(np.uint8)
    v2 = (rng.random(len(k2)) < 0.7).astype(np.uint8)
    c = hip.Context(2000, 1.2, 8, 20, 7, w, h, 1, 0, 0x1F)
    for ratio in (0.6, 0.75, 0.9):
        n_ref, m_ref = oracle.search_by_bow_keyframes(k1, d1, v1, fv1, k2, d2, v2, fv2, ratio, check_ori)
        n_got, m_got = c.match_bow_keyframes(k1, d1, v1, fv1, k2, d2, v2, fv2, ratio, check_ori)
        assert n_got == n_ref
        np.testing.assert_array_equal(m_got, m_ref)
        assert (v2[m_ref[m_ref >= 0]] == 1).all() and (v1[np.nonzero(m_ref >= 0)[0]] == 1).all()
    assert n_ref > 30
    # strict threshold: a pair at distance exactly TH_LOW = 50 is taken by the frame form and not by this one
    a = synth_descriptors(2, 3)
    b = a.copy()
    bits = np.arange(50)
    np.bitwise_xor.at(b[0], bits >> 3, (1 << (bits & 7)).astype(np.uint8))
    kp = np.zeros(2, oracle.KP_DTYPE)
    fv = ([7], [0, 2], [0, 1])
    ones = np.ones(2, np.uint8)
    n_kk, m_kk = c.match_bow_keyframes(kp, a, ones, fv, kp, b, ones, fv, 0.9, False)
    n_ref, m_ref = oracle.search_by_bow_keyframes(kp, a, ones, fv, kp, b, ones, fv, 0.9, False)
    n_kf, m_kf = c.match_bow(kp, a, ones, fv, kp, b, fv, 0.9, False)
    assert n_kk == n_ref and list(m_kk) == list(m_ref) == [-1, 1]
    assert n_kf == 2 and list(m_kf) == [0, 1]
    c.close()
    g.close()


@pytest.mark.parametrize("mark_matched", [False, True])
def test_search_for_triangulation(hip, oracle, mark_matched):
    """Two views that differ by an image-plane translation t: F12 = [[0,0,-ty],[0,0,tx],[ty,-tx,0]], every
    epipolar line runs along t through the first keypoint."""
    w, h = 752, 480
    e = oracle.Extractor(2000, 1.2, 8, 20, 7, oracle.QUIRKS_FORK)
    k1, d1 = e.extract(synth_scene(w, h, frame_seed(3, 2)))
    k2, d2 = e.extract(synth_scene(w, h, frame_seed(3, 2), shift=(2, 3), noise_seed=5))
    arrays = synth_vocabulary(10, 4, seed=41, prune=0.05, stop=0.02)
    g = hip.Vocabulary.from_arrays(10, 4, 0, 0, *arrays)
    tabs = e.tables()
    rng = np.random.default_rng(16)
    f1 = (rng.random(len(k1)) < 0.7).astype(np.uint8)
    f2 = (rng.random(len(k2)) < 0.7).astype(np.uint8)
    c = hip.Context(2000, 1.2, 8, 20, 7, w, h, 1, 0, 0x1F)
    dense = (float(np.median(k2["x"])), float(np.median(k2["y"])))
    total = 0
    for levelsup in (4, 2):
        _, fv1 = g.transform(d1, levelsup)
        _, fv2 = g.transform(d2, levelsup)
        for (tx, ty), ep, ori in (((2, 3), None, True), ((3, 2), dense, True), ((-2, -3), (300.0, 200.0), False),
                                  ((0, 0), None, True)):
            F = np.array([[0, 0, -ty], [0, 0, tx], [ty, -tx, 0]], np.float32)
            n_ref, m_ref = oracle.search_for_triangulation(k1, d1, f1, fv1, k2, d2, f2, fv2, F, ep, tabs["scale"],
                                                           tabs["sigma2"], mark_matched, ori)
            n_got, m_got = c.match_triangulation(k1, d1, f1, fv1, k2, d2, f2, fv2, F, ep, tabs["scale"], tabs["sigma2"],
                                                 mark_matched, ori)
            assert n_got == n_ref
            np.testing.assert_array_equal(m_got, m_ref)
            if (tx, ty) == (0, 0):
                assert n_ref == 0  # a zero matrix gives den == 0: no candidate passes
            total += n_ref
            hit = m_ref[m_ref >= 0]
            assert (f2[hit] == 1).all() and (f1[np.nonzero(m_ref >= 0)[0]] == 1).all()
            if mark_matched:
                assert len(np.unique(hit)) == len(hit)
    assert total > 1500
    # fewer levels in the tables than octaves in the frames: those candidates are left out, as in the oracle
    F = np.array([[0, 0, -3], [0, 0, 2], [3, -2, 0]], np.float32)
    n_ref, m_ref = oracle.search_for_triangulation(k1, d1, f1, fv1, k2, d2, f2, fv2, F, None, tabs["scale"][:3],
                                                   tabs["sigma2"][:3], mark_matched, True)
    n_got, m_got = c.match_triangulation(k1, d1, f1, fv1, k2, d2, f2, fv2, F, None, tabs["scale"][:3], tabs["sigma2"][:3],
                                         mark_matched, True)
    assert n_got == n_ref and (m_got == m_ref).all()
    n, m = c.match_triangulation(k1[:0], d1[:0], f1[:0], ([], [0], []), k2, d2, f2, fv2, F, None, tabs["scale"],
                                 tabs["sigma2"], mark_matched, True)
    assert n == 0 and len(m) == 0
    c.close()
    g.close()
