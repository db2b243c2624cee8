"""GPU parity of the bag-of-words path (vocabulary descent, BowVector / FeatureVector assembly,
SearchByBoW) through the C ABI against the CPU oracle: word, node and feature indices and match
lists bit-exact, word weights and normalised BowVector values equal as doubles."""
import numpy as np
import pytest

from orb_slam2_amd.synth import frame_seed, synth_descriptors, synth_scene, synth_vocabulary, write_vocabulary_text

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from orb_slam2_amd import binding

    binding.lib()
    return binding


def both(hip, oracle, k, L, scoring, weighting, arrays):
    return hip.Vocabulary.from_arrays(k, L, scoring, weighting, *arrays), oracle.Vocabulary.from_arrays(k, L, scoring, weighting, *arrays)


@pytest.mark.parametrize("shape", [(10, 3, 0.0), (10, 4, 0.15), (3, 6, 0.2), (20, 2, 0.1), (1, 4, 0.0), (17, 3, 0.3)],
                         ids=lambda s: "k%d_L%d" % (s[0], s[1]))
def test_vocabulary_descent(hip, oracle, shape):
    k, L, prune = shape
    arrays = synth_vocabulary(k, L, seed=100 + k * L, prune=prune, stop=0.05)
    g, o = both(hip, oracle, k, L, 0, 0, arrays)
    assert (g.k, g.L, g.nnodes, g.nwords) == (o.k, o.L, o.nnodes, o.nwords)
    rng = np.random.default_rng(k + L)
    # random descriptors, node descriptors themselves (distance 0 on the way down) and near copies
    feats = np.concatenate([rng.integers(0, 256, (1500, 32), dtype=np.uint8), arrays[2][:: max(1, len(arrays[2]) // 400)],
                            arrays[2][::7][:300] ^ (1 << rng.integers(0, 8, (len(arrays[2][::7][:300]), 32))).astype(np.uint8)])
    for levelsup in (0, 1, 2, 4, L, L + 3):
        for got, ref, what in zip(g.transform_features(feats, levelsup), o.transform_features(feats, levelsup), ("word", "weight", "node")):
            np.testing.assert_array_equal(got, ref, err_msg="%s, levelsup %d" % (what, levelsup))
    for n in (0, 1, 15, 16, 17):
        for got, ref in zip(g.transform_features(feats[:n], 2), o.transform_features(feats[:n], 2)):
            np.testing.assert_array_equal(got, ref)
    g.close()


def test_vocabulary_device_entry_point(hip, oracle):
    import torch

    arrays = synth_vocabulary(10, 4, seed=3, prune=0.1)
    g, o = both(hip, oracle, 10, 4, 0, 0, arrays)
    feats = np.random.default_rng(9).integers(0, 256, (2000, 32), dtype=np.uint8)
    d = torch.from_numpy(feats).cuda()
    word = torch.zeros(2000, dtype=torch.int32, device="cuda")
    node = torch.zeros(2000, dtype=torch.int32, device="cuda")
    weight = torch.zeros(2000, dtype=torch.float64, device="cuda")
    g.transform_features_device(d, 2, word, weight, node)
    torch.cuda.synchronize()
    rw, rwt, rn = o.transform_features(feats, 2)
    np.testing.assert_array_equal(word.cpu().numpy(), rw)
    np.testing.assert_array_equal(weight.cpu().numpy(), rwt)
    np.testing.assert_array_equal(node.cpu().numpy(), rn)
    g.close()


@pytest.mark.parametrize("scoring,weighting", [(0, 0), (1, 0), (0, 1), (0, 2), (5, 1), (2, 3)])
def test_bow_and_feature_vectors(hip, oracle, scoring, weighting):
    arrays = synth_vocabulary(10, 4, seed=11, prune=0.1, stop=0.1)
    g, o = both(hip, oracle, 10, 4, scoring, weighting, arrays)
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, oracle.QUIRKS_FORK)
    _, desc = e.extract(synth_scene(640, 480, frame_seed(1, 3)))
    for d in (desc, desc[:1], desc[:0]):
        for levelsup in (4, 2):
            (gi, gv), gf = g.transform(d, levelsup)
            (oi, ov), of = o.transform(d, levelsup)
            np.testing.assert_array_equal(gi, oi)
            np.testing.assert_array_equal(gv, ov)   # doubles, same operations in the same order
            for a, b in zip(gf, of):
                np.testing.assert_array_equal(a, b)
    (gi, gv), _ = g.transform(desc, 4)
    (hi, hv), _ = g.transform(desc[::2], 4)
    assert hip.bow_score_l1((gi, gv), (hi, hv)) == oracle.score_l1((gi, gv), (hi, hv))
    assert hip.bow_score_l1((gi, gv), (gi, gv)) == oracle.score_l1((gi, gv), (gi, gv))
    if scoring == 0:
        assert abs(hip.bow_score_l1((gi, gv), (gi, gv)) - 1.0) < 1e-12
    g.close()


def test_text_file_loader(hip, oracle, tmp_path):
    arrays = synth_vocabulary(5, 3, seed=21, prune=0.25, stop=0.1)
    path = tmp_path / "voc.txt"
    write_vocabulary_text(path, 5, 3, 0, 0, *arrays)
    g = hip.Vocabulary.load(path)
    o = oracle.Vocabulary.load(path)
    assert (g.k, g.L, g.scoring, g.weighting, g.nnodes, g.nwords) == (o.k, o.L, o.scoring, o.weighting, o.nnodes, o.nwords)
    feats = np.random.default_rng(4).integers(0, 256, (500, 32), dtype=np.uint8)
    for got, ref in zip(g.transform_features(feats, 1), o.transform_features(feats, 1)):
        np.testing.assert_array_equal(got, ref)
    g.close()
    with open(tmp_path / "bad.txt", "w") as f:
        f.write("30 6 0 0\n")
    with pytest.raises(hip.OrbError):
        hip.Vocabulary.load(tmp_path / "bad.txt")
    with pytest.raises(hip.OrbError):
        hip.Vocabulary.load(tmp_path / "missing.txt")


@pytest.mark.parametrize("check_ori", [True, False])
@pytest.mark.parametrize("levelsup", [4, 2, 0])
def test_search_by_bow_on_frames(hip, oracle, check_ori, levelsup):
    """Keyframe = a synthetic frame, frame = the same scene shifted with new noise; the vocabulary is
    built around the keyframe's own descriptors so that corresponding features share nodes."""
    w, h = 752, 480
    e = oracle.Extractor(2000, 1.2, 8, 20, 7, oracle.QUIRKS_FORK)
    kk, kd = e.extract(synth_scene(w, h, frame_seed(3, 1)))
    fk, fd = e.extract(synth_scene(w, h, frame_seed(3, 1), shift=(3, 2), noise_seed=77))
    arrays = synth_vocabulary(10, 4, seed=31, prune=0.05, stop=0.02)
    g, o = both(hip, oracle, 10, 4, 0, 0, arrays)
    _, kfv = g.transform(kd, levelsup)
    _, ffv = g.transform(fd, levelsup)
    valid = (np.random.default_rng(5).random(len(kk)) < 0.8).astype(np.uint8)
    c = hip.Context(2000, 1.2, 8, 20, 7, w, h, 1, 0, 0x1F)
    for ratio in (0.6, 0.75, 0.9):
        n_ref, m_ref = oracle.search_by_bow(kk, kd, valid, kfv, fk, fd, ffv, ratio, check_ori)
        n_got, m_got = c.match_bow(kk, kd, valid, kfv, fk, fd, ffv, ratio, check_ori)
        assert n_got == n_ref
        np.testing.assert_array_equal(m_got, m_ref)
    if levelsup == 4:
        assert n_ref > 50, "the pair is meant to produce matches (%d)" % n_ref
    # empty sides and disjoint nodes
    n, m = c.match_bow(kk[:0], kd[:0], valid[:0], ([], [0], []), fk, fd, ffv, 0.7, check_ori)
    assert n == 0 and (m == -1).all()
    shifted = (kfv[0] + 1000000, kfv[1], kfv[2])
    n, m = c.match_bow(kk, kd, valid, shifted, fk, fd, ffv, 0.7, check_ori)
    assert n == 0 and (m == -1).all()
    with pytest.raises(hip.OrbError):  # a feature listed under two nodes
        c.match_bow(kk, kd, valid, ([1, 2], [0, 1, 2], [0, 0]), fk, fd, ffv, 0.7, check_ori)
    c.close()
    g.close()


def test_search_by_bow_one_large_node(hip, oracle):
    """Every feature under one node, more than 4096 of them: all positions of the taken flags are used
    and most keyframe features find their counterpart taken or not according to the order."""
    n = 8000
    rng = np.random.default_rng(12)
    kd = synth_descriptors(n, 99)
    perm = rng.permutation(n)
    fd = kd[perm].copy()
    flips = rng.integers(0, 40, n)
    for j in range(n):
        bits = rng.choice(256, size=flips[j], replace=False)
        np.bitwise_xor.at(fd[j], bits >> 3, (1 << (bits & 7)).astype(np.uint8))
    fd[::50] = fd[1::50][: len(fd[::50])]  # duplicates: equal distances, the first one wins
    kk = np.zeros(n, dtype=oracle.KP_DTYPE)
    fk = np.zeros(n, dtype=oracle.KP_DTYPE)
    kk["angle"] = rng.random(n).astype(np.float32) * 360
    fk["angle"] = (kk["angle"][perm] + rng.normal(0, 8, n).astype(np.float32)) % np.float32(360)
    valid = (rng.random(n) < 0.9).astype(np.uint8)
    fv = ([0], [0, n], np.arange(n))
    fv_f = ([0], [0, n], rng.permutation(n))
    c = hip.Context(1000, 1.2, 8, 20, 7, 640, 480, 1, 0, 0x1F)
    for check_ori in (True, False):
        n_ref, m_ref = oracle.search_by_bow(kk, kd, valid, fv, fk, fd, fv_f, 0.8, check_ori)
        n_got, m_got = c.match_bow(kk, kd, valid, fv, fk, fd, fv_f, 0.8, check_ori)
        assert n_got == n_ref and n_ref > 3000
        np.testing.assert_array_equal(m_got, m_ref)
    with pytest.raises(hip.OrbError):
        c.match_bow(np.zeros(8193, oracle.KP_DTYPE), np.zeros((8193, 32), np.uint8), np.ones(8193, np.uint8),
                    ([0], [0, 1], [0]), fk, fd, fv_f, 0.8, True)
    c.close()
