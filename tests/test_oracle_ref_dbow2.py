"""The oracle's bag-of-words bookkeeping against the reference ITSELF: BowVector::addWeight / addIfNotExist /
normalize (Thirdparty/DBoW2/DBoW2/BowVector.cpp:40-95) and FeatureVector::addFeature (FeatureVector.cpp:37-52),
compiled from the reference's files as they lie into oracle/_ref/libdbow2_ref.so (oracle/ref_dbow2/Makefile), fed the
same per-feature (word id, weight, node id) streams as oracle/orb_oracle_bow.cc's bowAccumulate, in the call order of
TemplatedVocabulary::transform (TemplatedVocabulary.h:1136-1206).  Values must be bit-equal (same double operations
in the same order).  This pins the bookkeeping half of the vocabulary row only; the tree descent and everything on
the extraction and matching path stay unpinned (no part of them compiles without OpenCV)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.skipif(not oracle.ref_dbow2_available(),
                                reason="oracle/_ref/libdbow2_ref.so not built (needs /root/reference at build time)")

SCORINGS = {0: "L1_NORM", 1: "L2_NORM", 2: "CHI_SQUARE", 3: "KL", 4: "BHATTACHARYYA", 5: "DOT_PRODUCT"}
WEIGHTINGS = {0: "TF_IDF", 1: "TF", 2: "IDF", 3: "BINARY"}


def stream(rng, n, nwords, nnodes, stopped=0.1):
    ids = rng.integers(0, nwords, n).astype(np.int32)
    nids = rng.integers(1, nnodes + 1, n).astype(np.int32)
    w = rng.uniform(0.01, 9.0, n)
    w[rng.random(n) < stopped] = 0.0   # stopped words (weight 0) are skipped by transform()
    return ids, w, nids


def same(a, b):
    (ia, va), (na, oa, xa) = a
    (ib, vb), (nb, ob, xb) = b
    assert (ia == ib).all() and len(ia) == len(ib)
    assert va.tobytes() == vb.tobytes()          # bit-equal doubles
    assert (na == nb).all() and (oa == ob).all() and (xa == xb).all()


@pytest.mark.parametrize("weighting", sorted(WEIGHTINGS))
@pytest.mark.parametrize("scoring", sorted(SCORINGS))
def test_streams_match_reference_dbow2(scoring, weighting):
    rng = np.random.default_rng(1000 * scoring + weighting)
    for n, nwords, nnodes in ((0, 10, 5), (1, 10, 5), (7, 3, 2), (500, 40, 9), (2000, 100000, 1000), (2000, 50, 1000)):
        ids, w, nids = stream(rng, n, nwords, nnodes)
        same(oracle.bow_accumulate(ids, w, nids, scoring, weighting), oracle.ref_bow_accumulate(ids, w, nids, scoring, weighting))


def test_all_words_stopped_and_repeated_word():
    ids = np.array([4, 4, 4, 2], np.int32)
    nids = np.array([9, 9, 3, 3], np.int32)
    for weighting in WEIGHTINGS:
        for scoring in SCORINGS:
            z = np.zeros(4)
            same(oracle.bow_accumulate(ids, z, nids, scoring, weighting), oracle.ref_bow_accumulate(ids, z, nids, scoring, weighting))
            w = np.array([1.5, 2.5, 0.25, 3.0])
            a = oracle.bow_accumulate(ids, w, nids, scoring, weighting)
            same(a, oracle.ref_bow_accumulate(ids, w, nids, scoring, weighting))
            assert list(a[0][0]) == [2, 4] and list(a[1][0]) == [3, 9] and list(a[1][2]) == [2, 3, 0, 1]


def test_vocabulary_transform_goes_through_the_checked_function():
    """Vocabulary.transform = descent + bowAccumulate: its output equals the reference-compiled bookkeeping applied
    to the descent's per-feature triples."""
    rng = np.random.default_rng(5)
    k, L = 3, 3
    parent, leaf = [], []
    level = [0]
    nid = 0
    for d in range(L):
        nxt = []
        for p in level:
            for _ in range(k):
                nid += 1
                parent.append(p)
                leaf.append(1 if d == L - 1 else 0)
                nxt.append(nid)
        level = nxt
    desc = rng.integers(0, 256, (nid, 32), dtype=np.uint8)
    weight = rng.uniform(0.0, 5.0, nid)
    weight[rng.random(nid) < 0.1] = 0.0
    for scoring, weighting in ((0, 0), (1, 1), (5, 0), (0, 3), (5, 2)):
        voc = oracle.Vocabulary.from_arrays(k, L, scoring, weighting, parent, leaf, desc, weight)
        feats = rng.integers(0, 256, (300, 32), dtype=np.uint8)
        word, w, node = voc.transform_features(feats, 2)
        same(voc.transform(feats, 2), oracle.ref_bow_accumulate(word, w, node, scoring, weighting))
