"""The quadtree device code (orb_slam2_amd/csrc/quadtree_core.h), built
single-threaded for the host, against the oracle's literal std::list
restatement of DistributeOctTree (src/ORBextractor.cc:571-796)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from orb_slam2_amd.synth import frame_seed, splitmix64, synth_scene

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    so = os.path.join(HERE, "emu", "libqt_emu.so")
    src = os.path.join(HERE, "emu", "quadtree_emu.cc")
    core = os.path.join(HERE, "..", "orb_slam2_amd", "csrc", "quadtree_core.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core)):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return C.CDLL(so)


def run_emu(emu, keys, min_x, max_x, min_y, max_y, N, n3quirk):
    xs = np.ascontiguousarray(keys["x"], dtype=np.int32)
    ys = np.ascontiguousarray(keys["y"], dtype=np.int32)
    sc = np.ascontiguousarray(keys["response"], dtype=np.int32)
    w, h = max_x - min_x, max_y - min_y
    n_ini = int(math.floor(np.float32(w) / np.float32(h) + np.float32(0.5)))  # round(), half away from zero
    hx = np.float32(w) / np.float32(n_ini)
    out = np.zeros(len(keys) + 16, dtype=np.int32)
    err = C.c_int(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    cnt = emu.qt_emu_distribute(p(xs), p(ys), p(sc), len(keys), n_ini, C.c_float(hx), h, N, int(n3quirk), p(out), len(out), C.byref(err))
    assert cnt >= 0 and err.value == 0
    return out[:cnt]


HIST = {"runs": 0, "deeper": 0}


def run_emu_hist(emu, keys, min_x, max_x, min_y, max_y, N, n3quirk, bucket_cap):
    """The histogram form; None when it reports that the level needs the sorted form."""
    xs = np.ascontiguousarray(keys["x"], dtype=np.int32)
    ys = np.ascontiguousarray(keys["y"], dtype=np.int32)
    sc = np.ascontiguousarray(keys["response"], dtype=np.int32)
    w, h = max_x - min_x, max_y - min_y
    n_ini = int(math.floor(np.float32(w) / np.float32(h) + np.float32(0.5)))
    hx = np.float32(w) / np.float32(n_ini)
    out = np.zeros(len(keys) + 16, dtype=np.int32)
    err = C.c_int(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    db = emu.qt_emu_hist_depth(n_ini, N, bucket_cap)
    cnt = emu.qt_emu_distribute_hist(p(xs), p(ys), p(sc), len(keys), n_ini, C.c_float(hx), h, N, int(n3quirk), db, p(out),
                                     len(out), C.byref(err))
    assert cnt >= -1 and err.value == 0
    HIST["runs"] += 1
    if cnt == -1:
        HIST["deeper"] += 1
        return None
    return out[:cnt]


def check(emu, oracle, e, keys, min_x, max_x, min_y, max_y, N, quirk):
    ref = e.distribute(keys, min_x, max_x, min_y, max_y, N)
    got = run_emu(emu, keys, min_x, max_x, min_y, max_y, N, quirk)
    assert len(got) == len(ref)
    np.testing.assert_array_equal(keys["x"][got], ref["x"])
    np.testing.assert_array_equal(keys["y"][got], ref["y"])
    np.testing.assert_array_equal(keys["response"][got], ref["response"])
    # the histogram form either gives the same list or asks for the sorted form
    for cap in (4096, 64):
        h = run_emu_hist(emu, keys, min_x, max_x, min_y, max_y, N, quirk, cap)
        if h is not None:
            np.testing.assert_array_equal(h, got, err_msg="histogram form, %d buckets at most" % cap)


@pytest.mark.parametrize("quirk", [True, False])
def test_emu_matches_oracle_on_fast_keys(emu, oracle, quirk):
    q = oracle.QUIRKS_FORK if quirk else oracle.QUIRKS_CANONICAL
    for (w, h, nf, cfg) in [(640, 480, 1000, 1), (1241, 376, 2000, 2), (752, 480, 2000, 3)]:
        e = oracle.Extractor(nf, 1.2, 8, 20, 7, q)
        for f in range(2):
            img = synth_scene(w, h, frame_seed(cfg, f))
            e.extract(img)
            per = e.tables()["per_level"]
            for l in range(8):
                keys = e.level_keys(l, after_quadtree=False)
                lw, lh = e.level(l).shape[::-1]
                check(emu, oracle, e, keys, 16, lw - 16, 16, lh - 16, int(per[l]), quirk)
                before = dict(HIST)
                check(emu, oracle, e, keys, 16, lw - 16, 16, lh - 16, int(per[l]), quirk)
                # at the level's own quota and 4096 buckets the histogram form must do (it is the fast path)
                assert HIST["deeper"] - before["deeper"] <= 1, "level %d of %dx%d fell back to the sorted form" % (l, w, h)
                # other quotas exercise early termination, the fine phase and exhaustion
                for N in (1, 7, 50, len(keys) // 2, len(keys) + 10):
                    check(emu, oracle, e, keys, 16, lw - 16, 16, lh - 16, max(1, N), quirk)


@pytest.mark.parametrize("quirk", [True, False])
def test_emu_matches_oracle_on_random_keys(emu, oracle, quirk):
    q = oracle.QUIRKS_FORK if quirk else oracle.QUIRKS_CANONICAL
    e = oracle.Extractor(1000, 1.2, 8, 20, 7, q)
    for trial in range(60):
        r = splitmix64(1234 + trial, 8)
        w = int(r[0] % np.uint64(900)) + 40
        h = int(r[1] % np.uint64(500)) + 40
        if w < h // 2 + 1:
            w = h
        n = int(r[2] % np.uint64(1500)) + 1
        N = int(r[3] % np.uint64(300)) + 1
        clustered = trial % 3 == 0
        rr = splitmix64(99 + trial, 3 * n)
        if clustered:  # many keys in a few tight clumps, equal responses
            xs = (rr[:n] % np.uint64(min(w, 12))).astype(np.int64) + int(r[4] % np.uint64(max(1, w - 12)))
            ys = (rr[n : 2 * n] % np.uint64(min(h, 12))).astype(np.int64) + int(r[5] % np.uint64(max(1, h - 12)))
            sc = (rr[2 * n :] % np.uint64(3)).astype(np.int64) + 7
        else:
            xs = (rr[:n] % np.uint64(w)).astype(np.int64)
            ys = (rr[n : 2 * n] % np.uint64(h)).astype(np.int64)
            sc = (rr[2 * n :] % np.uint64(200)).astype(np.int64) + 7
        # distinct pixels only (FAST emits at most one key per pixel)
        _, first = np.unique(xs * 100000 + ys, return_index=True)
        first.sort()
        keys = np.zeros(len(first), dtype=oracle.KP_DTYPE)
        keys["x"], keys["y"], keys["response"] = xs[first], ys[first], sc[first]
        keys["size"], keys["angle"], keys["class_id"] = 7, -1, -1
        emu.qt_emu_set_bin_bits([12, 11, 10, 6, 1][trial % 5])  # bucket counts a launch may choose, and tiny ones
        check(emu, oracle, e, keys, 16, 16 + w, 16, 16 + h, N, quirk)
    emu.qt_emu_set_bin_bits(12)
    assert 0 < HIST["deeper"] < HIST["runs"]  # both outcomes of the histogram form occurred


def test_the_shortcut_over_the_first_coarse_passes_ran(emu, oracle):
    """The histogram form evaluates the run of coarse passes that neither finish nor start the fine phase from the
    bucket counts alone (qt_jump).  The tests above compared its lists with the oracle's; this one makes sure they
    went through the shortcut, with every way it can end and over several depths."""
    test_emu_matches_oracle_on_random_keys(emu, oracle, True)  # (also when this test runs alone)
    st = np.zeros(16, dtype=np.int32)
    emu.qt_emu_jump_stats(st.ctypes.data_as(C.c_void_p), len(st))
    finished, fine, coarse = (int(v) for v in st[:3])
    assert finished > 0 and fine > 0 and coarse > 0, st
    by_k = st[3:]
    assert by_k[1] > 0 and by_k[2] > 0 and by_k[3] > 0 and by_k[4] > 0, st
