#!/usr/bin/env python3
"""Randomised differential test of the HIP path against the CPU oracle: random frame sizes, feature
counts, level counts, scale factors, thresholds and presets; extraction, then the windowed searches on a
shifted copy.  usage: PYTHONPATH=. python tools/fuzz_parity.py [iterations] [seed]"""
import sys

import numpy as np

import oracle
from orb_slam2_amd import binding
from orb_slam2_amd.synth import splitmix64, synth_scene


def one(it, seed):
    r = splitmix64(seed * 1000003 + it, 16)
    w = int(r[0] % np.uint64(900)) + 120
    h = int(r[1] % np.uint64(600)) + 120
    nf = int(r[2] % np.uint64(2500)) + 50
    nl = int(r[3] % np.uint64(8)) + 1
    sf = [1.2, 1.1, 1.25, 1.3, 1.41, 1.15][int(r[4] % np.uint64(6))]
    ini = int(r[5] % np.uint64(40)) + 8
    mn = int(r[6] % np.uint64(ini)) + 1
    preset = 0x1F if r[7] & np.uint64(1) else 0
    # the coarsest level must hold one FAST cell plus borders
    while nl > 1 and min(w, h) / (sf ** (nl - 1)) < 70:
        nl -= 1
    tag = "it %d: %dx%d nf=%d nl=%d sf=%.2f th=%d/%d preset=%#x" % (it, w, h, nf, nl, sf, ini, mn, preset)
    try:
        c = binding.Context(nf, sf, nl, ini, mn, w, h, 1, 0, preset)
    except RuntimeError as e:
        print(tag, "-> refused:", e)
        return True
    e = oracle.Extractor(nf, sf, nl, ini, mn, preset)
    kind = int(r[8] % np.uint64(4))
    a = synth_scene(w, h, int(r[9] & np.uint64(0xFFFFFF)))
    if kind == 1:
        a = (splitmix64(int(r[10] & np.uint64(0xFFFFFF)), w * h) & np.uint64(255)).astype(np.uint8).reshape(h, w)
    elif kind == 2:
        a = (a // 8 + 100).astype(np.uint8)  # low contrast: minThFAST cells
    b = np.roll(a, (int(r[11] % np.uint64(7)) - 3, int(r[12] % np.uint64(7)) - 3), axis=(0, 1))
    ok = True
    res = []
    for img in (a, b):
        try:
            kg, dg = c.extract(img)
        except RuntimeError as ex:
            print(tag, "-> refused:", ex)
            c.close()
            return True
        ko, do = e.extract(img)
        if len(kg) != len(ko) or not (kg == ko).all() or not (dg == do).all():
            print(tag, "EXTRACTION MISMATCH", len(kg), len(ko))
            ok = False
        res.append((ko, do))
    (k1, d1), (k2, d2) = res
    if ok and it % 3 == 0:
        # the batch paths (a launch per pyramid level, frame groups, the quadtree's two passes) on the same geometry:
        # nine frames in one call must give what the one-frame calls gave
        cb = binding.Context(nf, sf, nl, ini, mn, w, h, 9, 0, preset)
        out = cb.extract_batch([a, b, a, b, b, a, a, b, a])
        for f, (kb, db_) in enumerate(out):
            ko, do = res[0] if f in (0, 2, 5, 6, 8) else res[1]
            if len(kb) != len(ko) or not (kb == ko).all() or not (db_ == do).all():
                print(tag, "BATCH MISMATCH frame", f)
                ok = False
        cb.close()
    if ok and len(k1) and len(k2) and len(k2) <= 8192:
        prev0 = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
        win = int(r[13] % np.uint64(150)) + 5
        no, mo, po = oracle.search_for_initialization(k1, d1, k2, d2, w, h, prev0, win, 0.9, True, preset)
        ng, mg, pg = c.match_window(k1, d1, k2, d2, w, h, prev0, win, 0.9, True)
        if ng != no or not (mg == mo).all() or not (pg == po).all():
            print(tag, "SEARCH-FOR-INITIALIZATION MISMATCH", ng, no)
            ok = False
        rad = np.float32(int(r[14] % np.uint64(30)) + 3) * (np.float32(sf) ** k1["octave"]).astype(np.float32)
        lvl = k1["octave"].astype(np.int32)
        qobs = (splitmix64(it, len(k1)) & np.uint64(3) != 0).astype(np.uint8)
        no, m2o = oracle.search_by_projection(k1["x"], k1["y"], rad, lvl - 1, lvl + 1, k1["angle"], d1, qobs, k2, d2, w, h, None, 100, True, preset)
        ng, m2g = c.match_projected(k1["x"], k1["y"], rad, lvl - 1, lvl + 1, k1["angle"], d1, qobs, k2, d2, w, h, None, 100, True)
        if ng != no or not (m2g == m2o).all():
            print(tag, "SEARCH-BY-PROJECTION MISMATCH", ng, no)
            ok = False
        # the overloads the reference only declares: local-map search, Fuse / SearchBySim3 candidate loop
        occ = (splitmix64(it + 7, len(k2)) % np.uint64(7) == 0).astype(np.uint8)
        ratio = [0.6, 0.8, 0.95][it % 3]
        u = k1["x"] + np.float32(int(r[11] % np.uint64(7)) - 3)
        v = k1["y"] + np.float32(int(r[12] % np.uint64(7)) - 3)
        no, m2o = oracle.search_local_map(u, v, rad, lvl - 1, lvl, d1, qobs, k2, d2, w, h, occ, 100, ratio, preset)
        ng, m2g = c.match_local_map(u, v, rad, lvl - 1, lvl, d1, qobs, k2, d2, w, h, occ, 100, ratio)
        if ng != no or not (m2g == m2o).all():
            print(tag, "LOCAL-MAP SEARCH MISMATCH", ng, no)
            ok = False
        sig = (1.0 / (np.float32(sf) ** np.arange(nl, dtype=np.float32)) ** 2).astype(np.float32) if it & 1 else None
        io, do_ = oracle.search_best(u, v, rad, lvl - 1, lvl, d1, k2, d2, w, h, sig, 5.99, 50 if it & 2 else 100, preset)
        ig, dg_ = c.match_best(u, v, rad, lvl - 1, lvl, d1, k2, d2, w, h, sig, 5.99, 50 if it & 2 else 100)
        if not (ig == io).all() or not (dg_ == do_).all():
            print(tag, "BEST-CANDIDATE SEARCH MISMATCH")
            ok = False
    c.close()
    if ok:
        print(tag, "ok (%d / %d keypoints)" % (len(k1), len(k2)), flush=True)
    return ok


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = sum(0 if one(i, seed) else 1 for i in range(n))
    print("%d of %d iterations failed" % (bad, n))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
