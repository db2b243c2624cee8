#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle on seeded synthetic inputs.

The reference ships no golden vectors for this path and can be neither built
nor run (compile errors in its hot-path sources, OpenCV absent), so these
fixtures pin the ORACLE's behaviour (a change in the oracle or in the HIP path
shows up as a diff against them); they do not pin the oracle to the reference.
Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_descriptors, synth_queries, synth_scene  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def extraction_case(name, w, h, nf, nl, seed, quirks):
    img = synth_scene(w, h, seed)
    e = oracle.Extractor(nf, 1.2, nl, 20, 7, quirks)
    k, d = e.extract(img)
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"), w=w, h=h, nfeatures=nf, nlevels=nl, seed=seed, quirks=quirks,
        image_crc=crc(img), keypoints=k, descriptors=d,
        level_crc=np.array([crc(e.level(l)) for l in range(nl)], np.uint32),
        fast_counts=np.array([len(e.level_keys(l, False)) for l in range(nl)], np.int32),
        level_counts=np.array([len(e.level_keys(l, True)) for l in range(nl)], np.int32))


def main():
    for preset, q in (("fork", oracle.QUIRKS_FORK), ("canonical", oracle.QUIRKS_CANONICAL)):
        extraction_case("extract_160x120_%s" % preset, 160, 120, 300, 4, frame_seed(1, 0), q)
        extraction_case("extract_640x480_%s" % preset, 640, 480, 1000, 8, frame_seed(1, 0), q)
        # frame pair + SearchForInitialization
        w, h = 752, 480
        seed = frame_seed(3, 0)
        e = oracle.Extractor(2000, 1.2, 8, 20, 7, q)
        k1, d1 = e.extract(synth_scene(w, h, seed))
        k2, d2 = e.extract(synth_scene(w, h, seed, shift=(3, 2), noise_seed=seed + 1))
        prev = np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32)
        n, m12, pout = oracle.search_for_initialization(k1, d1, k2, d2, w, h, prev, 100, 0.9, True, q)
        np.savez_compressed(os.path.join(OUT, "match_752x480_%s.npz" % preset), w=w, h=h, seed=seed, quirks=q, k1=k1, d1=d1,
                            k2=k2, d2=d2, nmatches=n, matches12=m12, prev_out=pout)
    db = synth_descriptors(4096, 41)
    qq = synth_queries(db, 256, 42)
    db[3000:3010] = db[5:15]
    qq[:10] = db[5:15]
    idx, b, s = oracle.bf_top2(qq, db)
    np.savez_compressed(os.path.join(OUT, "bruteforce_256x4096.npz"), db_seed=41, q_seed=42, idx=idx, best=b, second=s)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
