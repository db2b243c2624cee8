#!/usr/bin/env python3
"""Randomised differential test of the brute-force top-2 scan against the CPU oracle: random database and query
sizes (one row to a few hundred thousand, so one chunk to dozens of chunks that exchange thresholds), random row
numbering base, rows of extreme weight, exact and near copies of rows planted across the database, every case
queried twice (the second run sees other timing).  usage: PYTHONPATH=. python tools/fuzz_bf.py [iterations] [seed]"""
import sys

import numpy as np

import oracle
from orb_slam2_amd import binding


def one(it, rng, bf):
    kind = int(rng.integers(0, 4))
    nd = int(rng.integers(1, 400)) if kind == 0 else int(rng.integers(400, 30000)) if kind == 1 else int(rng.integers(30000, 400000))
    nq = int(rng.integers(1, 700)) if kind != 3 else int(rng.integers(700, 2500))
    db = rng.integers(0, 256, (nd, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    if rng.integers(0, 2):
        db[rng.integers(0, nd, max(1, nd // 50))] = 0
        db[rng.integers(0, nd, max(1, nd // 50))] = 255
    for k in range(min(nq, int(rng.integers(0, 300)))):
        home = int(rng.integers(0, nd))
        q[k] = db[home]
        for _ in range(int(rng.integers(0, 4))):       # exact copies: ties for the best
            db[int(rng.integers(0, nd))] = db[home]
        for _ in range(int(rng.integers(0, 3))):       # near copies
            r = int(rng.integers(0, nd))
            db[r] = db[home]
            for _ in range(int(rng.integers(1, 5))):
                db[r, int(rng.integers(0, 32))] ^= np.uint8(1 << int(rng.integers(0, 8)))
    base = int(rng.integers(0, 1 << 24))
    io, bo, so = oracle.bf_top2(q, db)
    io = np.where(io >= 0, io + base, io)
    bf.set_database(db, base)
    ok = True
    for run in range(2):
        ig, bg, sg = bf.query_top2(q)
        if not (np.array_equal(ig, io) and np.array_equal(bg, bo) and np.array_equal(sg, so)):
            bad = np.nonzero((ig != io) | (bg != bo) | (sg != so))[0]
            print("it %d run %d: %d rows x %d queries base %d: %d queries differ, first %d: gpu (%d, %d, %d) oracle (%d, %d, %d)"
                  % (it, run, nd, nq, base, len(bad), bad[0], ig[bad[0]], bg[bad[0]], sg[bad[0]], io[bad[0]], bo[bad[0]], so[bad[0]]))
            ok = False
    return ok, nd, nq


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    bf = binding.BruteForce(0)
    failed, rows, queries = 0, 0, 0
    for it in range(n):
        ok, nd, nq = one(it, rng, bf)
        failed += 0 if ok else 1
        rows += nd
        queries += nq
        if it % 20 == 19:
            print("%d iterations, %d failed" % (it + 1, failed), flush=True)
    bf.close()
    print("brute-force fuzz: %d iterations (seed %d, %d rows and %d queries in total, each case queried twice): %d failed"
          % (n, seed, rows, queries, failed))
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
