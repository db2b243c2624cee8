"""Development aid: phase times of the quadtree kernel's first workgroup (frame 0, level 0) from a library built
with -DQT_PROFILE (make -C orb_slam2_amd/csrc clean; make -C orb_slam2_amd/csrc EXTRA=-DQT_PROFILE)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402

names = ["gather", "histogram+scatter", "rank small buckets", "sort big buckets (+sync)", "distribute", "select", "emit"]
for (w, h, nf, batch, cfg) in [(640, 480, 1000, 1, 1), (640, 480, 1000, 64, 1), (3840, 2160, 4000, 1, 4), (3840, 2160, 4000, 8, 4)]:
    ctx = binding.Context(nf, 1.2, 8, 20, 7, w, h, batch, 0, binding.ORB_QUIRKS_FORK)
    base = synth_scene(w, h, frame_seed(cfg, 0))
    imgs = [base] * batch
    for _ in range(4):
        ctx.extract_batch(imgs)
    out = (C.c_ulonglong * 32)()
    binding.lib().orb_debug_quadtree_profile(out, 1)
    ctx.extract_batch(imgs)
    binding.lib().orb_debug_quadtree_profile(out, 0)
    v = [int(x) for x in out]
    print("%dx%d batch %d: keys %d -> %d nodes, form %s, coarse passes %d, fine rounds %d, big buckets %d" %
          (w, h, batch, v[8], v[9], {0: "sorted (global)", 1: "sorted (LDS)", 2: "histogram"}[v[10]], v[16], v[17], v[18]))
    if v[10] == 2:
        st, nm = [0, 1, 4, 5, 6, 7], ["count into buckets", "scan", "distribute", "select", "emit"]
    else:
        st, nm = list(range(8)), names
    us = [(v[st[i + 1]] - v[st[i]]) / 100.0 for i in range(len(st) - 1)]
    print("   " + ", ".join("%s %.1f us" % (n, u) for n, u in zip(nm, us)) + ", total %.1f us" % ((v[7] - v[0]) / 100.0))
    t0 = v[4]
    print("   distribute: coarse passes start at " + ", ".join("%.1f" % ((x - t0) / 100.0) for x in v[20:26] if x) +
          " us; fine rounds at " + ", ".join("%.1f" % ((x - t0) / 100.0) for x in v[26:30] if x) +
          " us (recorded nodes %d, list %d); end %.1f us" % (v[11], v[12], (v[5] - t0) / 100.0))
    if v[15]:
        print("   shortcut over the first %d coarse passes (%d cells looked at): classified at %.1f us, positions at %.1f, list written at %.1f"
              % (v[30], v[31], (v[13] - t0) / 100.0, (v[14] - t0) / 100.0, (v[15] - t0) / 100.0))
    ctx.close()
