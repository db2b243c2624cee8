#!/usr/bin/env python3
"""Development aid: what the FAST kernel's cells did, per level, from a library built with -DFAST_PROFILE
(make -C orb_slam2_amd/csrc clean; make -C orb_slam2_amd/csrc EXTRA=-DFAST_PROFILE): pixels that pass the candidate
test, calls of the 64-lane scoring step and how full they run, corners, kept keys, second attempts."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402

for (w, h, nf, batch, cfg) in [(640, 480, 1000, 64, 1), (1241, 376, 2000, 32, 2), (3840, 2160, 4000, 4, 4)]:
    ctx = binding.Context(nf, 1.2, 8, 20, 7, w, h, batch, 0, binding.ORB_QUIRKS_FORK)
    frames = np.stack([synth_scene(w, h, frame_seed(cfg, f)) for f in range(batch)])
    d = torch.from_numpy(frames).cuda()
    cap = ctx.cap
    k = torch.zeros((batch, cap, 28), dtype=torch.uint8, device="cuda")
    ds = torch.zeros((batch, cap, 32), dtype=torch.uint8, device="cuda")
    cn = torch.zeros(batch, dtype=torch.int32, device="cuda")
    out = (C.c_ulonglong * (12 * 8))()
    binding.lib().orb_debug_fast_profile(out, 1)
    ctx.extract_batch_device(d, k, ds, cn)
    torch.cuda.synchronize()
    binding.lib().orb_debug_fast_profile(out, 0)
    v = np.array(list(out), dtype=np.float64).reshape(12, 8)
    print("%dx%d, %d frames, nFeatures %d" % (w, h, batch, nf))
    print("  level  cells/frame  px/cell  candidates/cell (%% of px)  chunks/cell  mean fill  corners/cell  kept/cell  second attempts (%% of cells, candidates/cell)")
    tot = v.sum(axis=0)
    for l in list(range(8)) + ["all"]:
        r = tot if l == "all" else v[l]
        if r[0] == 0:
            continue
        cand = r[2] + r[7]
        print("  %5s  %11.1f  %7.1f  %15.1f (%4.1f%%)  %11.2f  %8.1f%%  %12.1f  %9.1f  %6.1f%%, %.1f" % (
            l, r[0] / batch, r[1] / r[0], cand / r[0], 100 * cand / r[1], r[3] / r[0], 100 * cand / (64 * r[3]) if r[3] else 0,
            r[4] / r[0], r[5] / r[0], 100 * r[6] / r[0], r[7] / r[6] if r[6] else 0))
    ctx.close()
