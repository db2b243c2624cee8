#!/usr/bin/env python3
"""orb_extract_batch on 64 frames of 640x480: pageable frames against page-locked ones (orb_alloc_host), wall time per call."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402

W, H = 640, 480
out = {}
for nb in (8, 64, 256):
    frames = [synth_scene(W, H, frame_seed(1, f % 64)) for f in range(nb)]
    ctx = binding.Context(1000, 1.2, 8, 20, 7, W, H, nb, 0, binding.ORB_QUIRKS_FORK)
    hk = np.zeros((nb, ctx.cap), binding.KP_DTYPE)
    hd = np.zeros((nb, ctx.cap, 32), np.uint8)
    hc = np.zeros(nb, np.int32)
    pinned = binding.alloc_host((nb, H, W))
    pinned[:] = np.stack(frames)
    pin = [pinned[f] for f in range(nb)]

    def wall(fn, steps=20, warm=3):
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        return (time.perf_counter() - t0) / steps

    tp = wall(lambda: ctx.extract_batch_into(frames, hk, hd, hc))
    tl = wall(lambda: ctx.extract_batch_into(pin, hk, hd, hc))
    out[str(nb)] = {"pageable_ms": round(tp * 1e3, 3), "pageable_frames_per_s": round(nb / tp, 1),
                    "page_locked_ms": round(tl * 1e3, 3), "page_locked_frames_per_s": round(nb / tl, 1)}
    ctx.close()
print(json.dumps(out))
