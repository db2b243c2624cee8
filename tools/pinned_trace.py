#!/usr/bin/env python3
"""A few orb_extract_batch calls on 64 page-locked frames, for a rocprofv3 --kernel-trace --memory-copy-trace run."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402

W, H, nb = 640, 480, int(os.environ.get("NB", "64"))
frames = [synth_scene(W, H, frame_seed(1, f % 64)) for f in range(nb)]
ctx = binding.Context(1000, 1.2, 8, 20, 7, W, H, nb, 0, binding.ORB_QUIRKS_FORK)
hk = np.zeros((nb, ctx.cap), binding.KP_DTYPE)
hd = np.zeros((nb, ctx.cap, 32), np.uint8)
hc = np.zeros(nb, np.int32)
pinned = binding.alloc_host((nb, H, W))
pinned[:] = np.stack(frames)
pin = [pinned[f] for f in range(nb)]
for _ in range(6):
    ctx.extract_batch_into(pin, hk, hd, hc)
ctx.close()
