#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-pointer entry points (orb_extract, orb_extract_batch,
orb_match_window) on one MI355X: frames start and results end in host memory.  These figures
are documentation only; bench.py times the device-resident path."""
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (loads the HIP runtime before liborbhip)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402


def timed(fn, steps, warmup=3):
    for _ in range(warmup):
        fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    return (time.perf_counter() - t0) / steps


def main():
    w, h = 640, 480
    frames = [synth_scene(w, h, frame_seed(1, f)) for f in range(16)]
    out = {}
    ctx = binding.Context(1000, 1.2, 8, 20, 7, w, h, 1, 0, binding.ORB_QUIRKS_FORK)
    dt = timed(lambda: ctx.extract(frames[0]), 200)
    ctx.enable_stage_timing(True)
    timed(lambda: ctx.extract(frames[0]), 50, warmup=0)
    ms, _ = ctx.stage_timings()
    ctx.enable_stage_timing(False)
    out["orb_extract_640x480"] = {"wall_ms": round(dt * 1e3, 4), "device_stage_ms": {k: round(v, 4) for k, v in ms.items()}}
    k1, d1 = ctx.extract(frames[0])
    k2, d2 = ctx.extract(frames[1])
    prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32))
    dtm = timed(lambda: ctx.match_window(k1, d1, k2, d2, w, h, prev.copy(), 100, 0.9, True), 100)
    out["orb_match_window_1000x1000"] = {"wall_ms": round(dtm * 1e3, 4)}
    ctx.close()
    for batch in (64, 256):
        ctx = binding.Context(1000, 1.2, 8, 20, 7, w, h, batch, 0, binding.ORB_QUIRKS_FORK)
        imgs = np.stack([frames[f % 16] for f in range(batch)])
        dt = timed(lambda: ctx.extract_batch(imgs), 10)
        out["orb_extract_batch_%d" % batch] = {"wall_ms": round(dt * 1e3, 3), "frames_per_s": round(batch / dt, 1)}
        ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
