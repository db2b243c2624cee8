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
from orb_slam2_amd.synth import frame_seed, synth_scene, synth_vocabulary  # noqa: E402


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
    # bag of words: a 10-ary vocabulary of depth 5 (111 110 nodes), transform at 4 levels up, SearchByBoW
    voc = binding.Vocabulary.from_arrays(10, 5, 0, 0, *synth_vocabulary(10, 5, seed=1))
    dtv = timed(lambda: voc.transform(d1, 4), 100)
    _, fv1 = voc.transform(d1, 4)
    _, fv2 = voc.transform(d2, 4)
    valid = np.ones(len(k1), np.uint8)
    dtb = timed(lambda: ctx.match_bow(k1, d1, valid, fv1, k2, d2, fv2, 0.7, True), 100)
    out["orb_voc_bow_1000"] = {"wall_ms": round(dtv * 1e3, 4)}
    out["orb_match_bow_1000x1000"] = {"wall_ms": round(dtb * 1e3, 4)}
    import torch as _t
    big = _t.randint(0, 256, (256 * 1000, 32), dtype=_t.uint8, device="cuda")
    word = _t.zeros(256 * 1000, dtype=_t.int32, device="cuda")
    node = _t.zeros_like(word)
    wt = _t.zeros(256 * 1000, dtype=_t.float64, device="cuda")

    def dev_step():
        voc.transform_features_device(big, 4, word, wt, node)
        _t.cuda.synchronize()

    dtd = timed(dev_step, 20)
    out["orb_voc_transform_device_256000"] = {"wall_ms": round(dtd * 1e3, 4), "descriptors_per_s": round(256000 / dtd, 0)}
    voc.close()
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
