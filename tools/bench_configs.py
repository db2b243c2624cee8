#!/usr/bin/env python3
"""Throughput of the non-headline configurations on one MI355X (BASELINE.json configs 2-4):
2) 64 x 1241x376 frames, 2000 features; 3) 64 pairs 752x480: extraction (2000-feature initial
extractor) + SearchForInitialization(window 100, ratio 0.9); 4) 64 x 3840x2160 frames, 4000 features.
Prints one JSON object per configuration.  Inputs are resident in HBM before timing starts."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402

dev = torch.device("cuda", 0)


def timed(fn, steps, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def extraction(cfg, w, h, nf, batch, steps, distinct):
    base = [synth_scene(w, h, frame_seed(cfg, f)) for f in range(distinct)]
    frames = np.stack([base[f % distinct] for f in range(batch)])
    d_imgs = torch.from_numpy(frames).to(dev)
    ctx = binding.Context(nf, 1.2, 8, 20, 7, w, h, batch, 0, binding.ORB_QUIRKS_FORK)
    cap = ctx.cap
    d_kps = torch.zeros((batch, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((batch, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(batch, dtype=torch.int32, device=dev)
    dt = timed(lambda: ctx.extract_batch_device(d_imgs, d_kps, d_desc, d_cnt), steps)
    ctx.enable_stage_timing(True)  # stage times from a second pass: stages back to back on one stream
    timed(lambda: ctx.extract_batch_device(d_imgs, d_kps, d_desc, d_cnt), steps, warmup=0)
    ms, _ = ctx.stage_timings()
    ctx.enable_stage_timing(False)
    ctx.check_last_batch()
    return ctx, dt, ms, (d_kps, d_desc, d_cnt), frames


def config2():
    ctx, dt, ms, _, _ = extraction(2, 1241, 376, 2000, 64, 10, 16)
    ctx.close()
    return {"config": 2, "workload": "64 x 1241x376, nFeatures 2000", "frames_per_s": round(64 / dt, 1),
            "ms_per_batch": round(dt * 1e3, 3), "stage_ms": {k: round(v, 4) for k, v in ms.items()}}


def config3():
    # pairs = (frame, shifted frame); both halves extracted in one batch of 128
    w, h, npairs = 752, 480, 64
    a = [synth_scene(w, h, frame_seed(3, p)) for p in range(8)]
    b = [synth_scene(w, h, frame_seed(3, p), shift=(3, 2), noise_seed=frame_seed(3, p) + 1) for p in range(8)]
    frames = np.stack([a[p % 8] for p in range(npairs)] + [b[p % 8] for p in range(npairs)])
    d_imgs = torch.from_numpy(frames).to(dev)

    class Lane:  # one batch of pairs in flight: a context, its buffers, a stream
        def __init__(self):
            self.ctx = binding.Context(2000, 1.2, 8, 20, 7, w, h, 2 * npairs, 0, binding.ORB_QUIRKS_FORK)
            cap = self.ctx.cap
            self.cap = cap
            self.kps = torch.zeros((2 * npairs, cap, 28), dtype=torch.uint8, device=dev)
            self.desc = torch.zeros((2 * npairs, cap, 32), dtype=torch.uint8, device=dev)
            self.cnt = torch.zeros(2 * npairs, dtype=torch.int32, device=dev)
            self.prev = torch.zeros((npairs, cap, 2), dtype=torch.float32, device=dev)
            self.m12 = torch.zeros((npairs, cap), dtype=torch.int32, device=dev)
            self.nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
            self.stream = torch.cuda.Stream()

        def extract(self, stream):
            self.ctx.extract_batch_device(d_imgs, self.kps, self.desc, self.cnt, stream)

        def pair_step(self, stream):
            self.extract(stream)
            with torch.cuda.stream(stream):
                # vbPrevMatched = first frame's keypoint positions (src/Tracking.cc:302-304)
                self.prev.copy_(self.kps[:npairs].view(torch.float32).view(npairs, self.cap, 7)[:, :, :2])
            self.ctx.match_window_batch_device(self.kps[:npairs], self.desc[:npairs], self.cnt[:npairs], self.kps[npairs:],
                                               self.desc[npairs:], self.cnt[npairs:], w, h, self.prev, self.m12, self.nm, 100, 0.9,
                                               True, stream)

    lanes = [Lane(), Lane()]
    cur = torch.cuda.current_stream()
    # one batch of pairs at a time
    dt = timed(lambda: lanes[0].pair_step(cur), 10)
    dt_ext = timed(lambda: lanes[0].extract(cur), 10)
    lanes[0].ctx.check_last_batch()
    # two batches in flight, each on a stream and a context of its own: the matcher's last kernel resolves a pair's
    # queries one after the other on one workgroup per pair (64 of 256 CUs busy), which the other batch's extraction hides
    step = [0]

    def piped():
        ln = lanes[step[0] & 1]
        ln.pair_step(ln.stream)
        step[0] += 1

    dt2 = timed(piped, 20)
    for ln in lanes:
        ln.ctx.check_last_batch()
    n1 = lanes[0].cnt[:npairs].cpu().numpy()
    same = bool(torch.equal(lanes[0].m12, lanes[1].m12) and torch.equal(lanes[0].nm, lanes[1].nm))
    res = {"config": 3, "workload": "64 pairs 752x480: extraction (2000 feats) + SearchForInitialization(100, 0.9)",
           "pairs_per_s": round(npairs / dt2, 1), "ms_per_batch": round(dt2 * 1e3, 3),
           "in_flight": "two batches of 64 pairs, each on its own stream and context (results of both equal: %s)" % same,
           "one_batch_at_a_time": {"pairs_per_s": round(npairs / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
                                   "extract_ms": round(dt_ext * 1e3, 3), "match_ms": round((dt - dt_ext) * 1e3, 3)},
           "mean_matches": float(lanes[0].nm.float().mean().item()), "mean_keypoints": float(n1.mean())}
    for ln in lanes:
        ln.ctx.close()
    return res


def config4():
    ctx, dt, ms, _, _ = extraction(4, 3840, 2160, 4000, 64, 5, 2)
    ctx.close()
    return {"config": 4, "workload": "64 x 3840x2160 per step, nFeatures 4000 (1024-frame job = 16 such steps per GPU)",
            "frames_per_s": round(64 / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "stage_ms": {k: round(v, 4) for k, v in ms.items()}}


def main():
    for o in (config2(), config3(), config4()):
        print(json.dumps(o))


if __name__ == "__main__":
    main()
