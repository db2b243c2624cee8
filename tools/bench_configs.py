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
    ctx = binding.Context(2000, 1.2, 8, 20, 7, w, h, 2 * npairs, 0, binding.ORB_QUIRKS_FORK)
    cap = ctx.cap
    d_kps = torch.zeros((2 * npairs, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((2 * npairs, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(2 * npairs, dtype=torch.int32, device=dev)
    d_prev = torch.zeros((npairs, cap, 2), dtype=torch.float32, device=dev)
    d_m12 = torch.zeros((npairs, cap), dtype=torch.int32, device=dev)
    d_nm = torch.zeros(npairs, dtype=torch.int32, device=dev)

    def pair_step():
        ctx.extract_batch_device(d_imgs, d_kps, d_desc, d_cnt)
        # vbPrevMatched = first frame's keypoint positions (src/Tracking.cc:302-304)
        d_prev.copy_(d_kps[:npairs].view(torch.float32).view(npairs, cap, 7)[:, :, :2])
        ctx.match_window_batch_device(d_kps[:npairs], d_desc[:npairs], d_cnt[:npairs], d_kps[npairs:], d_desc[npairs:],
                                      d_cnt[npairs:], w, h, d_prev, d_m12, d_nm, 100, 0.9, True)

    dt = timed(pair_step, 10)
    dt_ext = timed(lambda: ctx.extract_batch_device(d_imgs, d_kps, d_desc, d_cnt), 10)
    ctx.check_last_batch()
    n1 = d_cnt[:npairs].cpu().numpy()
    # comparisons of one pair's SearchForInitialization: candidates of its level-0 queries
    res = {"config": 3, "workload": "64 pairs 752x480: extraction (2000 feats) + SearchForInitialization(100, 0.9)",
           "pairs_per_s": round(npairs / dt, 1), "ms_per_batch": round(dt * 1e3, 3), "extract_ms": round(dt_ext * 1e3, 3),
           "match_ms": round((dt - dt_ext) * 1e3, 3), "mean_matches": float(d_nm.float().mean().item()),
           "mean_keypoints": float(n1.mean())}
    ctx.close()
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
