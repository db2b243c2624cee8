#!/usr/bin/env python3
"""Development aid: duration of every one of the first steps of the headline workload on a fresh context (HIP events
around each step), to see how long the device takes to reach its steady rate.  usage: python tools/warmup_curve.py [steps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
B, W, H = 512, 640, 480
dev = torch.device("cuda", 0)
frames = np.stack([synth_scene(W, H, frame_seed(1, f % 32)) for f in range(B)])
d = torch.from_numpy(frames).to(dev)
batches = [d, torch.flip(d, dims=[2]).contiguous(), torch.flip(d, dims=[1]).contiguous()]
ctx = binding.Context(1000, 1.2, 8, 20, 7, W, H, B, 0, binding.ORB_QUIRKS_FORK)
kps = torch.zeros((B, ctx.cap, 28), dtype=torch.uint8, device=dev)
desc = torch.zeros((B, ctx.cap, 32), dtype=torch.uint8, device=dev)
cnt = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
torch.cuda.synchronize()
ev[0].record(st)
for i in range(n):
    ctx.extract_batch_device(batches[i % 3], kps, desc, cnt, st)
    ev[i + 1].record(st)
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print(" ".join("%.3f" % m for m in ms))
