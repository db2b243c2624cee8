#!/usr/bin/env python3
"""64 pairs of 752x480 frames: extraction + SearchForInitialization, one batch at a time, for rocprofv3 --kernel-trace --stats."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402

w, h, npairs = 752, 480, 64
a = [synth_scene(w, h, frame_seed(3, p)) for p in range(npairs)]
b = [synth_scene(w, h, frame_seed(3, p), shift=(3, 2), noise_seed=frame_seed(3, p) + 1) for p in range(npairs)]
d = torch.from_numpy(np.stack(a + b)).cuda()
ctx = binding.Context(2000, 1.2, 8, 20, 7, w, h, 2 * npairs, 0, binding.ORB_QUIRKS_FORK)
cap = ctx.cap
kps = torch.zeros((2 * npairs, cap, 28), dtype=torch.uint8, device="cuda")
desc = torch.zeros((2 * npairs, cap, 32), dtype=torch.uint8, device="cuda")
cnt = torch.zeros(2 * npairs, dtype=torch.int32, device="cuda")
prev = torch.zeros((npairs, cap, 2), dtype=torch.float32, device="cuda")
m12 = torch.zeros((npairs, cap), dtype=torch.int32, device="cuda")
nm = torch.zeros(npairs, dtype=torch.int32, device="cuda")
for _ in range(12):
    ctx.extract_batch_device(d, kps, desc, cnt)
    prev.copy_(kps[:npairs].view(torch.float32).view(npairs, cap, 7)[:, :, :2])
    ctx.match_window_batch_device(kps[:npairs], desc[:npairs], cnt[:npairs], kps[npairs:], desc[npairs:], cnt[npairs:], w, h, prev,
                                  m12, nm, 100, 0.9, True)
    torch.cuda.synchronize()
ctx.check_last_batch()
print("matches", float(nm.float().mean().item()))
