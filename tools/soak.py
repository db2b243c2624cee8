#!/usr/bin/env python3
"""Determinism soak: the same batch extracted over and over must give bit-identical outputs (keypoints,
descriptors, counts) every time — races between the frame groups, the LDS-resident stages or the
asynchronous key-count read-back would show up as differences.  usage: python tools/soak.py [steps] [batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 96
    w, h = 640, 480
    dev = torch.device("cuda", 0)
    frames = np.stack([synth_scene(w, h, frame_seed(1, f % 24)) for f in range(B)])
    d_imgs = torch.from_numpy(frames).to(dev)
    ctx = binding.Context(1000, 1.2, 8, 20, 7, w, h, B, 0, binding.ORB_QUIRKS_FORK)
    cap = ctx.cap
    out = [torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
           torch.zeros(B, dtype=torch.int32, device=dev)]
    ref = None
    bad = 0
    for s in range(steps):
        for t in out:
            t.fill_(0xAB if t.dtype == torch.uint8 else -7)
        ctx.extract_batch_device(d_imgs, out[0], out[1], out[2])
        if s % 50 == 0 or s == steps - 1:
            torch.cuda.synchronize()
            ctx.check_last_batch()
            cnt = out[2].clone()
            mask = (torch.arange(cap, device=dev)[None, :] < cnt[:, None])
            cur = [out[0] * mask[:, :, None], out[1] * mask[:, :, None], cnt]
            if ref is None:
                ref = [c.clone() for c in cur]
            elif not all(torch.equal(a, b) for a, b in zip(cur, ref)):
                bad += 1
                print("step %d: outputs differ from step 0" % s)
    print("soak: %d steps of %d frames, %d comparisons failed" % (steps, B, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
