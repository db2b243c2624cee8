#!/usr/bin/env python3
"""Throughput of the non-headline configurations (BASELINE.json configs 2-4), on one GPU or sharded over the
ranks of a process group (one process per GPU):

2) 1241x376 frames, 2000 features, 64 per step and GPU (weak scaling: every rank extracts batches of its own);
3) 64 pairs 752x480 per step: extraction (2000-feature initial extractor) + SearchForInitialization(window 100,
   ratio 0.9) (one GPU; pairs are independent, more GPUs are replicas);
4) the 1024-frame 3840x2160 job, 4000 features: contiguous frame blocks [r 1024 / R, (r + 1) 1024 / R) per rank,
   64 frames per step, no exchange on the data path (strong scaling: the job is fixed); every frame's outputs go
   into a checksum that does not depend on R.

Inputs are resident in HBM before timing starts.  Consecutive steps read different frames: variants of the drawn
batch (rolled sideways, every other one mirrored), more of them than the 256 MB Infinity Cache holds.
Prints one JSON object per configuration."""
import json
import os
import sys
import time
import zlib

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.sharded import shard_range  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402

MALL_BYTES = 256 << 20


class Ranks:
    """How a configuration sees the process group; the default is a single process on cuda:0."""

    def __init__(self, rank=0, world=1, device=0, barrier=None, max_over_ranks=None, sum_over_ranks=None):
        self.rank, self.world, self.device = rank, world, device
        self.barrier = barrier or (lambda: None)
        self.max_over_ranks = max_over_ranks or (lambda x: x)
        self.sum_over_ranks = sum_over_ranks or (lambda x: x)

    @property
    def dev(self):
        return torch.device("cuda", self.device)


def variants(d_base, min_bytes=MALL_BYTES + (32 << 20)):
    """Distinct batches made from one on the device: variant k is the batch rolled 11 k columns sideways, mirrored
    left-right when k is odd (other corners, other keypoints); as many as it takes to exceed `min_bytes`."""
    n = max(3, -(-min_bytes // d_base.numel()))
    out = []
    for k in range(n):
        v = torch.roll(d_base, shifts=11 * k, dims=2) if k else d_base
        out.append(torch.flip(v, dims=[2]).contiguous() if k & 1 else v.contiguous())
    return out


PREHEAT_S = 0.1


def timed(fn, steps, R, warmup=2, preheat=None):
    """Mean wall time of `steps` calls of fn, maximum over the ranks; barrier + device synchronisation on both sides.
    In front of the warm-up calls the device is kept busy for PREHEAT_S (with `preheat`, or fn): the engine clock takes
    about 35 ms of load to reach its steady value."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < PREHEAT_S:
        (preheat or fn)()
        torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    R.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    R.barrier()
    return R.max_over_ranks(time.perf_counter() - t0) / steps


def config2(R=None):
    R = R or Ranks()
    w, h, nf, batch, steps = 1241, 376, 2000, 64, 20
    frames = np.stack([synth_scene(w, h, frame_seed(2, R.rank * batch + f)) for f in range(batch)])
    d_var = variants(torch.from_numpy(frames).to(R.dev))
    ctx = binding.Context(nf, 1.2, 8, 20, 7, w, h, batch, R.device, binding.ORB_QUIRKS_FORK)
    cap = ctx.cap
    d_kps = torch.zeros((batch, cap, 28), dtype=torch.uint8, device=R.dev)
    d_desc = torch.zeros((batch, cap, 32), dtype=torch.uint8, device=R.dev)
    d_cnt = torch.zeros(batch, dtype=torch.int32, device=R.dev)
    k = [0]

    def step():
        ctx.extract_batch_device(d_var[k[0] % len(d_var)], d_kps, d_desc, d_cnt)
        k[0] += 1

    dt = timed(step, steps, R)
    ctx.enable_stage_timing(True)  # stage times from a second pass: stages back to back on one stream
    timed(step, steps, R, warmup=0)
    ms, _ = ctx.stage_timings()
    ctx.enable_stage_timing(False)
    ctx.check_last_batch()
    ctx.close()
    return {"config": 2, "workload": "%d x 1241x376 per GPU and step, nFeatures 2000, %d distinct batches in turn (%d MB)"
                                     % (batch, len(d_var), len(d_var) * d_var[0].numel() >> 20),
            "n_gpus": R.world, "scaling": "weak", "frames_per_s": round(R.world * batch / dt, 1),
            "ms_per_batch": round(dt * 1e3, 3), "stage_ms_rank0": {k: round(v, 4) for k, v in ms.items()}}


def config3(R=None):
    """One GPU (rank 0's); frame pairs are independent, so more GPUs are replicas of this."""
    R = R or Ranks()
    w, h, npairs = 752, 480, 64
    a = [synth_scene(w, h, frame_seed(3, p)) for p in range(npairs)]
    b = [synth_scene(w, h, frame_seed(3, p), shift=(3, 2), noise_seed=frame_seed(3, p) + 1) for p in range(npairs)]
    # first and second frames of the pairs in one batch of 128; a variant rolls and mirrors both frames of a pair alike
    d_var = variants(torch.from_numpy(np.stack(a + b)).to(R.dev))

    class Lane:  # one batch of pairs in flight: a context, its buffers, a stream
        def __init__(self):
            self.ctx = binding.Context(2000, 1.2, 8, 20, 7, w, h, 2 * npairs, R.device, binding.ORB_QUIRKS_FORK)
            cap = self.ctx.cap
            self.cap = cap
            self.kps = torch.zeros((2 * npairs, cap, 28), dtype=torch.uint8, device=R.dev)
            self.desc = torch.zeros((2 * npairs, cap, 32), dtype=torch.uint8, device=R.dev)
            self.cnt = torch.zeros(2 * npairs, dtype=torch.int32, device=R.dev)
            self.prev = torch.zeros((npairs, cap, 2), dtype=torch.float32, device=R.dev)
            self.m12 = torch.zeros((npairs, cap), dtype=torch.int32, device=R.dev)
            self.nm = torch.zeros(npairs, dtype=torch.int32, device=R.dev)
            self.stream = torch.cuda.Stream()
            self.k = 0

        def extract(self, stream):
            self.ctx.extract_batch_device(d_var[self.k % len(d_var)], self.kps, self.desc, self.cnt, stream)
            self.k += 1

        def pair_step(self, stream):
            self.extract(stream)
            with torch.cuda.stream(stream):
                # vbPrevMatched = first frame's keypoint positions (src/Tracking.cc:302-304)
                self.prev.copy_(self.kps[:npairs].view(torch.float32).view(npairs, self.cap, 7)[:, :, :2])
            self.ctx.match_window_batch_device(self.kps[:npairs], self.desc[:npairs], self.cnt[:npairs], self.kps[npairs:],
                                               self.desc[npairs:], self.cnt[npairs:], w, h, self.prev, self.m12, self.nm, 100, 0.9,
                                               True, stream)

    one = Ranks(device=R.device)  # timed on this GPU alone
    lanes = [Lane(), Lane()]
    cur = torch.cuda.current_stream()
    # one batch of pairs at a time
    dt = timed(lambda: lanes[0].pair_step(cur), 12, one)
    dt_ext = timed(lambda: lanes[0].extract(cur), 12, one)
    lanes[0].ctx.check_last_batch()
    # two batches in flight, each on a stream and a context of its own (the other batch's extraction runs beside
    # the matcher's ordered stage)
    step = [0]

    def piped():
        ln = lanes[step[0] & 1]
        ln.pair_step(ln.stream)
        step[0] += 1

    dt2 = timed(piped, 24, one)
    for ln in lanes:
        ln.ctx.check_last_batch()
    # both lanes on the same variant: equal results
    for ln in lanes:
        ln.k = 0
        ln.pair_step(cur)
    torch.cuda.synchronize()
    n1 = lanes[0].cnt[:npairs].cpu().numpy()
    same = bool(torch.equal(lanes[0].m12, lanes[1].m12) and torch.equal(lanes[0].nm, lanes[1].nm))
    res = {"config": 3, "workload": "64 pairs 752x480 per step: extraction (2000 feats) + SearchForInitialization(100, 0.9), %d distinct batches in turn (%d MB)"
                                    % (len(d_var), len(d_var) * d_var[0].numel() >> 20),
           "n_gpus": 1, "pairs_per_s": round(npairs / dt2, 1), "ms_per_batch": round(dt2 * 1e3, 3),
           "in_flight": "two batches of 64 pairs, each on its own stream and context (results of both equal: %s)" % same,
           "one_batch_at_a_time": {"pairs_per_s": round(npairs / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
                                   "extract_ms": round(dt_ext * 1e3, 3), "match_ms": round((dt - dt_ext) * 1e3, 3)},
           "mean_matches": float(lanes[0].nm.float().mean().item()), "mean_keypoints": float(n1.mean())}
    for ln in lanes:
        ln.ctx.close()
    return res


def job_frames(lo, hi, w, h, cfg, scenes, dev):
    """Frames [lo, hi) of a job whose frame f is drawn scene f mod `scenes`, rolled 5 (f div scenes) columns
    sideways: every frame of the job is distinct and a frame does not depend on how the job is sharded."""
    base = [torch.from_numpy(synth_scene(w, h, frame_seed(cfg, s))).to(dev) for s in range(scenes)]
    out = torch.empty((hi - lo, h, w), dtype=torch.uint8, device=dev)
    for f in range(lo, hi):
        out[f - lo] = torch.roll(base[f % scenes], shifts=5 * (f // scenes), dims=1)
    return out


def frame_checksums(d_kps, d_desc, d_cnt):
    """CRC-32 of every frame's outputs (count, keypoint records, descriptors)."""
    cnt = d_cnt.cpu().numpy()
    kps = d_kps.cpu().numpy()
    desc = d_desc.cpu().numpy()
    return [zlib.crc32(desc[f, : cnt[f]].tobytes(), zlib.crc32(kps[f, : cnt[f]].tobytes(), int(cnt[f]))) for f in range(len(cnt))]


def config4(R=None, job=1024, step_frames=64, w=3840, h=2160, nf=4000, scenes=8):
    R = R or Ranks()
    lo, hi = shard_range(job, R.rank, R.world)
    n = hi - lo
    d_job = job_frames(lo, hi, w, h, 4, scenes, R.dev)
    ctx = binding.Context(nf, 1.2, 8, 20, 7, w, h, step_frames, R.device, binding.ORB_QUIRKS_FORK)
    cap = ctx.cap
    d_kps = torch.zeros((max(n, 1), cap, 28), dtype=torch.uint8, device=R.dev)
    d_desc = torch.zeros((max(n, 1), cap, 32), dtype=torch.uint8, device=R.dev)
    d_cnt = torch.zeros(max(n, 1), dtype=torch.int32, device=R.dev)

    def run_job():
        for s in range(0, n, step_frames):
            e = min(n, s + step_frames)
            ctx.extract_batch_device(d_job[s:e], d_kps[s:e], d_desc[s:e], d_cnt[s:e])

    def one_step():
        if n:
            e = min(n, step_frames)
            ctx.extract_batch_device(d_job[:e], d_kps[:e], d_desc[:e], d_cnt[:e])

    one_step()
    dt = timed(run_job, 1, R, warmup=0, preheat=one_step)
    ctx.check_last_batch()
    ms = {}
    if n:
        ctx.enable_stage_timing(True)
        one_step()
        torch.cuda.synchronize()
        ms, _ = ctx.stage_timings()
        ctx.enable_stage_timing(False)
    # one number for the whole job that does not depend on the sharding: the sum of the per-frame CRCs
    crc = frame_checksums(d_kps[:n], d_desc[:n], d_cnt[:n]) if n else []
    total = int(R.sum_over_ranks(float(sum(crc)))) if R.world > 1 else sum(crc)
    nkp = R.sum_over_ranks(float(d_cnt[:n].sum().item())) / job
    ctx.close()
    return {"config": 4, "workload": "%d x %dx%d job, nFeatures %d, contiguous frame blocks per GPU, %d frames per step, every frame distinct (%d drawn scenes, rolled)"
                                     % (job, w, h, nf, step_frames, scenes),
            "n_gpus": R.world, "scaling": "strong", "frames_per_s": round(job / dt, 1), "job_ms": round(dt * 1e3, 2),
            "frames_per_rank": n, "ms_per_step_rank0": round(dt * 1e3 / max(1, -(-n // step_frames)), 3),
            "stage_ms_rank0": {k: round(v, 4) for k, v in ms.items()}, "mean_keypoints": round(nkp, 1),
            "sum_of_frame_crc32": total}


def main():
    for o in (config2(), config3(), config4()):
        print(json.dumps(o), flush=True)


if __name__ == "__main__":
    main()
