#!/usr/bin/env python3
"""Benchmark of the ORB front end on MI355X.

One process per GPU (launched by torch.distributed.run for --gpus > 1).  A
"step" is one pass of ORBextractor::operator() over a batch of synthetic
640x480 frames that already sit in HBM (1000 features, 8 levels, scale 1.2 —
the configuration the headline metric is quoted on); consecutive steps take
different batches (three of them, 470 MB together, more than the 256 MB
Infinity Cache holds), so no step re-reads what the previous one left in
cache.  Frames are independent,
so ranks share nothing on the data path (weak scaling: every rank extracts its
own batch).  The same JSON line also carries the brute-force Hamming workload
(100k queries x 10M database rows, rows sharded over the ranks, packed top-2
all-gathered over RCCL and merged), the roofline of the dominant extraction
kernel from HIP-event stage timings, the CPU oracle timed on a bounded
sample of the same frames, and - so that one line tells the whole story, not
only the most favourable operating point - the same extraction at other batch
sizes, through the host-pointer entry points (PCIe-inclusive), one frame at a
time, and the other benchmark configurations (KITTI-shaped, EuRoC-shaped
pairs with matching, 4K), all timed in this run.  None of those enter `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, spec
# Ceiling of the Hamming loop (8 x v_xor_b32 + 8 x accumulating v_bcnt_u32_b32 per comparison), measured with
# tools/valu_peak.hip on this chip (profiles/r02_valu_peak_8waves_per_simd.txt): v_bcnt_u32_b32 is a single-rate
# instruction (16 lanes per clock and SIMD: 38 T lane-ops/s over 1024 SIMDs at 2.4 GHz), v_xor_b32 with two
# VGPR sources a double-rate one (28-29 lanes per clock: 70 T), so a pair costs 6 cycles per wavefront at
# best = 21.3 lanes per clock = 52.4 T lane-ops/s; the best order of the bare pair the micro-benchmark found
# sustains 49.4 T (20.1 lanes per clock).  That measured figure is the peak the scan kernel is priced against.
VALU_LANE_OPS = 49.4e12
# The scan makes the distances on the matrix cores (one comparison = 256 multiply-adds on FP4 operands holding the
# values 0 and +-1); its ceiling is the dense FP4 MFMA rate: v_mfma_scale_f32_32x32x64_f8f6f4 = 65536 multiply-adds
# per 32 cycles and SIMD, 1024 SIMDs, 2.4 GHz (about 10 PFLOP/s).
MFMA_FP4_OPS = 1024 * 2.4e9 * 4096
VALU_LANE_OPS_SINGLE_RATE = 256 * 4 * 16 * 2.4e9   # what both instructions would give at 16 lanes per clock
PREHEAT_S = 0.1   # untimed load in front of every timed section, so that it runs at the steady engine clock


def level_geometry(w, h, nlevels, scale):
    """Level sizes as src/ORBextractor.cc:917-918 computes them (float32 arithmetic)."""
    s = np.float32(1.0)
    out = []
    for l in range(nlevels):
        if l:
            s = np.float32(np.float64(s) * np.float64(np.float32(scale)))
        inv = np.float32(1.0) / s
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=512, help="frames per step and GPU")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--preset", default="fork", choices=["fork", "canonical"])
    ap.add_argument("--bf-queries", type=int, default=100000)
    ap.add_argument("--bf-rows", type=int, default=10000000, help="database rows over ALL ranks")
    ap.add_argument("--bf-steps", type=int, default=3)
    ap.add_argument("--no-hamming", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=300, help="frames the single-thread CPU baseline extracts (about 10 s)")
    ap.add_argument("--no-extras", action="store_true", help="skip batch sweep, host path, single frame and other configurations")
    ap.add_argument("--job-frames", type=int, default=1024, help="frames of the 3840x2160 job (configuration 4), over ALL ranks")
    args = ap.parse_args()

    # The one JSON line is all this program writes to its standard output: whatever libraries print there meanwhile
    # (RCCL announces its version on the first collective) goes to standard error instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from orb_slam2_amd import binding
    from orb_slam2_amd.synth import frame_seed, splitmix64, synth_scene

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("WORLD_SIZE=%d but --gpus %d: launch with torch.distributed.run --nproc-per-node N" % (world, args.gpus), file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    # ORB_BENCH_REHEARSE=1: several ranks share the visible GPUs and talk over gloo (a dry run of the
    # multi-rank code path on a box with fewer GPUs than ranks; its numbers mean nothing)
    rehearse = os.environ.get("ORB_BENCH_REHEARSE") == "1"
    if rehearse:
        local %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # ORB_BENCH_FORCE_GROUP=1: a process group (and every collective of the multi-rank path) even for one rank, so
    # that the RCCL calls can be exercised on a one-GPU box
    grouped = world > 1 or os.environ.get("ORB_BENCH_FORCE_GROUP") == "1"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def barrier():
        if grouped:
            dist.barrier()

    def max_over_ranks(x):
        if not grouped:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if not grouped:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    # a host-side group next to the RCCL one: ranks that only wait (while rank 0 runs a program of its own over
    # all GPUs) must not wait inside a collective kernel that spins on their GPU
    host_group = dist.new_group(backend="gloo") if grouped and not rehearse else None

    def host_barrier():
        if grouped:
            dist.barrier(group=host_group) if host_group is not None else dist.barrier()

    def note(what):
        if rehearse:
            print("[rank %d] %s" % (rank, what), file=sys.stderr, flush=True)

    quirks = binding.ORB_QUIRKS_FORK if args.preset == "fork" else binding.ORB_QUIRKS_CANONICAL
    W, H, B = args.width, args.height, args.batch

    # ---- extraction ---------------------------------------------------------------------
    frames = np.stack([synth_scene(W, H, frame_seed(1, rank * B + f)) for f in range(B)])
    d_base = torch.from_numpy(frames).to(dev)
    # three distinct batches: the drawn frames, mirrored left-right, mirrored top-bottom (other corners, other
    # keypoints; made on the device so that the run does not spend a minute drawing scenes)
    d_batches = [d_base, torch.flip(d_base, dims=[2]).contiguous(), torch.flip(d_base, dims=[1]).contiguous()]
    ctx = binding.Context(args.nfeatures, 1.2, 8, 20, 7, W, H, B, local, quirks)
    cap = ctx.cap
    d_kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device=dev)
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev)
    d_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    nstep = [0]

    def step():
        ctx.extract_batch_device(d_batches[nstep[0] % 3], d_kps, d_desc, d_cnt, stream)
        nstep[0] += 1

    def preheat(fn, seconds=PREHEAT_S):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            fn()
            fn()
            torch.cuda.synchronize()

    note("context created")
    # The engine clock needs about 35 ms of load to reach its steady value (tools/warmup_curve.py: a step takes 1.5 ms
    # right after an idle spell and 1.27 ms from the 25th step on), more than a handful of warm-up steps last: the
    # same steps are run untimed for PREHEAT_S first, then the W warm-up steps, then the K timed ones.
    preheat(step)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.check_last_batch()
    note("warm-up done")
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    ctx.check_last_batch()
    note("extraction timed")
    # Per-kernel durations for the roofline: the same steps again with HIP events between the stages.
    # In that mode the library enqueues a batch as one group on one stream, so that each event pair
    # brackets exactly one stage; the timed loop above lets it overlap groups of frames on several streams.
    ctx.enable_stage_timing(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    stage_ms, nrec = ctx.stage_timings()
    qt_second = ctx.quadtree_second_pass_levels()  # of the last batch: (frame, level) pairs that needed the finer histogram
    ctx.enable_stage_timing(False)
    ctx.check_last_batch()
    nstep[0] = 0
    step()  # leave the results of the drawn frames in the output buffers for the parity check below
    torch.cuda.synchronize()
    counts = d_cnt.cpu().numpy()
    fps = world * B * args.steps / dt

    # algorithmic bytes per frame and stage (u8 pixels; P_l = pixels of level l, n = keypoints)
    lv = level_geometry(W, H, 8, 1.2)
    P = [w * h for (w, h) in lv]
    nkp = float(counts.mean())
    stage_bytes = {
        "pyramid": sum(P[l - 1] + P[l] for l in range(1, 8)),      # read level l-1, write level l
        "fast": sum(P),                                             # every level read once
        "quadtree": 0,                                              # bookkeeping on keys, no pixel traffic
        "describe": nkp * (43 * 43 + 28 + 32),                      # source window + keypoint + descriptor
    }
    dom = max(stage_ms, key=lambda k: stage_ms[k])
    kernel_name = {"pyramid": "k_resize_tiled x7", "fast": "k_fast_cells", "quadtree": "k_quadtree", "describe": "k_describe"}[dom]
    achieved = stage_bytes[dom] * B / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
    # HBM-side bytes and VALU instruction counts of the same kernel from the committed rocprofv3 --pmc passes
    # (tools/pmc_collect.sh; counters are per launch at the batch size recorded in the file, scaled to this batch).
    traffic, valu, pmc_source = None, None, None
    import glob

    tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_per_kernel.json")))
    pmc_path = tables[-1] if tables else ""
    if os.path.exists(pmc_path) and dom != "pyramid" and args.width == 640 and args.height == 480 and args.nfeatures == 1000:
        with open(pmc_path) as fh:
            table = json.load(fh)
        pk = next((v for k, v in table.items() if kernel_name in k), None)  # (template instances carry their arguments)
        if pk:
            k = B / pk["batch"]
            cn = pk["counters_per_launch"]
            pmc_source = os.path.relpath(pmc_path, ROOT)
            if "FETCH_SIZE" in cn and "WRITE_SIZE" in cn:
                # counter unit KB; gfx950 tallies a 128-byte read request as 64 bytes (tools/fetch_calib.hip
                # reads 512 MiB with 4- and 16-byte loads per lane: FETCH_SIZE reports 256 MiB, WRITE_SIZE is exact)
                traffic = int((2 * cn["FETCH_SIZE"] + cn["WRITE_SIZE"]) * 1024 * k)
            if "SQ_INSTS_VALU" in cn:
                wi = cn["SQ_INSTS_VALU"] * k
                # a single-rate wave-instruction holds a SIMD's ALU for 4 cycles (double-rate ones, such as a
                # two-register v_add / v_xor / v_fma_f32, for 2: "util" prices every instruction at 4 and is an
                # upper bound of the issue time); 1024 SIMDs at 2.4 GHz
                valu = {"wave_instructions_per_launch": int(wi), "issue_cycles_per_instruction": 4,
                        "util": round(wi * 4 / (1024 * 2.4e9 * stage_ms[dom] * 1e-3), 4)}
    roofline = {
        "bound": "hbm", "kernel": kernel_name, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "valu_issue": valu,
        "traffic_source": ("%s (rocprofv3 --pmc passes of an earlier run of this command, scaled to this batch; not "
                           "collected in this run)" % pmc_source) if pmc_source else None,
        "algorithmic_bytes_per_launch": int(stage_bytes[dom] * B), "launch_ms": round(stage_ms[dom], 4),
        "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()}, "stage_batches": nrec,
        "stage_timing": "HIP events between the stages, same steps repeated after the timed loop with group overlap off",
        "quadtree_second_pass_levels": qt_second, "quadtree_levels_per_batch": B * 8,
    }

    # ---- brute-force Hamming ------------------------------------------------------------
    hamming = None
    if not args.no_hamming:
        nq, nd = args.bf_queries, args.bf_rows
        lo, hi = rank * nd // world, (rank + 1) * nd // world
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        d_db = torch.randint(0, 256, (hi - lo, 32), dtype=torch.uint8, device=dev, generator=g)
        gq = torch.Generator(device=dev)
        gq.manual_seed(99)
        d_q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev, generator=gq)
        # the class the multi-rank tests exercise (orb_slam2_amd/sharded.py): scan of this rank's rows, one
        # all-gather of the packed words, merge on every rank; the scan is bracketed by events for the roofline
        from orb_slam2_amd.sharded import ShardedBruteForce, hip_backend

        scan, merge, bf = hip_backend(d_db, lo, local)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kern_ms = []
        last = {}

        def timed_scan(q):
            ev0.record(stream)
            out = scan(q)
            ev1.record(stream)
            last["packed"] = out
            return out

        sbf = ShardedBruteForce(timed_scan, merge, always_gather=grouped)
        note("database ready")
        sbf.query(d_q)
        torch.cuda.synchronize()
        note("first brute-force step done")
        bf.clock_probe(True)  # (clears the sums: the timed scans below are what the probe sees)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.bf_steps):
            d_idx, d_b, d_s = sbf.query(d_q)
            torch.cuda.synchronize()
            kern_ms.append(ev0.elapsed_time(ev1))
        barrier()
        dth = max_over_ranks(time.perf_counter() - t0)
        clock_mhz = bf.clock_probe(False)  # shader-clock ticks per tick of the 100 MHz counter, over the timed scans' workgroups
        gcmp = nq * nd * args.bf_steps / dth / 1e9
        kms = float(np.mean(kern_ms))
        kern_gcmp = nq * (hi - lo) / (kms * 1e-3) / 1e9
        lane_ops = kern_gcmp * 1e9 * 16  # 8 v_xor_b32 + 8 v_bcnt_u32_b32 per comparison
        hamming = {
            "metric": "Hamming matches Gcmp/s (brute-force top-2)", "value": round(gcmp, 2), "unit": "Gcmp/s",
            "queries": nq, "db_rows": nd, "rows_per_rank": hi - lo, "steps": args.bf_steps, "scaling": "strong",
            "ms_per_step": round(dth / args.bf_steps * 1e3, 3),
            "roofline": {"bound": "mfma", "kernel": "k_bf_mfma", "achieved": round(kern_gcmp * 1e9 * 512 / 1e12, 1),
                         "peak": round(MFMA_FP4_OPS / 1e12, 1), "unit": "TOP/s",
                         "frac": round(kern_gcmp * 1e9 * 512 / MFMA_FP4_OPS, 4),
                         "peak_source": "dense FP4 MFMA: 4096 operations per cycle and SIMD x 1024 SIMDs x 2.4 GHz; a comparison is 256 multiply-adds",
                         "int8_form_gcmp_per_s": 6204,  # measured when the scan used v_mfma_i32_32x32x32_i8
                         "vector_alu_form": {"what": "the same scan as 8 x (v_xor_b32 + v_bcnt_u32_b32) per comparison, the library's kernel until this one (tools/bf_valu_scan.inc)",
                                             "gcmp_per_s": 2734, "ceiling_gcmp_per_s": round(VALU_LANE_OPS / 16 / 1e9),
                                             "this_kernel_over_its_ceiling": round(lane_ops / VALU_LANE_OPS, 3)},
                         "launch_ms": round(kms, 3),
                         # the same against what the matrix cores could do at the clock the launch actually ran at
                         # (the board lowers it under this kernel's power draw): 4096 operations per cycle and SIMD
                         "clock_mhz": round(clock_mhz, 1),
                         "frac_at_clock": round(kern_gcmp * 1e9 * 512 / (1024 * clock_mhz * 1e6 * 4096), 4) if clock_mhz > 0 else None,
                         "db_stream_GBs": round((hi - lo) * 32 * ((nq + 511) // 512) / (kms * 1e-3) / 1e9, 1)},
        }
        if rank == 0 and not args.no_cpu:
            import threading

            import oracle

            # CPU baseline: the scalar port on one thread over a bounded sample of the same data
            sq, sd = 64, min(500000, hi - lo)
            qh = d_q[:sq].cpu().numpy()
            dbh = d_db[:sd].cpu().numpy()
            t0 = time.perf_counter()
            oracle.bf_top2(qh, dbh)
            tc = time.perf_counter() - t0
            hamming["cpu_baseline"] = {"value": round(sq * sd / tc / 1e9, 3), "unit": "Gcmp/s", "cores": 1, "kind": "port",
                                       "sample": "%d queries x %d rows, 4x popcount64 per pair" % (sq, sd)}
            # parity of the TIMED result: a slice of its queries against the oracle's scan of the same rows
            # (this rank's shard for the packed words; at one GPU that is the whole database and the final answer)
            sel = np.arange(0, nq, max(1, nq // 192))[:192]
            qs = d_q[torch.from_numpy(sel).to(dev)].cpu().numpy()
            dball = d_db.cpu().numpy()
            nthr = max(1, min(16, os.cpu_count() or 1))
            parts = np.array_split(np.arange(len(sel)), nthr)
            res = [None] * nthr

            def work(k):
                res[k] = oracle.bf_top2(qs[parts[k]], dball)

            th = [threading.Thread(target=work, args=(k,)) for k in range(nthr)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            tall = time.perf_counter() - t0
            io, bo, so = (np.concatenate([r[k] for r in res]) for k in range(3))
            pk = last["packed"].cpu().numpy().view(np.uint64)[sel]
            ok = ((pk & np.uint64(0xFFFFFFFF)).astype(np.int64) == io.astype(np.int64) + lo).all() and \
                 ((pk >> np.uint64(48)).astype(np.uint16) == bo).all() and \
                 (((pk >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.uint16) == so).all()
            if world == 1:
                ok = ok and (d_idx.cpu().numpy()[sel] == io).all() and (d_b.cpu().numpy().view(np.uint16)[sel] == bo).all() \
                     and (d_s.cpu().numpy().view(np.uint16)[sel] == so).all()
            hamming["parity_checked"] = bool(ok)
            hamming["parity_sample"] = "%d queries of the timed scan x all %d rows of rank 0 against the oracle" % (len(sel), hi - lo)
            hamming["cpu_baseline"]["all_cores"] = {"value": round(len(sel) * (hi - lo) / tall / 1e9, 3), "unit": "Gcmp/s",
                                                    "cores": nthr, "sample": "the parity sample, one slice per thread"}

    # ---- CPU baseline + parity spot check (rank 0, single GPU runs only) ----------------
    cpu = None
    parity = None
    if rank == 0:
        import oracle

        e = oracle.Extractor(args.nfeatures, 1.2, 8, 20, 7, oracle.QUIRKS_FORK if args.preset == "fork" else oracle.QUIRKS_CANONICAL)
        ko, do = e.extract(frames[0])
        kg = d_kps[0, : counts[0]].cpu().numpy().view(binding.KP_DTYPE).reshape(-1)
        dg = d_desc[0, : counts[0]].cpu().numpy()
        parity = bool(len(kg) == len(ko) and (kg == ko).all() and (dg == do).all())
        if not args.no_cpu and world == 1:
            nc = min(args.cpu_frames, B)
            t0 = time.perf_counter()
            reps = 0
            while reps < args.cpu_frames:
                e.extract(frames[reps % nc])
                reps += 1
            tc = time.perf_counter() - t0
            cpu = {"value": round(reps / tc, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                   "sample": "%d frames of the same batch, single thread (the reference extracts on one tracking thread)" % reps}
            # the same port with one extractor per host thread (ctypes releases the GIL inside the call)
            import threading

            nthr = max(1, min(16, os.cpu_count() or 1))
            per = max(4, args.cpu_frames // 4)
            exts = [oracle.Extractor(args.nfeatures, 1.2, 8, 20, 7, oracle.QUIRKS_FORK if args.preset == "fork" else oracle.QUIRKS_CANONICAL)
                    for _ in range(nthr)]

            def work(k):
                for i in range(per):
                    exts[k].extract(frames[(k * per + i) % nc])

            th = [threading.Thread(target=work, args=(k,)) for k in range(nthr)]
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            tm = time.perf_counter() - t0
            cpu["all_cores"] = {"value": round(nthr * per / tm, 3), "unit": "frames/s", "cores": nthr,
                                "sample": "%d frames per thread, one extractor per thread" % per}

    # ---- the same extraction off the headline operating point, and the other configurations -----------
    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = {}

        def wall(fn, steps, warm=2):
            preheat(fn)  # (the CPU legs above left the device idle)
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps

        sweep = {}
        for nb, steps in ((1, 200), (8, 100), (64, 50)):
            if nb > B:
                continue
            k = [0]

            def sub():
                ctx.extract_batch_device(d_batches[k[0] % 3][(k[0] * nb) % (B - nb + 1):][:nb], d_kps[:nb], d_desc[:nb], d_cnt[:nb], stream)
                k[0] += 1

            t = wall(sub, steps)
            sweep[str(nb)] = {"frames_per_s": round(nb / t, 1), "ms_per_batch": round(t * 1e3, 4)}
        sweep[str(B)] = {"frames_per_s": round(fps, 1), "ms_per_batch": round(dt / args.steps * 1e3, 4)}
        # twice the headline batch (a context of its own, two of the batches above joined): what the launch boundaries cost
        try:
            ctxL = binding.Context(args.nfeatures, 1.2, 8, 20, 7, W, H, 2 * B, local, quirks)
            d_two = [torch.cat([d_batches[i], d_batches[(i + 1) % 3]]) for i in range(3)]
            d_kL = torch.zeros((2 * B, cap, 28), dtype=torch.uint8, device=dev)
            d_dL = torch.zeros((2 * B, cap, 32), dtype=torch.uint8, device=dev)
            d_cL = torch.zeros(2 * B, dtype=torch.int32, device=dev)
            kL = [0]

            def big():
                ctxL.extract_batch_device(d_two[kL[0] % 3], d_kL, d_dL, d_cL, stream)
                kL[0] += 1

            t = wall(big, 10)
            ctxL.check_last_batch()
            sweep[str(2 * B)] = {"frames_per_s": round(2 * B / t, 1), "ms_per_batch": round(t * 1e3, 4)}
            ctxL.close()
            del d_two, d_kL, d_dL, d_cL
        except Exception as ex:  # an extra, never a reason to lose the benchmark line
            sweep[str(2 * B)] = {"error": str(ex)[-200:]}
        extras["batch_sweep"] = {"what": "device-resident frames, batches enqueued back to back, one synchronisation at the end",
                                 "by_batch": sweep}
        ctx.check_last_batch()
        # host-pointer entry points: frames start and results end in pageable host memory (PCIe-inclusive)
        nb = min(64, B)
        host_imgs = [frames[f] for f in range(nb)]
        hk = np.zeros((nb, ctx.cap), binding.KP_DTYPE)
        hd = np.zeros((nb, ctx.cap, 32), np.uint8)
        hc = np.zeros(nb, np.int32)
        t = wall(lambda: ctx.extract_batch_into(host_imgs, hk, hd, hc), 10)
        t_alloc = wall(lambda: ctx.extract_batch(host_imgs), 10)
        # the same call with the frames in page-locked memory (orb_alloc_host): no staging copy on the host
        pinned = binding.alloc_host((nb, H, W))
        pinned[:] = frames[:nb]
        pinned_imgs = [pinned[f] for f in range(nb)]
        t_pin = wall(lambda: ctx.extract_batch_into(pinned_imgs, hk, hd, hc), 20)
        extras["host_path"] = {"entry_point": "orb_extract_batch: frames and results in pageable host memory, result arrays kept by the caller",
                               "batch": nb, "frames_per_s": round(nb / t, 1), "ms_per_batch": round(t * 1e3, 3),
                               "ms_per_batch_with_fresh_result_arrays": round(t_alloc * 1e3, 3),
                               "pinned_frames_per_s": round(nb / t_pin, 1), "pinned_ms_per_batch": round(t_pin * 1e3, 3),
                               "pinned": "the same call, frames in page-locked host memory: fetched by the DMA engine where they lie"}
        one = [0]

        def single():
            ctx.extract(frames[one[0] % min(B, 16)])
            one[0] += 1

        t = wall(single, 200, warm=5)
        single_ms = {"orb_extract_ctypes": round(t * 1e3, 4)}
        demo = os.path.join(ROOT, "examples", "orb_demo")
        if os.path.exists(demo):  # the drop-in C++ ORBextractor::operator() and Frame constructor, timed by the demo itself
            import subprocess
            import tempfile

            with tempfile.TemporaryDirectory() as td:
                fin = os.path.join(td, "in.bin")
                with open(fin, "wb") as fh:
                    fh.write(np.array([W, H, args.nfeatures, 100], np.int32).tobytes())
                    fh.write(np.array([0.9], np.float32).tobytes())
                    fh.write(frames[0].tobytes())
                    fh.write(frames[1 % B].tobytes())
                try:
                    r = subprocess.run([demo, "--time", fin, "300"], capture_output=True, text=True, timeout=120)
                    if r.returncode == 0:
                        single_ms.update(json.loads(r.stdout.strip().splitlines()[-1]))
                    else:
                        single_ms["orb_demo_error"] = (r.stderr or r.stdout)[-200:]
                except Exception as ex:  # the demo is an extra, never a reason to lose the benchmark line
                    single_ms["orb_demo_error"] = str(ex)[-200:]
        extras["single_frame_ms"] = single_ms
        # the other preset (every switch of the fork's deviations flipped: INTER_LINEAR pyramid, canonical umax, ...),
        # same frames, same step
        other_preset = "canonical" if args.preset == "fork" else "fork"
        oq = binding.ORB_QUIRKS_CANONICAL if other_preset == "canonical" else binding.ORB_QUIRKS_FORK
        ctx2 = binding.Context(args.nfeatures, 1.2, 8, 20, 7, W, H, B, local, oq)
        k2 = [0]

        def step2():
            ctx2.extract_batch_device(d_batches[k2[0] % 3], d_kps, d_desc, d_cnt, stream)
            k2[0] += 1

        t = wall(step2, args.steps, warm=args.warmup)
        ctx2.check_last_batch()
        k2[0] = 0
        step2()
        torch.cuda.synchronize()
        c2 = d_cnt.cpu().numpy()
        e2 = oracle.Extractor(args.nfeatures, 1.2, 8, 20, 7, oracle.QUIRKS_CANONICAL if other_preset == "canonical" else oracle.QUIRKS_FORK)
        ko2, do2 = e2.extract(frames[0])
        kg2 = d_kps[0, : c2[0]].cpu().numpy().view(binding.KP_DTYPE).reshape(-1)
        extras["other_preset"] = {"preset": other_preset, "frames_per_s": round(B / t, 1), "ms_per_step": round(t * 1e3, 4),
                                  "steps": args.steps, "keypoints_per_frame": round(float(c2.mean()), 1),
                                  "parity_checked": bool(len(kg2) == len(ko2) and (kg2 == ko2).all() and
                                                         (d_desc[0, : c2[0]].cpu().numpy() == do2).all())}
        ctx2.close()

    # ---- the other configurations, on one GPU or sharded over the ranks ---------------------------------------
    # (they make contexts of their own; this one's streams would compete with theirs for the process's few hardware
    # queues)
    other = None
    if not args.no_extras:
        host_barrier()  # (rank 0 has been busy with the CPU baseline: the others wait for it on the host, not in a collective kernel)
        ctx.close()
        if not args.no_hamming:
            bf.close()
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_configs

            R = bench_configs.Ranks(rank, world, local, barrier, max_over_ranks, sum_over_ranks)
            note("configuration 2")
            other = [bench_configs.config2(R)]
            if world == 1:
                other.append(bench_configs.config3(R))  # pairs are independent: more GPUs would be replicas
            note("configuration 4")
            other.append(bench_configs.config4(R, job=args.job_frames))
        except Exception as ex:
            if grouped:
                raise  # the ranks must stay in step: a rank that drops out would leave the others in a collective
            other = {"error": str(ex)[-300:]}
        if extras is None and rank == 0:
            extras = {}
        if rank == 0:
            extras["other_configs"] = other

    # ---- every visible GPU from ONE process, plain C++ (orb_bf_multi_*: RCCL all-gather inside the library) --------
    # rank 0 runs the example over as many devices as there are ranks (all visible ones in a single-rank run);
    # the other ranks wait on the host
    if not args.no_extras and not args.no_hamming and not rehearse:
        ndev = torch.cuda.device_count()
        use = min(ndev, world) if world > 1 else 1
        exe = os.path.join(ROOT, "examples", "orb_bf_multi")
        if rank == 0 and use > 1 and os.path.exists(exe):
            import subprocess

            try:
                r = subprocess.run([exe, ",".join(str(d) for d in range(use)), "4000000", "20000"], capture_output=True,
                                   text=True, timeout=240)
                extras["one_process_all_gpus"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else \
                    {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as ex:
                extras["one_process_all_gpus"] = {"error": str(ex)[-300:]}
        host_barrier()

    if rank == 0:
        out = {
            "metric": "ORB extract frames/s (640x480, 1k feats)", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "ORBextractor::operator() over %d frames %dx%d per GPU per step (three distinct batches in turn, inputs and outputs resident in HBM), nFeatures=%d, 8 levels, scale 1.2, iniThFAST 20, minThFAST 7, preset %s"
                                   % (B, W, H, args.nfeatures, args.preset),
                       "batch_per_gpu": B, "keypoints_per_frame": round(nkp, 1),
                       "preheat": "%.2f s of the same steps, untimed, before the warm-up steps of every timed section (the engine clock "
                                  "takes about 35 ms of load to reach its steady value)" % PREHEAT_S},
            "roofline": roofline, "cpu_baseline": cpu, "parity_checked": parity, "hamming": hamming,
        }
        if extras:
            out.update(extras)
        if rehearse:
            out["rehearsal"] = "ranks share GPUs, gloo collectives: not a measurement"
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if grouped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
