"""Development aid: wall time of one-frame calls through the C-ABI (orb_extract) at 640x480, and a check that
repeated calls return the same bytes as the first (the first call runs the plain launch sequence, later ones may be
replayed from a captured graph)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from orb_slam2_amd import binding
from orb_slam2_amd.synth import frame_seed, synth_scene

frames = [synth_scene(640, 480, frame_seed(1, f)) for f in range(8)]
ctx = binding.Context(1000, 1.2, 8, 20, 7, 640, 480, 1, 0, 0x1f)
first = [ctx.extract(f) for f in frames]
for rep in range(3):
    for i, f in enumerate(frames):
        k, d = ctx.extract(f)
        assert np.array_equal(k, first[i][0]) and np.array_equal(d, first[i][1]), (rep, i)
t0 = time.perf_counter()
for i in range(500):
    ctx.extract(frames[i % 8])
dt = (time.perf_counter() - t0) / 500
ctx.enable_stage_timing(True)
for i in range(50):
    ctx.extract(frames[i % 8])
ms, _ = ctx.stage_timings()
print("ctypes orb_extract %.1f us; stages %s" % (dt * 1e6, {k: round(v * 1e3, 1) for k, v in ms.items()}))
ctx.close()
