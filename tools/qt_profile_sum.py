"""Development aid: where the workgroups of a full quadtree launch spend their time.  Needs a library built with
-DQT_PROFILE -DQT_PROFILE_SUM (make -C orb_slam2_amd/csrc clean; make -C orb_slam2_amd/csrc EXTRA="-DQT_PROFILE -DQT_PROFILE_SUM")."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import binding  # noqa: E402
from orb_slam2_amd.synth import frame_seed, synth_scene  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
w, h, nf = 640, 480, 1000
ctx = binding.Context(nf, 1.2, 8, 20, 7, w, h, batch, 0, binding.ORB_QUIRKS_FORK)
scenes = [synth_scene(w, h, frame_seed(1, i)) for i in range(8)]
imgs = [scenes[i % 8] for i in range(batch)]
for _ in range(3):
    ctx.extract_batch(imgs)
out = (C.c_ulonglong * 32)()
binding.lib().orb_debug_quadtree_profile(out, 1)
ctx.extract_batch(imgs)
binding.lib().orb_debug_quadtree_profile(out, 0)
v = [int(x) for x in out]
names = {1: "count into buckets", 4: "scan + set-up", 13: "shortcut: classify", 14: "shortcut: positions",
         15: "shortcut: write list", 5: "remaining passes (coarse + fine)", 6: "select", 7: "emit", 2: "sorted form: scatter",
         3: "sorted form: rank"}
tot = sum(v[k] for k in names)
print("batch %d: %.1f M engine cycles over all workgroups (%.0f per (frame, level))" % (batch, tot / 1e6, tot / (batch * 8)))
for k in [1, 4, 13, 14, 15, 5, 6, 7, 2, 3]:
    print("  %-36s %6.1f %%  %8.0f cycles per (frame, level)" % (names[k], 100.0 * v[k] / max(tot, 1), v[k] / (batch * 8)))
ctx.close()
