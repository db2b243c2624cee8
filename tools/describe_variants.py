"""Development aid: describe-stage time of the library variants built by tools/describe_variants.sh.
usage (GPU box): python tools/describe_variants.py <variant number>   -> one line; output hash for equality"""
import sys, zlib
sys.path.insert(0, '.')
import numpy as np, torch
from orb_slam2_amd import binding
v = sys.argv[1]
binding.LIB_PATH = "tools/variants/liborbhip_%s.so" % v
from orb_slam2_amd.synth import frame_seed, synth_scene
B = 256
frames = np.stack([synth_scene(640, 480, frame_seed(1, f)) for f in range(B)])
d = torch.from_numpy(frames).cuda()
ctx = binding.Context(1000, 1.2, 8, 20, 7, 640, 480, B, 0, 0x1f)
cap = ctx.cap
k = torch.zeros((B, cap, 28), dtype=torch.uint8, device='cuda'); ds = torch.zeros((B, cap, 32), dtype=torch.uint8, device='cuda')
c = torch.zeros(B, dtype=torch.int32, device='cuda')
for i in range(3): ctx.extract_batch_device(d, k, ds, c)
torch.cuda.synchronize()
h = zlib.crc32(ds.cpu().numpy().tobytes()) ^ zlib.crc32(k.cpu().numpy().tobytes())
ctx.enable_stage_timing(True)
for i in range(20): ctx.extract_batch_device(d, k, ds, c)
ms, _ = ctx.stage_timings()
print("variant", v, "describe %.4f ms" % ms["describe"], "crc %08x" % h)
