"""Development aid: time the brute-force scan (100k queries x 10M rows by default) and print a checksum of the
packed results, so that kernel variants can be compared for speed and equality."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_amd import binding  # noqa: E402

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 10000000
g = torch.Generator(device="cuda")
g.manual_seed(1)
db = torch.randint(0, 256, (nd, 32), dtype=torch.uint8, device="cuda", generator=g)
q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device="cuda", generator=g)
bf = binding.BruteForce(0)
bf.set_database_device(db, 0)
out = torch.zeros(nq, dtype=torch.int64, device="cuda")
bf.query_packed_device(q, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    bf.query_packed_device(q, out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("%d x %d: %.1f ms %.0f Gcmp/s checksum %d" % (nq, nd, ms, nq * nd / ms / 1e6, int(out.sum().item())), flush=True)
