"""Development aid: time every tuning variant of the brute-force scan kernel."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:] or ["q4g4", "q4g4p", "q4g4m", "q4g4pm", "q8g2", "q8g2pm", "q8g4m", "q4g8m", "q2g8m", "q2g8pm", "q6g4pm"]
code = r'''
import sys, time, torch
sys.path.insert(0, %r)
from orb_slam2_amd import binding
nq, nd = 100000, 10000000
g = torch.Generator(device="cuda"); g.manual_seed(1)
db = torch.randint(0, 256, (nd, 32), dtype=torch.uint8, device="cuda", generator=g)
q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device="cuda", generator=g)
bf = binding.BruteForce(0); bf.set_database_device(db, 0)
out = torch.zeros(nq, dtype=torch.int64, device="cuda")
bf.query_packed_device(q, out); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); bf.query_packed_device(q, out); bf.query_packed_device(q, out); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 2
print("%%s %%.1f ms %%.0f Gcmp/s checksum %%d" %% (sys.argv[1], ms, nq * nd / ms / 1e6, int(out.sum().item())))
''' % ROOT
for n in names:
    env = dict(os.environ, ORBHIP_BF_VARIANT=n)
    r = subprocess.run([sys.executable, "-c", code, n], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
