#!/usr/bin/env python3
"""Extract the 256x4 rBRIEF sampling table (data only) from the reference's
src/ORBextractor.cc:108-366 (`bit_pattern_31_`) and write it as a bare
comma-separated integer list, 4 ints per line, to the .inc files that the
CPU oracle (x0, y0, x1, y1 per pair) and the HIP kernel (x0, x1, y0, y1) include.  Run in the build container only
(the reference tree does not travel to the GPU box); the outputs are committed.
"""
import re, sys, pathlib

ref = pathlib.Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/ORBextractor.cc")
txt = ref.read_text(encoding="utf-8", errors="replace")
m = re.search(r"bit_pattern_31_\s*\[\s*256\s*\*\s*4\s*\]\s*=\s*\{(.*?)\};", txt, re.S)
body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
body = re.sub(r"//[^\n]*", "", body)
vals = [int(v) for v in re.findall(r"-?\d+", body)]
assert len(vals) == 1024, len(vals)
assert max(abs(v) for v in vals) <= 13
lines = [", ".join(f"{v:3d}" for v in vals[i:i + 4]) + "," for i in range(0, 1024, 4)]
out = "\n".join(lines) + "\n"
root = pathlib.Path(__file__).resolve().parent.parent
(root / "oracle/orb_pattern.inc").write_text(out)
# the kernel reads a pair as (x0, x1, y0, y1): the coordinates of the two test points side by side, ready for packed math
lines2 = [", ".join(f"{v:3d}" for v in (vals[i], vals[i + 2], vals[i + 1], vals[i + 3])) + "," for i in range(0, 1024, 4)]
(root / "orb_slam2_amd/csrc/orb_pattern_xxyy.inc").write_text("\n".join(lines2) + "\n")
import zlib
print("wrote 1024 ints, crc32=%08x" % zlib.crc32(out.encode()))
