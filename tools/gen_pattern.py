#!/usr/bin/env python3
"""Extract the 256x4 rBRIEF sampling table (data only) from the reference's
src/ORBextractor.cc:108-366 (`bit_pattern_31_`) and write it as a bare
comma-separated integer list, 4 ints per line, to the two .inc files that the
CPU oracle and the HIP kernels include.  Run in the build container only
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
for p in ("oracle/orb_pattern.inc", "orb_slam2_amd/csrc/orb_pattern.inc"):
    (root / p).write_text(out)
import zlib
print("wrote 1024 ints, crc32=%08x" % zlib.crc32(out.encode()))
