#!/bin/bash
# usage: gpurun_single.sh <outdir>   (development aid: one-frame latency through ctypes and through the C++ classes)
set -e
out=gpurun_out/$1
mkdir -p $out
timeout -k 10 120 python tools/single_frame.py > $out/single.txt 2>&1 || { cat $out/single.txt; exit 1; }
python - <<PY
import sys; sys.path.insert(0, ".")
import numpy as np, struct
from orb_slam2_amd.synth import frame_seed, synth_scene
a = synth_scene(640, 480, frame_seed(1, 0)); b = synth_scene(640, 480, frame_seed(1, 1))
open("$out/in.bin", "wb").write(struct.pack("<4if", 640, 480, 1000, 100, 0.9) + a.tobytes() + b.tobytes())
PY
timeout -k 10 60 examples/orb_demo --time $out/in.bin 500 >> $out/single.txt 2>&1
grep -v amdgpu.ids $out/single.txt
