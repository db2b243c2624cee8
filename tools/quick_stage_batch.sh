#!/bin/bash
# Development aid: step and stage times at several batch sizes (run on the GPU box).
for b in "$@"; do
timeout -k 10 200 python bench.py --no-hamming --no-cpu --no-extras --steps 50 --warmup 10 --batch $b > gpurun_out/b_$b.json && python -c "
import json;d=json.load(open('gpurun_out/b_$b.json'));print($b, round(d['value']),d['ms_per_step'],d['roofline']['stage_ms'])" || exit 1
done
