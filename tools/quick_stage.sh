#!/bin/bash
# Development aid: headline step and stage times without the extras (run on the GPU box).
timeout -k 10 200 python bench.py --no-hamming --no-cpu --no-extras --steps 30 --warmup 5 > gpurun_out/b1.json && python -c "
import json;d=json.load(open('gpurun_out/b1.json'));print(round(d['value']),d['ms_per_step'],d['roofline']['stage_ms'])"
