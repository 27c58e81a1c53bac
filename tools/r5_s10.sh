#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
for t in 300 0 200 300 0; do AVSR_TN_SPLIT_MAX_TILES=$t timeout 300 python bench.py --fixed A --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/s10_$t.json 2>gpurun_out/s10.err; echo "tn_split_tiles=$t $(python -c "import json;d=json.load(open('gpurun_out/s10_$t.json'));print(d['ms_per_step'])")"; done
