#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_basic.py -q -m gpu -x -k "pair" 2>&1 | tail -2
for t in "20=0" "20=1" "20=0" "20=1"; do AVSR_TUNE=$t timeout 300 python bench.py --fixed A --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/s12.json 2>gpurun_out/s12.err; echo "tune $t $(python -c "import json;d=json.load(open('gpurun_out/s12.json'));print(d['ms_per_step'])")"; done
