#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_basic.py -q -m gpu -x -k "pair" 2>&1 | tail -2
for t in "19=1" "19=0" "19=2" "19=1" "19=0"; do AVSR_TUNE=$t timeout 300 python bench.py --fixed A --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/s11.json 2>gpurun_out/s11.err; echo "tune $t $(python -c "import json;d=json.load(open('gpurun_out/s11.json'));print(d['ms_per_step'])")"; done
bash tools/gpu_timeline.sh s11_tn128 --no-bf16-leg > /dev/null 2>&1; grep "gemm_pair" gpurun_out/s11_tn128_timeline.txt | head -6 | cut -c1-150
