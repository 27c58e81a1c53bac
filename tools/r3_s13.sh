#!/bin/bash
# round-3 session 13: 8-wave weight-gradient kernel -- GPU tests, microbench, step A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_kernels.py -q -m gpu -k "wgrad" -x 2>&1 | tail -5 > gpurun_out/s13_tests.txt
timeout 300 python tools/microbench_wgrad.py > gpurun_out/s13_wgrad.txt 2>&1
for k in 0 1 0 1; do
  AVSR_TUNE="16=$k" timeout 300 python bench.py --fixed A --no-cpu-baseline --no-roofline --no-parity --no-precise-leg --steps 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('knob16=$k', d['ms_per_step'])" >> gpurun_out/s13_ab.txt
done
cat gpurun_out/s13_tests.txt gpurun_out/s13_wgrad.txt gpurun_out/s13_ab.txt
