#!/bin/bash
# round-3 session 15: pipelined weight-gradient kernel -- GPU tests of the conv + module suites, step-level check
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_kernels.py tests/test_modules.py -q -m gpu -x 2>&1 | tail -3 > gpurun_out/s15_tests.txt
timeout 300 python tools/microbench_wgrad.py 2>&1 | grep -v amdgpu.ids > gpurun_out/s15_wgrad.txt
for i in 1 2; do
  timeout 300 python bench.py --fixed A --no-cpu-baseline --no-roofline --no-parity --no-precise-leg --steps 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fixedA', d['ms_per_step'])" >> gpurun_out/s15_ab.txt
done
timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-precise-leg --steps 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('default', d['ms_per_step'], d['value'], d['parity']['grad_cosine_min'] if 'parity' in d else '')" >> gpurun_out/s15_ab.txt
cat gpurun_out/s15_tests.txt gpurun_out/s15_wgrad.txt gpurun_out/s15_ab.txt
