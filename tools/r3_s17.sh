#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_convmod_kernels.py tests/test_conv_kernels.py -q -m gpu -x -k "dwconv or multi_weight_permute" 2>&1 | tail -3
timeout 300 python tools/microbench_small.py 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
timeout 300 python bench.py --fixed A --no-cpu-baseline --no-roofline --no-parity --no-precise-leg --steps 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fixedA', d['ms_per_step'])"
done
