#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_convmod_kernels.py tests/test_optim.py tests/test_modules.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/c12_bench.log 2>&1; tail -1 gpurun_out/c12_bench.log | cut -c100-260
