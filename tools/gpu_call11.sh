#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_conv_kernels.py tests/test_bench_parity.py -x -q -m gpu 2>&1 | tail -2
for w in 1 0; do AVSR_WGRAD_STREAM=$w timeout 300 python bench.py --shapes 4 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/c11_bench_w$w.log 2>&1; tail -1 gpurun_out/c11_bench_w$w.log | cut -c100-260; done
