#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 300 python tools/rccl_world1.py > gpurun_out/c8_rccl.log 2>&1; tail -3 gpurun_out/c8_rccl.log | cut -c1-900
timeout 300 python tools/rccl_world1.py --full > gpurun_out/c8_rccl_full.log 2>&1; tail -2 gpurun_out/c8_rccl_full.log | cut -c1-900
timeout 300 python -m pytest tests/test_kernels_basic.py tests/test_modules.py -x -q -m gpu -k "layernorm or handoff" 2>&1 | tail -2
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/c8_bench.log 2>&1; tail -1 gpurun_out/c8_bench.log | cut -c100-260
