#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/c14_tests.log 2>&1; tail -2 gpurun_out/c14_tests.log
timeout 300 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/c14_bench.log 2>&1; tail -1 gpurun_out/c14_bench.log | cut -c100-260
