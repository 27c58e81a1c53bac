#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bench_parity.py -x -q -s > gpurun_out/c3_parity.log 2>&1; grep "PARITY\|passed\|failed" gpurun_out/c3_parity.log | cut -c1-900
timeout 300 python tools/microbench_xcd.py > gpurun_out/c3_xcd.log 2>&1; tail -6 gpurun_out/c3_xcd.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/c3_bench.log 2>&1; tail -1 gpurun_out/c3_bench.log
