#!/bin/bash
# round 6 session 1: baseline bench of the round-5 tree on this round's box + 8-wave two-plane tile microbench
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-bf16-leg --no-parity --steps 16 --warmup 4 2>/dev/null | tail -1 > $O/r6_s1_bench_default.json; cut -c1-200 $O/r6_s1_bench_default.json
timeout 600 python tools/microbench_h16x2.py > $O/r6_s1_microbench_h16x2.txt 2>&1; tail -20 $O/r6_s1_microbench_h16x2.txt
