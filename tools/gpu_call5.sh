#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_convmod_kernels.py tests/test_modules.py -x -q -m gpu -k "pool" > gpurun_out/c5_tests.log 2>&1; tail -2 gpurun_out/c5_tests.log
rm -rf gpurun_out/c5_prof
AVSR_FUSE_STEM_POOL=1 timeout 400 rocprofv3 --kernel-trace -d gpurun_out/c5_prof -o r -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/c5_prof.log 2>&1
db=$(find gpurun_out/c5_prof -name "*.db" | head -1)
python tools/rocpd_summary.py "$db" gpurun_out/c5_kernel_stats_pool1.txt > /dev/null 2>&1
find gpurun_out/c5_prof -name "*.db" -delete
grep -i "pool\|stem" gpurun_out/c5_kernel_stats_pool1.txt | head
AVSR_FUSE_STEM_POOL=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/c5_bench_pool1.log 2>&1; tail -1 gpurun_out/c5_bench_pool1.log | cut -c100-260
