#!/bin/bash
# round-2 GPU call 1: tests incl. fused stem-pool on hardware, bench A/B of AVSR_FUSE_STEM_POOL, ring-depth sweep, PMC passes
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/c1_tests.log 2>&1; tail -3 gpurun_out/c1_tests.log
timeout 300 python tools/microbench_ring.py > gpurun_out/c1_ring.log 2>&1; tail -12 gpurun_out/c1_ring.log
AVSR_FUSE_STEM_POOL=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c1_bench_pool0.log 2>&1; tail -1 gpurun_out/c1_bench_pool0.log | cut -c1-200
AVSR_FUSE_STEM_POOL=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c1_bench_pool1.log 2>&1; tail -1 gpurun_out/c1_bench_pool1.log | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_$c -o r -- python tools/pmc_step.py > gpurun_out/c1_pmc_$c.log 2>&1
  tail -2 gpurun_out/c1_pmc_$c.log | cut -c1-300
  ls -la $(find gpurun_out/pmc_$c -name "*.db" | head -1)
done
