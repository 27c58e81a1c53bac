#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
for t in "1=2" "1=0"; do
AVSR_TUNE=$t timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/c4_bench_$t.log 2>&1; tail -1 gpurun_out/c4_bench_$t.log | cut -c100-260
done
for t in "1=2" "1=0"; do
  rm -rf gpurun_out/pmcx_$t
  AVSR_TUNE=$t timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmcx_$t -o r -- python tools/pmc_step.py > gpurun_out/c4_pmc_$t.log 2>&1
done
timeout 600 python bench.py --steps 10 --warmup 3 --precise --no-cpu-baseline --no-parity > gpurun_out/c4_bench_precise.log 2>&1; tail -1 gpurun_out/c4_bench_precise.log | cut -c1-400
