#!/bin/bash
# round 5, GPU session 1: two-plane f16 weights -- kernel microbench, whole-tensor parity of candidate policies (A, B, AA), step time
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
X2="encoder=f16x2,decoder=f16x2,trunk3=f16x2,trunk4=f16x2,atrunk3=f16x2,atrunk4=f16x2"
R4="encoder=f16,decoder=f16,trunk3=f16,trunk4=f16,atrunk3=f16,atrunk4=f16"
timeout 300 python tools/microbench_h16x2.py > gpurun_out/s1_microbench.txt 2>&1; tail -15 gpurun_out/s1_microbench.txt
timeout 600 python tools/mixed_sweep.py --tags=A,B,AA "$R4" "$X2" "encoder=f16x2,decoder=f16x2,trunk3=f16,trunk4=f16,atrunk3=f16,atrunk4=f16" \
  "encoder=f16x2,decoder=f16,trunk3=f16x2,trunk4=f16x2,atrunk3=f16x2,atrunk4=f16x2" "encoder=f16x2,decoder=f16x2,dec_out=f16x2,trunk3=f16x2,trunk4=f16x2,atrunk3=f16x2,atrunk4=f16x2" \
  "encoder=f16x2,decoder=f16x2,trunk2=f16x2,trunk3=f16x2,trunk4=f16x2,atrunk2=f16x2,atrunk3=f16x2,atrunk4=f16x2" > gpurun_out/s1_sweep.txt 2>gpurun_out/s1_sweep.err; cat gpurun_out/s1_sweep.txt; tail -3 gpurun_out/s1_sweep.err
timeout 420 python bench.py > gpurun_out/s1_bench_default.json 2>gpurun_out/s1_bench.err; cat gpurun_out/s1_bench_default.json; tail -3 gpurun_out/s1_bench.err
AVSR_MIXED_POLICY=$R4 timeout 300 python bench.py --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg > gpurun_out/s1_bench_r4policy.json 2>>gpurun_out/s1_bench.err; cat gpurun_out/s1_bench_r4policy.json
timeout 300 python bench.py --fixed A --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg > gpurun_out/s1_bench_fixedA.json 2>>gpurun_out/s1_bench.err; cat gpurun_out/s1_bench_fixedA.json
timeout 900 python -m pytest tests/test_mixed_mode.py -q -m gpu -x 2>&1 | tail -3
