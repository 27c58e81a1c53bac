#!/bin/bash
mkdir -p gpurun_out; cd /root/repo; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_config0.py tests/test_e2e_av.py tests/test_rccl_single.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python tools/bench_av.py --steps 8 --warmup 3 > gpurun_out/c10_av.log 2>&1; tail -2 gpurun_out/c10_av.log | cut -c1-600
timeout 300 python bench.py --modality audio --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/c10_audio.log 2>&1; tail -1 gpurun_out/c10_audio.log | cut -c1-1500
timeout 400 python bench.py --steps 24 --warmup 8 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/c10_bench8.log 2>&1; tail -1 gpurun_out/c10_bench8.log | cut -c1-900
for f in A B; do timeout 300 python bench.py --fixed $f --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/c10_fixed$f.log 2>&1; tail -1 gpurun_out/c10_fixed$f.log | cut -c100-330; done
timeout 200 python tools/eager_overhead.py > gpurun_out/c10_eager.log 2>&1; tail -4 gpurun_out/c10_eager.log
