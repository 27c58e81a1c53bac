#!/bin/bash
# round 5, GPU session 6: split8 activations between the split-plane trunk stages -- tests, parity, A/B step time on one box
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out; export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests/test_mixed_mode.py -q -m gpu -x 2>&1 | tail -2
timeout 900 python -m pytest tests/test_bench_parity.py -q -m gpu -x -k "mixed" -s 2>&1 | grep "PARITY\|passed\|failed" | cut -c1-420
for ps in 1 0 1 0; do AVSR_PRESPLIT=$ps timeout 300 python bench.py --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg > gpurun_out/s6_bench_ps$ps.json 2>gpurun_out/s6_bench.err; echo "presplit=$ps $(cut -c100-260 gpurun_out/s6_bench_ps$ps.json)"; done
for ps in 1 0; do AVSR_PRESPLIT=$ps timeout 300 python bench.py --fixed A --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg > gpurun_out/s6_benchA_ps$ps.json 2>>gpurun_out/s6_bench.err; echo "fixedA presplit=$ps $(cut -c100-260 gpurun_out/s6_benchA_ps$ps.json)"; done
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_modules.py tests/test_e2e_av.py -q -m gpu -x 2>&1 | tail -2
