#!/bin/bash
# round 6, session 11: fused convolution-module middle -- tests, microbench, same-box A/B of the replayed step
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_convmod_kernels.py tests/test_modules.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python tools/microbench_convmod.py 2>&1 | tee gpurun_out/r6_microbench_convmod.txt | tail -12
for rep in 1 2; do for f in 1 0; do
  AVSR_CONVMOD_FUSED=$f timeout 300 python bench.py --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/ab.json 2>gpurun_out/ab.err
  echo "fused=$f: $(python -c "import json;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
done; done
