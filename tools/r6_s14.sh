#!/bin/bash
# round 6, session 14: fill launches removed (loss-gradient pads written by the kernels, 1x1 weight gradients from the zero arena)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_loss_kernels.py tests/test_conv_kernels.py tests/test_bench_parity.py tests/test_e2e_gpu.py tests/test_train_eval_loops.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do
  timeout 300 python bench.py --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/ab.json 2>gpurun_out/ab.err
  echo "bench: $(python -c "import json;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
done
bash tools/gpu_timeline.sh r6_s14 --no-bf16-leg > /dev/null 2>&1; head -1 gpurun_out/r6_s14_timeline.txt; grep -c FillFunctor gpurun_out/r6_s14_timeline.txt
