#!/bin/bash
# round 6, session 12: CTC branch on a second stream -- tests, same-box A/B of the replayed step, DP1 legs
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_deterministic.py tests/test_train_eval_loops.py tests/test_trainer_standin.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2; do for f in 1 0; do
  AVSR_SIDE_BRANCH=$f timeout 300 python bench.py --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/ab.json 2>gpurun_out/ab.err
  echo "side=$f: $(python -c "import json;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"; tail -2 gpurun_out/ab.err
done; done
AVSR_DDP=buckets-graph1 timeout 300 python bench.py --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/ab.json 2>gpurun_out/ab.err
echo "DP1 buckets-graph1 side=1: $(python -c "import json;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"; tail -2 gpurun_out/ab.err
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-bf16-leg > gpurun_out/s12_parity.json 2>gpurun_out/s12_parity.err; tail -c 1500 gpurun_out/s12_parity.json
