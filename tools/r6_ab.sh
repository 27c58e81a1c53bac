#!/bin/bash
# same-box A/B of avsr_tune settings on the replayed step: tools/r6_ab.sh "<bench flags>" "<tune spec>" "<tune spec>" ...  (a spec of "-" = no knobs)
cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out
flags=$1; shift
for rep in 1 2; do for t in "$@"; do
  spec=$t; [ "$t" = "-" ] && spec=""
  AVSR_TUNE=$spec timeout 300 python bench.py $flags --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/ab.json 2>gpurun_out/ab.err
  echo "tune [$t] $flags: $(python -c "import json;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print(d['ms_per_step'])")"
done; done
