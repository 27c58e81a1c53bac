#!/bin/bash
# round 6 session 2: step time with the 256x128 8-wave two-plane tiles (default + fixed A), parity tests of the mixed mode
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for f in "" "--fixed A"; do timeout 300 python bench.py $f --no-cpu-baseline --no-roofline --no-bf16-leg --steps 16 --warmup 4 2>/dev/null | tail -1 > $O/r6_s2_bench.json; python - <<PY
import json; d=json.loads(open('$O/r6_s2_bench.json').read()); print('bench $f', d['ms_per_step'], d['value'], d.get('parity'))
PY
done
timeout 900 python -m pytest tests/test_bench_parity.py tests/test_mixed_mode.py tests/test_conv_kernels.py -x -q -m gpu 2>&1 | tail -5
