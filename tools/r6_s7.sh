#!/bin/bash
# round 6 session 7: deterministic mode on the device (bit-identical runs under graph replay), its cost on the full-size step, full GPU suite
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_deterministic.py -x -q -m gpu -s 2>&1 | grep -v "^  File\|Extension modules\|amdgpu.ids" | tail -8
for f in "" "--deterministic"; do timeout 400 python bench.py $f --no-cpu-baseline --no-roofline --no-bf16-leg --no-parity --steps 12 --warmup 3 2>/dev/null | tail -1 > $O/r6_s7.json; python -c "
import json; d=json.loads(open('$O/r6_s7.json').read()); print('bench [$f]', d['ms_per_step'], d['value'], d['config']['deterministic'], d['config']['final_loss'])"; done
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6) > $O/r6_s7_gputests.txt 2>&1; tail -5 $O/r6_s7_gputests.txt
