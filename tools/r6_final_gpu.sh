#!/bin/bash
# last GPU session of round 6: full GPU suite + smoke, the default bench line, the fixed-batch lines (mixed and bf16)
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6; python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3) > $O/r6_final_gputests.txt 2>&1; tail -4 $O/r6_final_gputests.txt
timeout 400 python bench.py > $O/r6_final_bench_default_lastbox.json 2> $O/r6_final_bench_default_lastbox.err; tail -1 $O/r6_final_bench_default_lastbox.json | cut -c1-330
for f in A B; do timeout 300 python bench.py --fixed $f --no-cpu-baseline --no-roofline --no-bf16-leg --no-parity --steps 16 --warmup 4 2>/dev/null | tail -1 > $O/r6_final_bench_fixed${f}_lastbox.json; python -c "import json; d=json.loads(open('$O/r6_final_bench_fixed${f}_lastbox.json').read()); print('fixed $f', d['ms_per_step'], d['value'])"; done
