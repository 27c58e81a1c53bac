#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
B="python bench.py --no-cpu-baseline --no-roofline --no-parity --no-precise-leg --fixed A"
for v in 0 256 384 768 0 256; do
AVSR_TUNE=15=$v timeout 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('knob15=$v', d['ms_per_step'])"
done
