#!/bin/bash
# A/B: trunk weight gradients on a side stream (AVSR_WGRAD_STREAM=1)
for v in 0 1 0 1; do
AVSR_WGRAD_STREAM=$v timeout 300 python bench.py --fixed A --no-cpu-baseline --no-roofline --no-precise-leg --steps 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('wgrad side stream=$v', d['ms_per_step'], 'grad cos', d['parity']['grad_sample_cos_min'], 'loss err', d['parity']['loss_rel_err'])"
done
