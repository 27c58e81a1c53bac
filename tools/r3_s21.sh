#!/bin/bash
# stability: 300 replayed steps (N = 1) and 200 replayed data-parallel steps on one rank; losses must stay finite and fall
mkdir -p gpurun_out
timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-parity --no-precise-leg --steps 300 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('N=1 300 steps:', d['ms_per_step'], d['value'], 'final loss', d['config']['final_loss'])"
AVSR_BENCH_FORCE_DP=1 timeout 600 python bench.py --no-cpu-baseline --steps 200 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('forced-DP graph 200 steps:', d['ms_per_step'], d['value'], 'final loss', d['config']['final_loss'])"
