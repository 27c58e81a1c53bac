#!/bin/bash
# the graph-replayed data-parallel path (one rank) in the other configurations: audio, hpf numerics, max-frames 3200
export AVSR_BENCH_FORCE_DP=1
B="python bench.py --no-cpu-baseline --steps 12 --warmup 3"
for cfg in "--modality audio" "--mode hpf" "--max-frames 3200" "--modality audio --babble"; do
  timeout 600 $B $cfg 2> /tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', d['ms_per_step'], d['value'], 'loss', d['config']['final_loss'], 'graph' if 'hipGraph replay' in d['config']['workload'] else 'EAGER')"
  grep -i "failed\|falling\|Error" /tmp/err.txt | head -2
done
