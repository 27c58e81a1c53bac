#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s6_bench_default.json 2> gpurun_out/s6_bench_default.err; echo "bench rc=$?"
python -c "
import json
d = json.loads(open('gpurun_out/s6_bench_default.json').readline())
print('default', d['ms_per_step'], d['value'], '| parity logits', d['parity']['dec_logits_rel_l2'], 'ctc_logp', d['parity']['ctc_logp_rel_l2'])
print('  bf16 leg', d['bf16']['ms_per_step'], d['bf16']['value'], d['bf16']['parity']['dec_logits_rel_l2'])
print('  roofline', {k: d['roofline'][k] for k in ('kernel', 'launches', 'avg_us', 'achieved', 'frac')})
"
timeout 300 python bench.py --fixed A --steps 12 --warmup 3 --no-cpu-baseline --no-roofline --no-bf16-leg --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fixed A', d['ms_per_step'])"
bash tools/r4_pmc.sh
