#!/bin/bash
# hpf mode with bf16 twins from the attention / head-bias / depthwise producers: GPU tests + bench
timeout 900 python -m pytest tests/test_modules.py tests/test_attention.py tests/test_convmod_kernels.py tests/test_bench_parity.py -q -m gpu -x -k "hpf or attention or dwconv or precise" 2>&1 | tail -2
for i in 1 2; do
timeout 400 python bench.py --mode hpf --no-cpu-baseline --no-roofline --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('hpf', d['ms_per_step'], d['value'], 'logits', d['parity']['dec_logits_rel_l2'], 'grad cos', d['parity']['grad_sample_cos_min'])"
done
