#!/bin/bash
timeout 600 python -m pytest tests/test_modules.py tests/test_bench_parity.py tests/test_e2e_av.py -q -m gpu -x -k "hpf or precise or av" 2>&1 | tail -2
timeout 400 python bench.py --mode hpf --no-cpu-baseline --no-roofline --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('hpf', d['ms_per_step'], d['value'], 'logits', d['parity']['dec_logits_rel_l2'], 'grad cos', d['parity']['grad_sample_cos_min'], 'loss err', d['parity']['loss_rel_err'])"
timeout 400 python bench.py --no-cpu-baseline --no-roofline --steps 16 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('default', d['ms_per_step'], 'hpf leg', d['precise']['ms_per_step'], d['precise']['parity']['dec_logits_rel_l2'])"
