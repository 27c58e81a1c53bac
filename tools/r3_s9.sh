#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
bash tools/gpu_timeline.sh r3f_hpf --mode hpf
timeout 900 python bench.py --no-cpu-baseline --steps 16 --warmup 4 > $O/r3f_bench_default.json 2> $O/r3f_bench_default.err; echo "default rc=$?"; python -c "
import json
d=json.loads(open('$O/r3f_bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['precise']['ms_per_step'], d['precise']['parity']['dec_logits_rel_l2'])"
