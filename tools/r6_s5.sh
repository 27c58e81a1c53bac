#!/bin/bash
# round 6 session 5: new GPU tests (capacity policy, eviction release, trainer stand-in), same-box A/B of the round's tile changes, bench line with the paired-backward roofline object
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_eval_loops.py tests/test_trainer_standin.py tests/test_optim.py -x -q -m gpu 2>&1 | tail -6
bash tools/r6_ab.sh "" - "17=-1,18=7,21=23" 2>&1 | tail -4
timeout 500 python bench.py --no-cpu-baseline --no-bf16-leg --steps 12 --warmup 3 2>$O/r6_s5_bench.err | tail -1 > $O/r6_s5_bench.json
python - <<'PY'
import json; d=json.loads(open('gpurun_out/r6_s5_bench.json').read())
print(d['ms_per_step'], d['value']); 
for k in ('roofline','roofline_fwd','roofline_pair','roofline_hbm'):
    if k in d: r=d[k]; print(k, r['kernel'][:60], r['launches'], r['avg_us'], r.get('total_ms'), r['achieved'], r['frac'], r.get('traffic'), r.get('algorithmic_bytes_per_launch'))
PY
