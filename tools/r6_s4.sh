#!/bin/bash
# round 6 session 4: is the noisy batch-A gradient sample of session 2's default bench line reproducible, and does it follow the new tiles?
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for t in "-" "17=7,18=7,21=23" "-" "18=7" "17=7"; do
  spec=$t; [ "$t" = "-" ] && spec=""
  AVSR_TUNE=$spec timeout 400 python bench.py --no-cpu-baseline --no-roofline --no-bf16-leg --steps 8 --warmup 2 2>/dev/null | tail -1 > $O/r6_s4.json
  python - <<PY
import json; d=json.loads(open('$O/r6_s4.json').read()); p=d['parity']
print('tune [$t]', d['ms_per_step'], {k:(p[k]['grad_sample_cos_min'], p[k]['grad_sample_rel_l2_median'], p[k]['grad_norm_rel_err_median'], p[k]['dec_logits_full_rel_l2']) for k in ('A','B')})
PY
done
