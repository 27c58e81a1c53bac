#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_rccl_single.py tests/test_ddp_gpu_two_ranks.py -q -m gpu -x 2>&1 | tail -2
for v in "AVSR_DDP=auto" "AVSR_DDP=buckets"; do
  n=s9_dp1_$(echo $v | tr -c 'a-zA-Z0-9\n' '_')
  env $v AVSR_BENCH_FORCE_DP=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-parity --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/$n.json 2> gpurun_out/$n.err
  python -c "
import json; d=json.loads(open('gpurun_out/$n.json').readline()); c=d['config']; bo=c.get('bucket_overlap') or {}
print('DP1 $v', d['ms_per_step'], {k: c.get(k) for k in ('ddp_mode','communicators','grad_wire','rccl_ranks')}, [(b['bucket'], b['start_ms']) for b in bo.get('buckets', [])], bo.get('backward_done_ms'), bo.get('exposed_ms'))" | cut -c1-900
  tail -2 gpurun_out/$n.err | cut -c1-300
done
