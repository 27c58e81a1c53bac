#!/bin/bash
# round 5, GPU session 8: bf16 wire with the fused gather on one rank, DDP GPU tests, default bench (roofline traffic from the r5 counters)
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_rccl_single.py tests/test_ddp_gpu_two_ranks.py -q -m gpu -x 2>&1 | tail -2
for v in "AVSR_DDP=auto" "AVSR_DDP=buckets-graph1 AVSR_GRAD_WIRE=f32"; do
  n=s8_dp1_$(echo $v | tr -c 'a-zA-Z0-9\n' '_')
  env $v AVSR_BENCH_FORCE_DP=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-parity --no-bf16-leg --steps 16 --warmup 4 > gpurun_out/$n.json 2> gpurun_out/$n.err
  python -c "import json; d=json.loads(open('gpurun_out/$n.json').readline()); c=d['config']; print('DP1 $v', d['ms_per_step'], {k: c.get(k) for k in ('ddp_mode','communicators','grad_wire','rccl_ranks')}, c.get('bucket_overlap'))" | cut -c1-900
done
timeout 420 python bench.py > gpurun_out/s8_bench_default.json 2>gpurun_out/s8_bench.err; python -c "
import json; d=json.load(open('gpurun_out/s8_bench_default.json')); print('bench', d['ms_per_step'], d['value']); r=d['roofline']; print(r['achieved'], r['frac'], r['traffic'], r['traffic_source'][:160])"
