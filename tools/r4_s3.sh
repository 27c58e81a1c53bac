#!/bin/bash
# round 4, session 3: GPU gate of the mixed mode (kernel tests, parity at batch A / B in all four modes, RCCL single-rank incl. the
# bf16 wire), the default bench line, the data-parallel machinery on one rank, one-step timeline of the mixed step
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mixed_mode.py tests/test_bench_parity.py tests/test_rccl_single.py -q -m gpu -x -s 2>&1 | grep -E "PARITY|passed|failed|Error|error" | cut -c1-600 | tee gpurun_out/s3_tests.txt
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/s3_bench_default.json 2> gpurun_out/s3_bench_default.err; echo "bench rc=$?"
python -c "
import json
d = json.loads(open('gpurun_out/s3_bench_default.json').readline())
print('default', d['ms_per_step'], d['value'], d['dtype'][:40], '| parity logits', d['parity']['dec_logits_rel_l2'], 'ctc_logp', d['parity']['ctc_logp_rel_l2'])
print('  bf16 leg', d['bf16']['ms_per_step'], d['bf16']['value'], d['bf16']['parity']['dec_logits_rel_l2'])
print('  roofline', {k: d['roofline'][k] for k in ('kernel', 'launches', 'avg_us', 'achieved', 'frac')})
print('  cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
for v in "AVSR_DDP=buckets-graph" "AVSR_DDP=buckets-graph1" "AVSR_DDP=buckets-graph AVSR_GRAD_WIRE=bf16" "AVSR_DDP=torch"; do
  env $v AVSR_BENCH_FORCE_DP=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-parity --no-bf16-leg --steps 16 --warmup 4 2>gpurun_out/s3_dp.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['config']; print('DP1 $v', d['ms_per_step'], {k: c[k] for k in ('ddp_mode','communicators','grad_wire','rccl_ranks')})" || tail -3 gpurun_out/s3_dp.err
done
bash tools/gpu_timeline.sh s3_mixed --no-bf16-leg > gpurun_out/s3_tl.out 2>&1; head -75 gpurun_out/s3_mixed_timeline.txt
