#!/bin/bash
# round 5, GPU session 4: bench after the fixes (default + fixed A + bf16 leg), train.py under graph replay, evidence tests
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out; export PYTHONFAULTHANDLER=1
timeout 420 python bench.py > gpurun_out/s4_bench_default.json 2>gpurun_out/s4_bench.err; python -c "
import json; d=json.load(open('gpurun_out/s4_bench_default.json')); print('bench', d['ms_per_step'], d['value'], d['dtype']); print({k:(v['dec_logits_full_rel_l2'], v['ctc_logits_raw_rel_l2']) for k,v in d['parity'].items() if isinstance(v,dict)}); print('bf16 leg', d['bf16']['ms_per_step']); print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline']['sample'][-90:])"; grep -v "Warn\|warn" gpurun_out/s4_bench.err | tail -3
timeout 600 python -u train.py --synthetic --synthetic-utterances 400 --steps 81 --time-last 20 --exp-dir '' --val-batches 0 --log-every 27 > gpurun_out/s4_train.log 2>&1; grep -v "Warn\|warn" gpurun_out/s4_train.log | tail -8
timeout 300 python bench.py --fixed A --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg > gpurun_out/s4_bench_fixedA.json 2>>gpurun_out/s4_bench.err; cut -c1-330 gpurun_out/s4_bench_fixedA.json
timeout 900 python -m pytest tests/test_train_eval_loops.py tests/test_trajectory.py -q -m gpu -s > gpurun_out/s4_tests.log 2>&1; grep -n "trajectory\[\|mean deviation\|losses eager\|passed\|failed\|Error" gpurun_out/s4_tests.log | cut -c1-500 | head
bash tools/gpu_timeline.sh r5_mixed_fixedA --no-bf16-leg > gpurun_out/s4_timeline.log 2>&1; head -45 gpurun_out/r5_mixed_fixedA_timeline.txt | cut -c1-150
