#!/bin/bash
# round 5, GPU session 2: new evidence tests, refactored bench, train.py under graph replay, DP path on one rank, policy variant
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_wer_trained.py tests/test_trajectory.py tests/test_dropout_stats.py tests/test_train_eval_loops.py -q -m gpu -x -s 2>&1 | grep -v "Warn\|warn" | tail -40
timeout 420 python bench.py > gpurun_out/s2_bench_default.json 2>gpurun_out/s2_bench.err; python -c "
import json; d=json.load(open('gpurun_out/s2_bench_default.json')); print('bench', d['ms_per_step'], d['value'], d['dtype']); print({k:(v['dec_logits_full_rel_l2'] if isinstance(v,dict) and 'dec_logits_full_rel_l2' in v else None) for k,v in d['parity'].items()})"; tail -2 gpurun_out/s2_bench.err
timeout 600 python train.py --synthetic --synthetic-utterances 400 --steps 75 --time-last 20 --exp-dir '' --val-batches 0 --log-every 25 > gpurun_out/s2_train.log 2>&1; grep -v Warn gpurun_out/s2_train.log | tail -6
timeout 600 python train.py --synthetic --synthetic-utterances 400 --steps 30 --time-last 10 --exp-dir '' --val-batches 0 --log-every 25 --no-graph > gpurun_out/s2_train_eager.log 2>&1; grep -v Warn gpurun_out/s2_train_eager.log | tail -3
AVSR_BENCH_FORCE_DP=1 timeout 300 python bench.py --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg > gpurun_out/s2_bench_dp1.json 2>gpurun_out/s2_dp1.err; cat gpurun_out/s2_bench_dp1.json | cut -c1-3000; tail -3 gpurun_out/s2_dp1.err
P1="encoder=f16x2,decoder=f16x2"
timeout 300 python tools/mixed_sweep.py --tags=A,B,AA "$P1" > gpurun_out/s2_sweep.txt 2>/dev/null; cat gpurun_out/s2_sweep.txt
AVSR_MIXED_POLICY=$P1 timeout 300 python bench.py --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg > gpurun_out/s2_bench_p1.json 2>>gpurun_out/s2_bench.err; cut -c1-400 gpurun_out/s2_bench_p1.json
timeout 900 python -m pytest tests/test_bench_parity.py -q -m gpu -x -k "mixed" 2>&1 | tail -3
