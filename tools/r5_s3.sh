#!/bin/bash
# round 5, GPU session 3: train.py under graph replay (fault handler on), evidence tests, bench after the StepGraphs refactoring
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out; export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_train_eval_loops.py -q -m gpu -x -s -k graph > gpurun_out/s3_graphtest.log 2>&1; grep -v "Warn\|warn" gpurun_out/s3_graphtest.log | tail -40
timeout 900 python -m pytest tests/test_wer_trained.py tests/test_trajectory.py tests/test_dropout_stats.py -q -m gpu -s > gpurun_out/s3_evidence.log 2>&1; grep -n "trajectory\[\|WER\[\|dropout run\|dropout sites\|passed\|failed\|Error\|assert" gpurun_out/s3_evidence.log | cut -c1-400 | head -40
timeout 600 python -u train.py --synthetic --synthetic-utterances 400 --steps 75 --time-last 20 --exp-dir '' --val-batches 0 --log-every 25 > gpurun_out/s3_train.log 2>&1; grep -v "Warn\|warn" gpurun_out/s3_train.log | tail -12
timeout 420 python bench.py > gpurun_out/s3_bench_default.json 2>gpurun_out/s3_bench.err; python -c "
import json; d=json.load(open('gpurun_out/s3_bench_default.json')); print('bench', d['ms_per_step'], d['value'], d['dtype']); print({k:(v['dec_logits_full_rel_l2'], v['ctc_logits_raw_rel_l2']) for k,v in d['parity'].items() if isinstance(v,dict)})"; tail -2 gpurun_out/s3_bench.err
timeout 900 python -m pytest tests/test_bench_parity.py -q -m gpu -x -k "mixed" -s 2>&1 | grep "PARITY\|passed\|failed" | cut -c1-700
