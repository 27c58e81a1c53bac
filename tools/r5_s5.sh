#!/bin/bash
# round 5, GPU session 5: pre-split ablation of the split-plane convolutions, train.py replay timing, trajectory tests
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out; export PYTHONFAULTHANDLER=1
timeout 300 python tools/microbench_presplit.py 2>&1 | grep -v Warn | tail -6
timeout 600 python -u train.py --synthetic --synthetic-utterances 400 --steps 110 --time-last 30 --exp-dir '' --val-batches 0 --log-every 34 > gpurun_out/s5_train.log 2>&1; grep -v "Warn\|warn" gpurun_out/s5_train.log | tail -6
timeout 300 python bench.py --no-parity --no-cpu-baseline --no-roofline --no-bf16-leg > gpurun_out/s5_bench.json 2>gpurun_out/s5_bench.err; cut -c1-330 gpurun_out/s5_bench.json
timeout 900 python -m pytest tests/test_trajectory.py -q -m gpu -s > gpurun_out/s5_tests.log 2>&1; grep -n "trajectory\[\|mean deviation\|passed\|failed\|Error" gpurun_out/s5_tests.log | cut -c1-400 | head
