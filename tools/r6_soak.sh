#!/bin/bash
# round 6: order-dependence screen (the GPU suite with its files in REVERSE order: the arena bug of this round showed only in one order)
# and a 600-step soak of the native training loop (graph replay, mixed numerics, dropout on)
cd "$(dirname "$0")/.."; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest $(ls -r tests/test_*.py) -x -q -m gpu 2>&1 | tail -6 > $O/r6_gputests_reversed.txt; tail -3 $O/r6_gputests_reversed.txt
timeout 900 python -u train.py --synthetic --synthetic-utterances 400 --steps 600 --time-last 100 --exp-dir "" --val-batches 0 --log-every 100 > $O/r6_soak_train.log 2>&1; tail -4 $O/r6_soak_train.log
